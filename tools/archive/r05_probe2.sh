#!/bin/bash
# ablations of the task-list join (MODEST_PP4_DBG: 1 no pair loop, 2 no packed chunks, 8 no one-cell tasks) and its grid size
cd $GRAFT_REPO_ROOT
for d in ${DBGS:-0 1 2 8 10}; do
  echo "== dbg $d"; MODEST_PP4_DBG=$d python tools/pp_block_probe.py --scans 16 --reps 3 --shards 1 2>&1 | grep "^block"
done
for j in ${JWGS:-2 3 6 8}; do
  echo "== jwg $j"; MODEST_PP4_JWG=$j python tools/pp_block_probe.py --scans 16 --reps 3 --shards 1 2>&1 | grep "^block"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_p2 -o p2 -- python $GRAFT_REPO_ROOT/tools/pp_block_probe.py --scans 16 --reps 4 --shards 2 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_p2 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:24]:
    print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
