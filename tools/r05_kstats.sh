#!/bin/bash
# per-kernel times of the block path (rocprofv3 --kernel-trace --stats) through tools/pp_block_probe.py; args: extra probe flags
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ks -o p -- python tools/pp_block_probe.py --scans ${PP_BLOCK:-16} --reps 4 --shards 2 "$@" > gpurun_out/prof_ks.log 2>&1
f=$(find gpurun_out/prof_ks -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${KS_NAME:-r05_pp_block_kernel_stats.csv}
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:30]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
rm -rf gpurun_out/prof_ks
