#!/bin/bash
# per-kernel durations of any python tool: tools/prof_py.sh <name> tools/x.py args...   (always bounded by timeout)
cd /tmp && export TMPDIR=/tmp
NAME=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$NAME
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $NAME -- python $GRAFT_REPO_ROOT/"$@" > $OUT/run.log 2>&1
tail -6 $OUT/run.log
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel stats"); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print(f"{r['Name'].replace('(anonymous namespace)::','')[:60]:60s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
