#!/bin/bash
# GPU box: per-kernel times of the frame sort (tools/sort_probe.py: 11 / 44 / 361 frames per launch batch) under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/sort_probe.py > /dev/null 2>&1   # warm the box
rm -rf /tmp/sp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sp -o sp -- python tools/sort_probe.py 2>/dev/null | grep "frames:"
python - <<'EOF'
import csv, glob
from collections import defaultdict
f = glob.glob("/tmp/sp/**/sp_kernel_trace.csv", recursive=True)[0]
by = defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "frame_" in n:
        key = n.replace("(anonymous namespace)::", "").split("(")[0]
        by[(key, int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0)))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g), v in sorted(by.items()):
    v.sort()
    print("%-24s grid %8d: %3d launches, median %.1f us, min %.1f us" % (k, g, len(v), v[len(v) // 2] / 1e3, v[0] / 1e3))
EOF
