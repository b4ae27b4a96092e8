#!/bin/bash
# kernel times of the grid frame sort (clear / rank / scan / place) for 11 and 361 frames per batch
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_sort
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sort -o p -- python tools/sort_probe.py > gpurun_out/prof_sort.log 2>&1
f=$(find gpurun_out/prof_sort -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "frame_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i, out = 0, []
while i + 3 < len(rows):
    g = rows[i:i + 4]
    if "frame_clear" in g[0]["Kernel_Name"]:
        d = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in g]
        out.append((g[2].get("Grid_Size", g[2].get("Grid_Size_X", "?")), d, (int(g[-1]["End_Timestamp"]) - int(g[0]["Start_Timestamp"])) / 1e3))
        i += 4
    else:
        i += 1
for o in out:
    print("scan-kernel grid", o[0], "clear/rank/scan/place us", [round(x, 1) for x in o[1]], "first start -> last end", round(o[2], 1))
PY
rm -rf gpurun_out/prof_sort
