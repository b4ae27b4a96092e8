#!/bin/bash
# kernel timeline of ONE mask-stage chain (tools/rsd_latency.py B): start, duration, gap to the previous kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=${1:-4}
rm -rf gpurun_out/prof_chain
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_chain -o p -- python tools/rsd_latency.py $B > gpurun_out/prof_chain.log 2>&1
grep "device" gpurun_out/prof_chain.log | tail -1
f=$(find gpurun_out/prof_chain -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last complete chain: from the last rsd_draw / cdb / mad start backwards -> find last occurrence of the first kernel of a chain
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "") for r in rows]
first = max(i for i, n in enumerate(names) if n.startswith("cdb_candidates") or n.startswith("rsd_cand") or n.startswith("rsd_prepare"))
# walk back to the beginning of that chain: previous kernel gap > 200 us
i0 = first
while i0 > 0 and int(rows[i0]["Start_Timestamp"]) - int(rows[i0 - 1]["End_Timestamp"]) < 200000:
    i0 -= 1
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
tot_k = 0
for r, n in list(zip(rows, names))[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {n[:50]}")
    tot_k += e - s
    prev_end = e
print(f"chain: {(prev_end - t0) / 1e3:.1f} us wall on the GPU timeline, {tot_k / 1e3:.1f} us inside kernels")
PY
rm -rf gpurun_out/prof_chain
