#!/bin/bash
# kernel + copy timeline of the whole chain of B scans (mask stage, box tail, label stage) of tools/chain_latency.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=${1:-16}
rm -rf gpurun_out/prof_chain
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_chain -o p -- python tools/chain_latency.py $B > gpurun_out/prof_chain.log 2>&1
grep "chain of\|alone" gpurun_out/prof_chain.log
python - <<'PY'
import csv, glob
ev = []
for f in glob.glob("gpurun_out/prof_chain/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")))
for f in glob.glob("gpurun_out/prof_chain/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
ev.sort()
# the last occurrence of rsd_draw preceded by a long gap = start of the last mask stage inside r.steps(); print from the mark before
idx = [i for i, e in enumerate(ev) if e[2].startswith("cdb_candidates")]
# r.steps chain is the 6th-from-last cdb (4 reps of mask_stage_batch alone follow) -> take the one before the last 4
i0 = idx[-5]
t0 = ev[i0][0]
prev = t0
for s, e, n in ev[i0:idx[-4]]:
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {n[:60]}")
    prev = max(prev, e)
PY
rm -rf gpurun_out/prof_chain
