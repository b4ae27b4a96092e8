#!/bin/bash
# GPU box: ordered list of the GPU launches of ONE cycle of a process (full pipeline, one process, one stream):
# the PP chain of the next 4 scans, then the mask / box / label launches of 4 scans
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace1
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace1 -o t -- python bench.py --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 --steps 16 --warmup 4 > gpurun_out/trace1.log 2>&1
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/trace1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:50], 'k'))
for f in glob.glob('gpurun_out/trace1/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY %s %s B' % (r.get('Direction',''), r.get('Bytes', r.get('Size',''))), 'c'))
rows.sort()
# one cycle: from one chain's join to the next chain's join
j = [i for i, r in enumerate(rows) if 'ppb_join' in r[2] or 'pp3_join' in r[2]]
lo, hi = j[2], j[3]   # a cycle of the timed full-pipeline section (the last launches are the isolated PP chains)
# start the listing at the first launch after the previous scan's last kernel
t0 = rows[lo][0]
out = open('gpurun_out/scan_trace.txt', 'w')
prev = None
for r in rows[lo:hi]:
    gap = (r[0] - prev) / 1e3 if prev else 0.0
    out.write("%9.1f us  +%7.1f gap  %7.1f us  %s\n" % ((r[0] - t0) / 1e3, gap, (r[1] - r[0]) / 1e3, r[2]))
    prev = r[1]
out.close()

PY
