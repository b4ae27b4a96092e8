#!/bin/bash
# which copies does the pipeline issue?  (rocprofv3 --kernel-trace --memory-copy-trace on one process, one stream)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_cp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/prof_cp -o p -- python bench.py --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 --steps 64 --warmup 16 --sharing best > gpurun_out/prof_cp.log 2>&1
ls gpurun_out/prof_cp/*/ 2>/dev/null | head; 
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_cp/**/*memory_copy_trace.csv', recursive=True)
print(f)
if f:
    rows = list(csv.DictReader(open(f[0])))
    print(rows[0].keys())
    acc = collections.Counter(); tim = collections.Counter(); 
    for r in rows:
        key = (r.get('Direction'), int(r.get('Bytes', r.get('Size', 0)) or 0) // 1024)
        acc[key] += 1; tim[key] += (int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k, v in sorted(acc.items(), key=lambda kv: -tim[kv[0]])[:40]:
        print(k, 'KiB x', v, 'total %.2f ms avg %.1f us' % (tim[k] / 1e6, tim[k] / v / 1e3))
f = glob.glob('gpurun_out/prof_cp/**/*kernel_trace.csv', recursive=True)
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if 'copyBuffer' in r['Kernel_Name']]
    print('copyBuffer launches', len(rows))
    import statistics
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
    d.sort()
    print('durations us: p10 %.1f p50 %.1f p90 %.1f max %.1f' % tuple(x / 1e3 for x in (d[len(d)//10], d[len(d)//2], d[len(d)*9//10], d[-1])))
    print(collections.Counter((r.get('Grid_Size') or r.get('Grid_Size_X'), r.get('Workgroup_Size') or r.get('Workgroup_Size_X')) for r in rows).most_common(12))
PY
rm -rf gpurun_out/prof_cp
