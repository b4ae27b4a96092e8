#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for o in shuffled azimuth; do
echo "== order $o"; SYNTH_ORDER=$o timeout 300 python tools/pp_microbench.py 2>&1 | grep "ms_per_scan" | tail -1 | cut -c1-70
rm -rf gpurun_out/abl
SYNTH_ORDER=$o timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl -o a -- python tools/pp_microbench.py > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/abl/**/*kernel_stats.csv', recursive=True)[0]
out=[]
for r in csv.DictReader(open(f)):
    for k in ('pp3_stream<true>','pp3_stream<false>','pp3_join'):
        if k in r['Name']: out.append('%s %.1f' % (k, float(r['AverageNs'])/1e3))
print(' | '.join(out))
PY
done
