#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc -o p -- python bench.py --steps 3 --warmup 1 --pp-only --cpu-scans 0 > gpurun_out/pmc.log 2>&1
ls gpurun_out/pmc
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc/*counter_collection.csv')
print(f)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'pp2' in k:
        print(k, {c: sum(x)/len(x) for c,x in v.items()})
PY
