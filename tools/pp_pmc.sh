#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ issue counters of the PP kernels.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for pass in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_$tag -o p -- python bench.py --steps 8 --warmup 4 --pp-only --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 > gpurun_out/pmc_$tag.log 2>&1
done
python - <<'PY'
import csv,glob,collections,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
out={}
for k,v in sorted(acc.items()):
    if 'pp' not in k and 'fillBuffer' not in k: continue
    out[k]={c: sum(x)/len(x) for c,x in v.items()}
    out[k]['launches']=max(len(x) for x in v.values())
    print(k, out[k])
json.dump(out, open('gpurun_out/pp_pmc.json','w'), indent=1)
PY
python tools/pp_traffic.py
