#!/bin/bash
# PMC counters of the block path's kernels (separate passes), via tools/pp_block_probe.py (PROBE_ARGS: other shapes, e.g. C5:
# PROBE_ARGS="--nusc --trav 20 --frames 16 --n 35000")
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for pass in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" ; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmc4_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc4_$tag -o p -- python tools/pp_block_probe.py --scans ${PP_BLOCK:-16} --reps 2 --shards 2 $PROBE_ARGS > gpurun_out/pmc4_$tag.log 2>&1
done
python - <<'PY'
import csv,glob,collections,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc4_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
out={}
for k,v in sorted(acc.items()):
    if not (k.startswith('b4_') or k.startswith('ppb_')): continue
    out[k]={c: sum(x)/len(x) for c,x in v.items()}
    out[k]['launches']=max(len(x) for x in v.values())
    print(k, {a: (round(b) if b > 100 else b) for a, b in out[k].items()})
json.dump(out, open('gpurun_out/pp4_pmc.json','w'), indent=1)
PY
rm -rf gpurun_out/pmc4_*/
