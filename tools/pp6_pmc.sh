#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MODEST_PP_FRAMES_PATH=gather-wave
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_REQ_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmc6_$tag
  NSCAN=1 timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc6_$tag -o p -- python tools/pp5_microbench.py > gpurun_out/pmc6_$tag.log 2>&1
done
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc6_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    if 'pp6' in k or 'pp5_plan' in k:
        print(k[:40], {c: round(sum(x)/len(x)/1e6,2) for c,x in v.items()}, '(x1e6 per launch)')
PY
