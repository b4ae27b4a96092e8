#!/bin/bash
# round 6, first GPU call: the new tests (mixed-T blocks, blocks of 32 at full size), the isolated block timings the round starts
# from, the join's phase times (MODEST_PP4_DBG=512) and its issue counters (what bounds b4_join -- VERDICT r5 item 1a)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_block.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06_tests_new.txt
cat gpurun_out/r06_tests_new.txt
python tools/pp_block_probe.py --scans 32 --reps 6 --shards 2 2>&1 | grep 'PARITY\|^block\|^chain\|DIFFER' > gpurun_out/r06_probe_best.txt
python tools/pp_block_probe.py --scans 32 --reps 6 --shards 2 --matched 8,3,15 2>&1 | grep 'PARITY\|^block\|^chain\|DIFFER\|sharing' > gpurun_out/r06_probe_matched.txt
cat gpurun_out/r06_probe_best.txt gpurun_out/r06_probe_matched.txt
MODEST_PP4_DBG=512 python tools/pp_block_probe.py --scans 32 --reps 1 --shards 1 2>&1 | grep 'b4_join' | tail -3 | cut -c1-900 > gpurun_out/r06_join_phases.txt
cat gpurun_out/r06_join_phases.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_LEVEL_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" ; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmc6_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc6_$tag -o p -- python tools/pp_block_probe.py --scans 32 --reps 2 --shards 2 > gpurun_out/pmc6_$tag.log 2>&1
done
python - <<'PY'
import csv,glob,collections,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc6_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
out={}
for k,v in sorted(acc.items()):
    if not k.startswith('b4_join'): continue
    out[k]={c: sum(x)/len(x) for c,x in v.items()}
    out[k]['launches']=max(len(x) for x in v.values())
    print(k, {a: round(b) for a, b in out[k].items()})
json.dump(out, open('gpurun_out/r06_join_issue_counters.json','w'), indent=1)
PY
rm -rf gpurun_out/pmc6_*/
