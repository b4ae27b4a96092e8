#!/bin/bash
# generic A/B of -D variants of pp_v4.hip on the GPU box: VARIANTS="flagsA|flagsB|..." bash tools/pp_variants_ab.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/ab.txt
IFS='|' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  export MODEST_EXTRA_CXXFLAGS="$v"
  python -c "from modest_amd import build; build.build(verbose=False)" > gpurun_out/ab_build.log 2>&1 || { echo "BUILD FAILED $v" >> gpurun_out/ab.txt; tail -5 gpurun_out/ab_build.log >> gpurun_out/ab.txt; continue; }
  for g in ${GS:-16 4}; do
    echo "[$v] scans $g: $(python tools/pp_block_probe.py --scans $g --reps 8 --shards 2 $PROBE_ARGS 2>&1 | grep 'PARITY\|^block\|DIFFER' | tr '\n' ' ')" >> gpurun_out/ab.txt
  done
  if [ -n "$KS" ]; then PP_BLOCK=16 KS_NAME=ab_ks.csv bash tools/r05_kstats.sh $PROBE_ARGS 2>&1 | grep "b4_" | head -${KS} >> gpurun_out/ab.txt; fi
  unset MODEST_EXTRA_CXXFLAGS
done
cat gpurun_out/ab.txt
