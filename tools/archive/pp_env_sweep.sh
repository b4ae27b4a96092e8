#!/bin/bash
# env-variable sweeps of the block probe: SWEEP="VAR:v1,v2,..." GS="16 4"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/sweep.txt
var=${SWEEP%%:*}; vals=${SWEEP#*:}
for v in ${vals//,/ }; do
  for g in ${GS:-16 4}; do
    echo "$var=$v scans $g: $(env $var=$v python tools/pp_block_probe.py --scans $g --reps 8 --shards 2 $PROBE_ARGS 2>&1 | grep 'PARITY\|^block\|DIFFER' | tr '\n' ' ')" >> gpurun_out/sweep.txt
  done
done
cat gpurun_out/sweep.txt
