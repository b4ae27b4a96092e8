#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_mad
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mad -o m -- python tools/mad_bench.py > gpurun_out/mad.log 2>&1
grep -E "candidates|mad |Error|error" gpurun_out/mad.log | head -6
grep mad_kernel gpurun_out/prof_mad/m_kernel_stats.csv | cut -c50-140
timeout 600 python -m pytest tests/test_gpu_stages.py -m gpu -q -x 2>&1 | tail -2
