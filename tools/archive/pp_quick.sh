#!/bin/bash
# GPU box: PP parity tests, then the ordered launch list of one scan
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_pp.py tests/test_gpu_frames.py -m gpu -x -q 2>&1 | grep -i "passed\|failed\|error" | tail -3
bash tools/scan_trace.sh > /dev/null 2>&1
python - <<'PY'
rows = [l.rstrip('\n') for l in open('gpurun_out/scan_trace.txt')]
tot = 0.0; n = 0; pp = 0.0; npp = 0
for l in rows[1:]:
    f = l.split()
    d = float(f[5]); name = ' '.join(f[7:])
    if 'pp' in name: pp += d; npp += 1; print(l)
    else: tot += d; n += 1
print("pp kernels: %.1f us in %d launches; everything else %.1f us in %d launches" % (pp, npp, tot, n))
PY
