#!/bin/bash
# GPU box: stage parity tests, then the ordered launch list of one full-pipeline scan
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -4
bash tools/scan_trace.sh > /dev/null 2>&1
python - <<'PY'
rows = [l.rstrip('\n') for l in open('gpurun_out/scan_trace.txt')]
tot = 0.0; n = 0; pp = 0.0
for l in rows:
    f = l.split()
    d = float(f[5]); name = ' '.join(f[7:])
    if name.startswith('pp') or 'pp3' in name or 'pp_' in name: pp += d
    else: tot += d; n += 1
    print(l)
print("mask+label stage: %.1f us in %d launches; pp stage %.1f us" % (tot, n, pp))
PY
