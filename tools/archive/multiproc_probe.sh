#!/bin/bash
# How far is one process from what the GPU can take?  P independent bench processes on the same GPU.
for P in 1 2 4; do
  for p in $(seq 1 $P); do
    python bench.py --streams 2 --steps 240 --warmup 8 --cpu-scans 0 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" > /tmp/mp_$p.txt &
  done
  wait
  python - <<PY
vals=[float(open('/tmp/mp_%d.txt'%p).read()) for p in range(1,$P+1)]
print("processes $P x 2 streams: per-process", [round(v) for v in vals], "sum %.0f scans/s" % sum(vals))
PY
done
