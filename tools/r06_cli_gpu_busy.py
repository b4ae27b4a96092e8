"""GPU box, after `rocprofv3 --kernel-trace -d DIR -- python tools/r06_cli_trace.py --modes pp|fused --reps 1`: how busy the GPU is
while the worker processes run (union of the kernels' [start, end) over all processes) and which kernels fill it."""
import csv
import glob
import sys
from collections import defaultdict

def kname(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].split("<")[0][:60]


rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kname(r["Kernel_Name"]), f))
files = sorted({r[3] for r in rows})
print("trace files:", len(files), "kernels:", len(rows))
# the worker processes: the files with the most kernels; the window: from the 5th percentile of their starts to the last end
per = defaultdict(list)
for r in rows:
    per[r[3]].append(r)
workers = sorted(per, key=lambda f: -len(per[f]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 8]
W = [r for f in workers for r in per[f]]
W.sort()
for f in workers:
    s = sorted(per[f])
    busy = sum(e - b for b, e, _, _ in s)
    print("  worker %s: %d kernels, first %.1f ms, span %.1f ms, kernel time %.1f ms" % (f.split("/")[-1][:24], len(s), (s[0][0] - W[0][0]) / 1e6, (s[-1][1] - s[0][0]) / 1e6, busy / 1e6))
t0, t1 = W[0][0], max(r[1] for r in W)
# steady part: drop the first 15 % (cold starts) and the last 10 % (stragglers)
a, b = t0 + 0.15 * (t1 - t0), t1 - 0.10 * (t1 - t0)
ev = []
tot = defaultdict(float)
for s, e, n, _ in W:
    s2, e2 = max(s, a), min(e, b)
    if e2 > s2:
        ev.append((s2, 1)); ev.append((e2, -1))
        tot[n] += e2 - s2
ev.sort()
busy = 0; depth = 0; last = a; conc = 0.0
for t, d in ev:
    if depth > 0:
        busy += t - last
        conc += depth * (t - last)
    depth += d; last = t
print("window %.1f ms (steady part of %.1f ms): GPU has >= 1 kernel running %.1f %% of it, mean kernels in flight while busy %.2f"
      % ((b - a) / 1e6, (t1 - t0) / 1e6, 100.0 * busy / (b - a), conc / max(busy, 1)))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-60s %.1f %% of the window (summed over processes)" % (n, 100.0 * v / (b - a)))
