"""GPU box: host-to-device copy rate from pinned memory, N processes at once (what bounds the ingest-inclusive rates)."""
import subprocess
import sys
import time

code = r'''
import sys, time, torch
mb = int(sys.argv[1])
h = torch.empty((mb << 20,), dtype=torch.uint8, pin_memory=True); h.zero_()
d = torch.empty_like(h, device="cuda")
for _ in range(3):
    d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 1.5:
    for _ in range(4):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); n += 4
dt = time.perf_counter() - t0
print("  %d MB pieces: %.1f GB/s" % (mb, n * (mb << 20) / dt / 1e9), flush=True)
'''
for procs in (1, 4, 8):
    for mb in (5, 21, 170):
        print("%d process(es):" % procs, flush=True)
        ps = [subprocess.Popen([sys.executable, "-c", code, str(mb)]) for _ in range(procs)]
        for p in ps:
            p.wait()
