"""Times the PP kernels on a full-size synthetic scan with HIP events (GPU box)."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from modest_amd import ops, synth

def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 36
    t0 = time.time()
    import os
    s = synth.make_scan(0, n_live=30000, n_trav=T, n_frames=F, point_order=os.environ.get("SYNTH_ORDER", "shuffled"))
    print("gen %.1fs" % (time.time() - t0), flush=True)
    dev = torch.device("cuda:0")
    off = np.cumsum([0] + [len(h) for h in s.hist])
    live = torch.from_numpy(s.live_xyz).to(dev)
    hist = torch.from_numpy(np.concatenate(s.hist)).to(dev)
    M = int(off[-1])
    H = ops.pp_score(live, hist, off, 0.3)
    torch.cuda.synchronize()
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K = 10
        for _ in range(K):
            H = ops.pp_score(live, hist, off, 0.3)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        bytes_alg = 12 * M + 16 * 30000
        print(json.dumps(dict(M=M, ms_per_scan=ms, alg_GBps=bytes_alg / ms / 1e6, frac_of_8TBps=bytes_alg / ms / 1e6 / 8000)), flush=True)
    print("H stats", float(H.mean()), float(H.min()), float(H.max()))

if __name__ == "__main__":
    main()
