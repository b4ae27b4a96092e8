"""modest_frame_sort: time of one launch batch for a scan's 11 new frames and for a cold scan's 361 (HIP events around the
store's insert_many), C3 and C5 frame sizes."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import _lib, synth
from modest_amd.frame_store import FrameStore

def main():
    _lib.load()
    dev = torch.device("cuda:0")
    world = synth.make_world(0)
    l2e, K = synth.default_l2e(), synth.kitti2nu(False)
    for n_pts in (30000, 35000):
        frames = []
        for j in range(24):
            pose = synth._pose_matrix(2.0 * j, 0.3, 0.01)
            frames.append((torch.from_numpy(synth.sample_frame(world, 100 + j, n_pts, pose, l2e)).to(dev), pose @ l2e @ K))
        for nb in (11, 44, 361):
            store = FrameStore(dev, 0.3)
            store.insert_many([(10**6 + k, frames[k % 24][0], frames[k % 24][1]) for k in range(nb)])   # warm: arena, anchor
            ts = []
            for rep in range(5):
                items = [(rep * 1000 + k, frames[k % 24][0], frames[k % 24][1]) for k in range(nb)]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                store.insert_many(items, blocking=False)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            # correctness of the last batch: a permutation whose runs follow the table
            f = store.frames[4000]
            xyz, perm, tab = f.xyz.cpu().numpy(), f.perm.cpu().numpy().astype(np.int64), f.tab.cpu().numpy().astype(np.int64)
            raw = frames[0][0].cpu().numpy()
            ok = np.array_equal(np.sort(perm), np.arange(len(perm))) and np.array_equal(xyz, raw[perm, :3]) and np.all(np.diff(tab) >= 0)
            print(f"{n_pts} pts x {nb} frames: {np.median(ts):.1f} us per batch (min {min(ts):.1f}), permutation ok: {ok}", flush=True)

if __name__ == "__main__":
    main()
