"""GPU box: determinism soak of the mask + box + label stages under multi-process load: P processes repeat the
same scans R times and compare every output with their first result."""
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def work(args):
    pid, reps = args
    import torch
    from modest_amd import config, synth
    from modest_amd.gen_label_files import gen_label_scan
    from modest_amd.generate_mask import generate_mask_scan
    from modest_amd.utils import kitti_util
    dev = torch.device("cuda:0")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "c.txt"), "w").write(synth.CALIB_TXT)
        calib = kitti_util.Calibration(os.path.join(d, "c.txt"))
    margs = config.compose("generate_mask", ["data_root=/unused"])
    largs = config.compose("generate_label_files", ["data_root=/unused"])
    scans = []
    for k in range(3):
        s = synth.make_scan(50 + k, n_live=30000, n_trav=2, n_frames=1)
        raw = np.ascontiguousarray(s.live_raw)
        rng = np.random.default_rng(k)
        pp = np.clip(0.5 + 0.5 * np.sin(raw[:, 0] * 0.3) + rng.normal(0, 0.03, len(raw)), 0, 1).astype(np.float32)
        scans.append((raw, pp, torch.from_numpy(raw).to(dev), torch.from_numpy(pp).to(dev)))
    first, bad = {}, []
    # the PP stage too (frame path): counts of one mid-size scan, repeated
    from modest_amd.frame_store import FrameStore
    sp = synth.make_scan(21, n_live=20000, n_trav=5, n_frames=8, n_per_frame=20000, keep_frames=True)
    st = FrameStore(dev, 0.3)
    items, hist, rels = [], [], []
    for t, fr in enumerate(sp.frames):
        for f, (rawf, rel, W) in enumerate(fr):
            items.append(((t, f), torch.from_numpy(rawf).to(dev), W))
            hist.append(((t, f), t))
            rels.append(rel)
    items.append(("live", torch.from_numpy(sp.live_raw).to(dev), sp.live_W))
    st.insert_many(items)
    rels = np.stack(rels)
    c0 = None
    desc = st.describe("live", sp.live_rel, [k for k, _ in hist], [t for _, t in hist], rels, False)
    for r in range(max(reps // 4, 1)):
        if r % 2 == 0:
            _, c = st.pp_score("live", sp.live_rel, hist, rels, sp.world_from_ref, 5, return_counts=True)
            cs = [c]
        else:   # the chain of several scans per launch (modest_pp_score_frames_batch): three times the same scan
            _, cs = st.pp_score_batch(["live"] * 3, [desc] * 3, 5, return_counts=True)
        for c in cs:
            c = c.cpu().numpy()
            if c0 is None:
                c0 = c
            elif not np.array_equal(c, c0):
                bad.append((pid, r, "pp", ["counts"], int((c != c0).sum())))
    from modest_amd.generate_mask import generate_mask_chain
    for r in range(reps):
        if r % 2 == 1:   # every other repeat: the three scans as ONE chain of launches (what the CLI and bench.py run)
            chain = generate_mask_chain([dict(ptc=raw, pp_score=pp, random_state=np.random.RandomState(k), ptc_dev=rd, pp_dev=pd)
                                         for k, (raw, pp, rd, pd) in enumerate(scans)], calib, margs, as_rows=True)
        for k, (raw, pp, rd, pd) in enumerate(scans):
            if r % 2 == 1:
                labels, rows, info = chain[k]
            else:
                labels, rows, info = generate_mask_scan(raw, pp, calib, margs, random_state=np.random.RandomState(k), ptc_dev=rd,
                                                        pp_dev=pd, as_rows=True)
            text, _ = gen_label_scan(rows, calib, largs)
            cur = (labels, rows, text, info["plane"])
            if k not in first:
                first[k] = cur
            else:
                f = first[k]
                what = [n for n, a, b in (("labels", f[0], cur[0]), ("boxes", f[1], cur[1]), ("plane", f[3], cur[3]))
                        if a.shape != b.shape or not np.array_equal(a, b)]
                if f[2] != cur[2]:
                    what.append("text")
                if what:
                    bad.append((pid, r, k, what, int((f[0] != cur[0]).sum()) if f[0].shape == cur[0].shape else -1))
    return bad


if __name__ == "__main__":
    P, R = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 400
    with mp.get_context("spawn").Pool(P) as pool:
        res = pool.map(work, [(p, R) for p in range(P)])
    bad = [b for r in res for b in r]
    print(f"{P} processes x ({R} repeats x 3 scans of the mask / box / label stages + {max(R // 4, 1)} of the PP stage) = "
          f"{P * (R * 3 + max(R // 4, 1))} runs; mismatches: {len(bad)}")
    for b in bad[:20]:
        print(" ", b)
