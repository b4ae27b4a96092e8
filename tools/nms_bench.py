"""NMS at detector scale through the C ABI (modest_nms_bev / modest_nms_normal): keep list vs the
oracle and time per call, for this build and (argv[1]) an older libmodest_hip.so (GPU box)."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import labels as ol

def bench(path, tag):
    lib = C.CDLL(path)
    ctx = C.c_void_p()
    assert lib.modest_ctx_create(0, C.byref(ctx)) == 0
    rng = np.random.default_rng(0)
    for n in (512, 2048, 4096, 5000, 12000, 20000):
        side = 12.0 * np.sqrt(n / 300.0)
        big = np.c_[rng.uniform(-side, side, (n, 2)), np.zeros(n), rng.uniform(1, 5, (n, 2)), np.ones(n),
                    rng.uniform(-3.2, 3.2, n)].astype(np.float32)
        order = np.argsort(-rng.uniform(size=n), kind="stable")
        srt = np.ascontiguousarray(big[order])
        b = torch.from_numpy(srt).to("cuda:0")
        keep = np.zeros(n, dtype=np.int64)
        nk = C.c_int(0)
        for rotated, fn in ((True, lib.modest_nms_bev), (False, lib.modest_nms_normal)):
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
            assert fn(ctx, b.data_ptr(), n, 0.1, keep.ctypes.data, C.byref(nk), None) == 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn(ctx, b.data_ptr(), n, 0.1, keep.ctypes.data, C.byref(nk), None)
            ms = (time.perf_counter() - t0) / 20 * 1e3
            ok = None
            if n <= 2048 or n in (5000, 20000):
                ok = bool(np.array_equal(keep[: nk.value], ol.nms(srt, 0.1, rotated=rotated)))
            print(f"[{tag}] n={n} rotated={rotated}: {ms:.3f} ms/call, kept {nk.value}, equals oracle: {ok}", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        bench(sys.argv[1], "old")
    bench("modest_amd/lib/libmodest_hip.so", "new")
