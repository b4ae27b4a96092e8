"""HBM bytes per scan of the PP stage from the PMC passes of tools/pp_pmc.sh (gpurun_out/pp_pmc.json):
FETCH_SIZE / WRITE_SIZE are reported in KiB per launch; a launch of the batched chain (ppb_*) processes PP_BATCH
scans (bench.py --pp-batch, default 4).  Both counters are calibrated on the count pass (ppb_stream<false>), whose
traffic is known (it reads the 12-byte history points once and writes its count matrix).
Writes gpurun_out/pp_traffic.json (copied to profiles/rNN_pp_traffic.json, which bench.py reads)."""
import json
import os

M, N_LIST = 10_800_000, 7168
B = int(os.environ.get("PP_BATCH", "4"))
N_WG = max(32, (512 // B) & ~1)   # streaming workgroups per scan of a chain on a 256-CU context (pp_count.hip)
d = {k: v for k, v in json.load(open("gpurun_out/pp_pmc.json")).items() if k.startswith("ppb_")}
per_scan = {k: 1.0 / B for k in d}      # launches per scan
fetch = sum(v.get("FETCH_SIZE", 0.0) * 1024 * per_scan[k] for k, v in d.items())
write = sum(v.get("WRITE_SIZE", 0.0) * 1024 * per_scan[k] for k, v in d.items())
cnt = [v for k, v in d.items() if "ppb_stream<false" in k][0]
f_fac = B * (12.0 * M + 28 * 1024) / (cnt["FETCH_SIZE"] * 1024)
w_fac = B * (N_WG * N_LIST * 4.0) / (cnt["WRITE_SIZE"] * 1024)
out = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 8 "
              "--warmup 4 --pp-only --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1; tools/pp_pmc.sh + tools/pp_traffic.py",
    "scans_per_launch": B,
    "fetch_raw_bytes_per_scan": fetch,
    "write_raw_bytes_per_scan": write,
    "fetch_calibration": {
        "kernel": "ppb_stream<count> reads 12 B x 10.8 M points per scan (frame store, through the descriptor table) + 28 KB of tables with 16-byte coalesced loads",
        "factor": f_fac,
        "note": "MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read on gfx950 -> doubled",
    },
    "write_calibration": {"kernel": "ppb_stream<false> writes a %d x 7168 x 4 B count matrix per scan" % N_WG, "factor": w_fac},
    "hbm_bytes_per_scan": fetch * f_fac + write * w_fac,
    "algorithmic_bytes_per_scan": 12.0 * M + 16.0 * 30_000,
    "per_kernel_bytes": {k: {"fetch": v.get("FETCH_SIZE", 0.0) * 1024 * f_fac * per_scan[k],
                             "write": v.get("WRITE_SIZE", 0.0) * 1024 * w_fac * per_scan[k]} for k, v in sorted(d.items())},
}
json.dump(out, open("gpurun_out/pp_traffic.json", "w"), indent=1)
print("HBM bytes per scan %.1f MB (fetch %.1f x %.3f, write %.1f x %.3f)" %
      (out["hbm_bytes_per_scan"] / 1e6, fetch / 1e6, f_fac, write / 1e6, w_fac))
