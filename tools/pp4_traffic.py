"""HBM bytes per scan of the PP BLOCK path (modest_pp_score_block) from the PMC passes of tools/pp4_pmc.sh
(gpurun_out/pp4_pmc.json): FETCH_SIZE / WRITE_SIZE are reported in KiB per launch; a launch processes G scans.
Calibration on kernels whose traffic is known: b4_seg_hist reads every record of the block store once (16 B x R) with wide
coalesced loads, b4_seg_scatter writes every record once (16 B x R) -- so write factor = 1 by construction of R, and the
fetch factor follows (MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of a wide coalesced streaming read on gfx950).
Writes gpurun_out/pp4_traffic.json (copied to profiles/r05_pp_block_traffic.json, which bench.py reads)."""
import json
import os
import sys

G = int(os.environ.get("PP_BLOCK", "16"))
N_LIVE, T, F = (int(os.environ.get(k, d)) for k, d in (("PP_N_LIVE", 30_000), ("PP_TRAV", 10), ("PP_FRAMES", 36)))   # (C5: 35000, 20, 16)
d = {k: v for k, v in json.load(open("gpurun_out/pp4_pmc.json")).items() if k.startswith("b4_")}
rec_bytes = d["b4_seg_scatter"]["WRITE_SIZE"] * 1024            # = 16 B x records of the block store (write factor 1)
f_fac = rec_bytes / (d["b4_seg_hist"]["FETCH_SIZE"] * 1024)
fetch = sum(v.get("FETCH_SIZE", 0.0) * 1024 for v in d.values()) * f_fac
write = sum(v.get("WRITE_SIZE", 0.0) * 1024 for v in d.values())
alg = float(os.environ["PP_ALG_BYTES"]) if os.environ.get("PP_ALG_BYTES") else 12.0 * N_LIVE * T * F + 16.0 * N_LIVE   # (other shapes: the probe's own figure)
out = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python tools/pp_block_probe.py --scans %d; "
              "tools/pp4_pmc.sh + tools/pp4_traffic.py" % G + ((" " + os.environ["PROBE_ARGS"]) if os.environ.get("PROBE_ARGS") else ""),
    "block_path": True, "scans_per_launch": G,
    "records_in_block_store": rec_bytes / 16.0,
    "fetch_calibration": {"kernel": "b4_seg_hist reads the block store once (16 B x records, coalesced); records from b4_seg_scatter's WRITE_SIZE",
                          "factor": f_fac},
    "write_calibration": {"factor": 1.0},
    "hbm_bytes_per_launch": fetch + write,
    "hbm_bytes_per_scan": (fetch + write) / G,
    "algorithmic_bytes_per_scan": alg,
    "ratio_to_algorithmic": (fetch + write) / G / alg,
    "per_kernel_bytes_per_scan": {k: {"fetch": v.get("FETCH_SIZE", 0.0) * 1024 * f_fac / G, "write": v.get("WRITE_SIZE", 0.0) * 1024 / G}
                                  for k, v in sorted(d.items())},
}
json.dump(out, open(os.environ.get("PP_TRAFFIC_OUT", "gpurun_out/pp4_traffic.json"), "w"), indent=1)
print("block of %d scans: HBM bytes per scan %.1f MB = %.2f x algorithmic (fetch factor %.3f)" %
      (G, out["hbm_bytes_per_scan"] / 1e6, out["ratio_to_algorithmic"], f_fac))
