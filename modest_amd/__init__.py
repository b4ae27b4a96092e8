"""MI355X-native seed-label hot path of MODEST (pre_compute_pp_score -> generate_mask ->
gen_label_files): Python mirrors of the reference's interfaces over libmodest_hip.so."""
import numpy as _np

# np.percentile on float32 data interpolates in float32 since NumPy 2.0 (it did so in float64
# before); the cluster filter (clustering_utils.py:107-116 of the reference, `percentile_from_
# order_stats` here and in cluster_stats.hip / boxfilter.hip) reproduces the NumPy >= 2 arithmetic
# the fixtures were generated with.  A NumPy 1.x host would decide borderline clusters differently.
if int(_np.__version__.split(".")[0]) < 2:
    raise ImportError("modest_amd reproduces NumPy >= 2 percentile arithmetic (float32 interpolation); "
                      f"found NumPy {_np.__version__}")
