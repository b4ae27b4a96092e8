"""RANSAC plane regression z ~ (x, y) with the device doing every O(N) loop.

Behavioural mirror of ``sklearn.linear_model.RANSACRegressor()`` with default
arguments, as the reference calls it (``utils/pointcloud_utils.py:52``):
LinearRegression base model, ``min_samples = 3``, residual threshold =
MAD(z), absolute loss, ``max_trials = 100`` with the dynamic stop
(``stop_probability = 0.99``), accept rule "more inliers, or as many with a
better R^2", final least-squares refit on the consensus set.

Split of work:
  host   - the random triplets (the caller's RandomState is advanced by exactly the
           draws sklearn would consume for the executed trials), the sequential
           accept rule and dynamic trial bound (a few dozen scalar decisions);
  device - MAD threshold (exact float32 medians), the 3-point plane of each
           trial, residual / inlier counting / R^2 sums of a whole batch of
           trials -- all in ONE round trip per batch -- and the final refit
           (float64 normal equations).

Numerical contract: inlier decisions are float32 like sklearn's (pred =
fmaf(y,c1,x*c0)+b); trial planes and the refit are computed in float64 and
rounded to float32, so the plane agrees with sklearn's float32 LAPACK path to
~1e-6 relative (tolerance in tests: 1e-4 relative, as BASELINE.json states).
"""
from __future__ import annotations

import numpy as np
import torch
import numbers

from .. import ops
from .._lib import ModestHipError

_EPSILON = np.spacing(1)
NATIVE_DRIVER = True   # tests switch it off to compare the library's trial loop with the Python statement below


def check_random_state(seed):
    """sklearn.utils.check_random_state: None / np.random -> numpy's global RandomState, an int seeds a fresh one, a
    RandomState is passed through."""
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, numbers.Integral):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError("%r cannot be used to seed a numpy.random.RandomState instance" % seed)


def sample_without_replacement(n_population: int, n_samples: int, random_state) -> np.ndarray:
    """What sklearn.utils.random.sample_without_replacement(..., method="auto") draws from a RandomState
    (sklearn/utils/_random.pyx): for 0.01 < n_samples / n_population < 0.99 the first n_samples entries of
    `permutation(n_population)`; below 0.2 otherwise "tracking selection" -- `randint(n_population)` until the value is
    new; else reservoir sampling.  The library's modest_mt19937_triplets makes the same draws for large populations;
    this is the statement for all of them (tests/test_abi_and_host.py compares values AND generator state with
    sklearn's)."""
    rs = check_random_state(random_state)
    ratio = n_samples / n_population if n_population != 0 else 1.0
    if 0.01 < ratio < 0.99:
        return rs.permutation(n_population)[:n_samples]
    out = np.empty(n_samples, dtype=np.int64)
    if ratio < 0.2:
        seen = set()
        for i in range(n_samples):
            j = int(rs.randint(n_population))
            while j in seen:
                j = int(rs.randint(n_population))
            seen.add(j)
            out[i] = j
        return out
    out[:] = np.arange(n_samples)
    for i in range(n_samples, n_population):
        j = int(rs.randint(0, i + 1))
        if j < n_samples:
            out[j] = i
    return out


def dynamic_max_trials(n_inliers, n_samples, min_samples, probability):
    inlier_ratio = n_inliers / float(n_samples)
    nom = max(_EPSILON, 1 - probability)
    denom = max(_EPSILON, 1 - inlier_ratio ** min_samples)
    if nom == 1:
        return 0
    if denom == 1:
        return float("inf")
    return abs(float(np.ceil(np.log(nom) / np.log(denom))))


def planes_through_triplets(p: np.ndarray) -> np.ndarray:
    """(B,3,3) float32 points -> (B,3) float32 models [c0, c1, b] of the exact-fit
    planes z = c0 x + c1 y + b (float64 arithmetic, centred like a least-squares
    fit; degenerate triplets fall back to the minimum-norm solution)."""
    p = p.astype(np.float64)
    mean = p.mean(axis=1, keepdims=True)
    q = p - mean
    A, z = q[:, :, :2], q[:, :, 2]
    ata = np.einsum("bij,bik->bjk", A, A)
    atz = np.einsum("bij,bi->bj", A, z)
    det = ata[:, 0, 0] * ata[:, 1, 1] - ata[:, 0, 1] * ata[:, 1, 0]
    coef = np.zeros((p.shape[0], 2))
    scale = np.maximum(ata[:, 0, 0] * ata[:, 1, 1], 1e-300)
    ok = np.abs(det) > 1e-12 * scale
    coef[ok, 0] = (atz[ok, 0] * ata[ok, 1, 1] - atz[ok, 1] * ata[ok, 0, 1]) / det[ok]
    coef[ok, 1] = (atz[ok, 1] * ata[ok, 0, 0] - atz[ok, 0] * ata[ok, 0, 1]) / det[ok]
    for b in np.nonzero(~ok)[0]:
        coef[b] = np.linalg.lstsq(A[b], z[b], rcond=None)[0]
    icpt = mean[:, 0, 2] - coef[:, 0] * mean[:, 0, 0] - coef[:, 1] * mean[:, 0, 1]
    return np.concatenate([coef, icpt[:, None]], axis=1).astype(np.float32)


def r2_from_sums(n, sse, sy, syy):
    """R^2 over the inliers from float64 sums (sklearn r2_score semantics for
    the degenerate denominators)."""
    if n < 2:
        return float("nan")
    den = syy - sy * sy / n
    if den <= 0.0:
        return 1.0 if sse == 0.0 else 0.0
    return 1.0 - sse / den


class RansacResult:
    __slots__ = ("coef", "intercept", "n_trials", "n_inliers", "threshold", "triplets", "best_model")


def draw_triplets(rs, n_population, n_trials):
    """`n_trials` consecutive ``sample_without_replacement(n_population, 3, random_state=rs)``
    draws.  For 3/n_population < 0.01 sklearn's 'tracking_selection' is `rs.randint(n)` until
    three distinct values are found; that stream is consumed here in one vectorised call
    (tests/test_abi_and_host.py checks triplets AND generator state against sklearn).
    Returns (triplets (B,3) int64, draws consumed after each trial (B,) or None)."""
    if 3.0 / n_population >= 0.01:    # small populations: sklearn's permutation / reservoir methods (statement above)
        return np.stack([sample_without_replacement(n_population, 3, random_state=rs)
                         for _ in range(n_trials)]), None
    state = rs.get_state()
    draws = rs.randint(n_population, size=3 * n_trials + 16)
    trip = draws[:3 * n_trials].reshape(n_trials, 3)
    clean = (trip[:, 0] != trip[:, 1]) & (trip[:, 0] != trip[:, 2]) & (trip[:, 1] != trip[:, 2])
    if clean.all():
        used = 3 * np.arange(1, n_trials + 1)
    else:                             # a duplicate inside a triplet (probability ~3/n): replay scalar-wise
        rs.set_state(state)
        draws = rs.randint(n_population, size=4 * n_trials + 64)
        out, used, pos = [], [], 0
        for _ in range(n_trials):
            sel = []
            while len(sel) < 3:
                j = int(draws[pos])
                pos += 1
                if j not in sel:
                    sel.append(j)
            out.append(sel)
            used.append(pos)
        trip, used = np.asarray(out), np.asarray(used)
    rs.set_state(state)               # the caller advances the stream by the trials it executes
    return trip, used


def _degenerate_refit(cand, best_model, thr):
    """sklearn's LinearRegression on a degenerate consensus set (fewer than 3 points, or a singular normal
    matrix) returns the minimum-norm least-squares fit; the device refit reports such sets instead of
    dividing by zero.  Scalar work on the host, once in a blue moon."""
    pts = cand.cpu().numpy().astype(np.float32)
    pred = (pts[:, 1] * best_model[1] + pts[:, 0] * best_model[0]) + best_model[2]
    inl = np.abs(pts[:, 2] - pred) <= np.float32(thr)
    X, z = pts[inl, :2].astype(np.float64), pts[inl, 2].astype(np.float64)
    xm, zm = X.mean(axis=0), z.mean()
    coef = np.linalg.lstsq(X - xm, z - zm, rcond=None)[0]
    return np.array([coef[0], coef[1], zm - xm @ coef]), int(inl.sum())


def _ransac_plane_native(cand, rs, thr, max_trials, stop_probability, batch, ctx) -> "RansacResult":
    """The loop below behind one library call (modest_ransac_plane: generator, batches, accept rule,
    refit): same draws, same decisions, no interpreter between the device round trips."""
    status, model64, best_model, trip, n_trials, n_final = ops.ransac_plane_native(
        cand, rs, thr, max_trials, stop_probability, batch, ctx=ctx)
    if status == 1:
        raise ValueError("RANSAC could not find a valid consensus set. All `max_trials` iterations were "
                         "skipped because each randomly chosen sub-sample failed the passing criteria.")
    if status == 2:
        model64, n_final = _degenerate_refit(cand, best_model, thr)
    res = RansacResult()
    res.coef = model64[:2].astype(np.float32)
    res.intercept = np.float32(model64[2])
    res.n_trials = n_trials
    res.n_inliers = n_final
    res.threshold = thr
    res.triplets = trip
    res.best_model = best_model
    return res


def ransac_plane(cand: torch.Tensor, random_state=None, max_trials: int = 100, stop_probability: float = 0.99,
                 batch: int = 48, ctx=None, thr=None) -> RansacResult:
    """cand: (m,3) float32 device tensor of candidate ground points (x, y, z).
    ``thr``: the residual threshold MAD(z) when the caller already has it (ops.mad_threshold_batch)."""
    n_samples = int(cand.shape[0])
    min_samples = 3
    if n_samples < min_samples:
        raise ValueError("`min_samples` may not be larger than number of samples: n_samples = %d." % n_samples)
    rs = check_random_state(random_state)
    # thr None: MAD(z) is computed on the device with the first batch
    if NATIVE_DRIVER and thr is not None and n_samples > 300 and batch <= 64 and isinstance(rs, np.random.RandomState) \
            and rs.get_state()[0] == "MT19937":
        return _ransac_plane_native(cand, rs, thr, max_trials, stop_probability, batch, ctx)

    n_inliers_best, score_best, best_model = 1, -np.inf, None
    n_trials, limit = 0, max_trials
    triplets = []
    while n_trials < limit:
        nb = int(min(batch, limit - n_trials))
        state = rs.get_state()
        trip, consumed = draw_triplets(rs, n_samples, nb)
        # one device round trip: MAD threshold (first batch), exact-fit planes, scores of all nb trials
        thr, models, n_in, sse, sy, syy = ops.ransac_trials(cand, trip, thr, ctx=ctx)
        used = 0
        for k in range(nb):
            if not (n_trials < limit):
                break
            n_trials += 1
            used += 1
            nk = int(n_in[k])
            if nk < n_inliers_best:
                continue
            score = r2_from_sums(nk, sse[k], sy[k], syy[k])
            if nk == n_inliers_best and score < score_best:
                continue
            n_inliers_best, score_best, best_model = nk, score, models[k].copy()
            limit = min(limit, dynamic_max_trials(n_inliers_best, n_samples, min_samples, stop_probability))
        triplets.append(trip[:used])
        # advance the caller's stream by exactly the executed trials (what sklearn would have consumed)
        if consumed is not None:
            if used:
                rs.randint(n_samples, size=int(consumed[used - 1]))
        elif used < nb:
            rs.set_state(state)
            for _ in range(used):
                sample_without_replacement(n_samples, min_samples, random_state=rs)
    if best_model is None:
        raise ValueError("RANSAC could not find a valid consensus set. All `max_trials` iterations were "
                         "skipped because each randomly chosen sub-sample failed the passing criteria.")
    try:
        model64, n_final = ops.ransac_refit(cand, best_model, thr, ctx=ctx)
    except ModestHipError:
        model64, n_final = _degenerate_refit(cand, best_model, thr)
    res = RansacResult()
    res.coef = model64[:2].astype(np.float32)          # LinearRegression on float32 data stores float32
    res.intercept = np.float32(model64[2])
    res.n_trials = n_trials
    res.n_inliers = n_final
    res.threshold = thr
    res.triplets = np.concatenate(triplets) if triplets else np.zeros((0, 3), dtype=np.int64)
    res.best_model = best_model
    return res
