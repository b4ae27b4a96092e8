"""The generator algorithm of the device-side RANSAC loops (csrc/plane.hip: mt_refill / mt_advance / rsd_draw), restated in
numpy and pinned to numpy's RandomState and to sklearn's sample_without_replacement on the CPU: the three-sweep parallel
regeneration of the 624 state words, the position arithmetic that advances a state by a number of consumed words, and the
tempered / masked / filtered value stream from which triplet k is values 3k..3k+2 (or, when a triplet repeats an index, the
sequential walk).  The kernels themselves are compared with the host loop on the GPU (tests/test_gpu_parity_r4.py)."""
import numpy as np
import pytest

U, L, A = np.uint32(0x80000000), np.uint32(0x7fffffff), np.uint32(0x9908b0df)


def refill_parallel(key):
    """mt_refill: three sweeps of independent words (reads old [kk+1] and old / new [kk+397 mod 624]), then the last word"""
    key = key.copy()
    for lo, hi, src in ((0, 227, 397), (227, 454, -227), (454, 623, -227)):
        kk = np.arange(lo, hi)
        y = (key[kk] & U) | (key[kk + 1] & L)
        v = key[kk + src] ^ (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), A, np.uint32(0))
        key[kk] = v                      # (all reads of the sweep happen before its writes: numpy evaluates the right side first)
    y = (key[623] & U) | (key[0] & L)
    key[623] = key[396] ^ (y >> np.uint32(1)) ^ (A if y & np.uint32(1) else np.uint32(0))
    return key


def temper(y):
    y = y ^ (y >> np.uint32(11))
    y = y ^ ((y << np.uint32(7)) & np.uint32(0x9d2c5680))
    y = y ^ ((y << np.uint32(15)) & np.uint32(0xefc60000))
    return y ^ (y >> np.uint32(18))


def advance(key, pos, words):
    """mt_advance: the state `words` draws further (pos = 624 means: regenerate before the next draw)"""
    total = pos + words
    r = (total - 1) // 624 if total >= 1 else 0
    for _ in range(r):
        key = refill_parallel(key)
    return key, total - 624 * r


def draw(key, pos, n, K):
    """rsd_draw: K triplets of distinct indices < n, and the words consumed up to each"""
    rng = np.uint32(n - 1)
    mask = rng
    for s in (1, 2, 4, 8, 16):
        mask = mask | (mask >> np.uint32(s))
    vals, vpos, consumed = [], [], 0
    need = 3 * K
    while True:
        while len(vals) < need:
            if pos >= 624:
                key, pos = refill_parallel(key), 0
            m = min(256, 624 - pos)
            v = temper(key[pos:pos + m]) & mask
            ok = v <= rng
            vals += v[ok].tolist()
            vpos += (consumed + np.nonzero(ok)[0] + 1).tolist()
            pos += m
            consumed += m
        a = np.array(vals[:3 * K]).reshape(K, 3)
        if not ((a[:, 0] == a[:, 1]) | (a[:, 0] == a[:, 2]) | (a[:, 1] == a[:, 2])).any() and need == 3 * K:
            return a, np.array(vpos)[2:3 * K:3]
        trip, used, cur = [], [], []
        for i, j in enumerate(vals):
            if j in cur:
                continue
            cur.append(j)
            if len(cur) == 3:
                trip.append(cur)
                used.append(vpos[i])
                cur = []
                if len(trip) == K:
                    return np.array(trip), np.array(used)
        need = len(vals) + 3 * (K - len(trip))


def test_parallel_refill_and_advance_equal_numpy():
    for seed in (0, 1, 12345):
        rs = np.random.RandomState(seed)
        _, key, pos, *_ = rs.get_state()
        key = key.astype(np.uint32)
        for words in (0, 1, 5, 623, 624, 625, 1300, 3000):
            ref = np.random.RandomState(seed)
            if words:
                ref.randint(0, 2 ** 32, size=words, dtype=np.uint64)      # one 32-bit word per draw (range 2^32: no rejection)
            k2, p2 = advance(key, int(pos), words)
            rk, rp = ref.get_state()[1].astype(np.uint32), int(ref.get_state()[2])
            # the same generator up to the representation of "at the end of the block": compare the NEXT outputs
            a = np.random.RandomState()
            a.set_state(("MT19937", k2, p2, 0, 0.0))
            b = np.random.RandomState()
            b.set_state(("MT19937", rk, rp, 0, 0.0))
            assert np.array_equal(a.randint(0, 2 ** 31, size=700), b.randint(0, 2 ** 31, size=700)), (seed, words)


@pytest.mark.parametrize("n", [301, 337, 1000, 15000, 70000])
def test_draw_equals_sklearn_sample_without_replacement(n):
    from sklearn.utils.random import sample_without_replacement
    K = 100
    for seed in range(6):
        rs = np.random.RandomState(seed)
        if seed % 2:
            rs.randint(0, 10, size=617)        # start close to the end of a block of 624 words
        _, key, pos, *_ = rs.get_state()
        trip, used = draw(key.astype(np.uint32), int(pos), n, K)
        ref = np.array([sample_without_replacement(n, 3, random_state=rs) for _ in range(K)])
        assert np.array_equal(trip, ref), (n, seed)
        # the generator behind the K triplets = the start state advanced by the words the last one consumed
        k2, p2 = advance(key.astype(np.uint32), int(pos), int(used[-1]))
        a = np.random.RandomState()
        a.set_state(("MT19937", k2, p2, 0, 0.0))
        assert np.array_equal(a.randint(0, 2 ** 31, size=50), rs.randint(0, 2 ** 31, size=50)), (n, seed)
