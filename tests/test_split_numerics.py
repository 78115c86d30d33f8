"""The arithmetic behind smp_level_c64_split.hip, emulated in numpy (no GPU): an fp32 operand carried as two f16 halves at a
power-of-two scale (the low half at 2^11 times its value), a product evaluated as ah bh + 2^-11 (ah bl' + al' bh) with fp32 accumulation, is as close to the fp64 product as the
fp32 product is -- over magnitudes from 1e-4 to 1e3 and rows eight decades apart.  (The kernels themselves are held to the SMP
parity bar by tests/test_smp_gpu.py::test_split_operand_products_*.)"""
import numpy as np


LOW = np.float32(2048.0)   # the low half is carried at 2^11 (smp_level_c64_split.hip: split_pair)


def split(x, dt=np.float16, low=LOW):
    h = x.astype(dt).astype(np.float32)
    l = ((x - h).astype(np.float32) * low).astype(dt).astype(np.float32)
    return h, l


def pow2_scale(m):
    """2^k that puts magnitude m into [2^13, 2^14) -- pow2_scale() of the kernel (exponent clamped at 14)."""
    e = (np.maximum(m, 1e-38).astype(np.float32).view(np.uint32) >> 23).astype(np.int64)
    e = np.maximum(e, 14)
    return np.exp2((267 - e) - 127).astype(np.float32)


def test_two_half_products_are_fp32_grade():
    rng = np.random.default_rng(0)
    R, K, N = 2048, 64, 64
    for mag in (1.0, 1e3, 1e-4):
        A = (rng.standard_normal((R, K)) * mag * np.exp(rng.standard_normal((R, 1)) * 4)).astype(np.float32)   # rows decades apart
        W = (rng.uniform(-1, 1, (K, N)) * 0.3).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64)
        f32 = A @ W
        sA = pow2_scale(np.abs(A).max(axis=1, keepdims=True))          # one exponent per row
        sW = pow2_scale(np.abs(W).max())                                # one per weight block
        Ah, Al = split(A * sA)
        Wh, Wl = split(W * sW)
        assert np.abs(Ah).max() < 65504 and np.isfinite(Ah).all()
        got = (((Al @ Wh) + (Ah @ Wl)) / LOW + (Ah @ Wh)) / (sA * sW)

        def row_rel(x):   # per row: relative to the row's own largest output
            return (np.abs(x - ref).max(axis=1) / np.abs(ref).max(axis=1)).max()

        def rms(x):
            return np.sqrt(((x - ref) ** 2).mean() / (ref ** 2).mean())

        assert row_rel(got) <= 2.0 * row_rel(f32) + 1e-7, (mag, row_rel(got), row_rel(f32))
        assert rms(got) <= 1.25 * rms(f32), (mag, rms(got), rms(f32))
        assert row_rel(got) < 2e-6


def test_one_half_is_not_enough():
    """(what the second half buys: a single f16 operand pair is 1e-4, three orders of magnitude off the bar)"""
    rng = np.random.default_rng(1)
    A = rng.standard_normal((256, 64)).astype(np.float32)
    W = rng.uniform(-1, 1, (64, 64)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64)
    Ah, _ = split(A * pow2_scale(np.abs(A).max(axis=1, keepdims=True)))
    Wh, _ = split(W * pow2_scale(np.abs(W).max()))
    got = (Ah @ Wh) / (pow2_scale(np.abs(A).max(axis=1, keepdims=True)) * pow2_scale(np.abs(W).max()))
    assert np.abs(got - ref).max() / np.abs(ref).max() > 1e-5


def test_scaled_low_half_has_no_window():
    """One loud channel 2^24 (and 2^27) times larger than the rest of its row, weights that ignore it in half of the output columns:
    those outputs are made of the small entries alone.  With the low half carried unscaled they keep ~15 bits (1e-4); at 2^11 they are
    as good as the fp32 product."""
    rng = np.random.default_rng(3)
    rows = 1000
    for big in (2.0 ** 24, 2.0 ** 27):
        A = rng.standard_normal((rows, 64)) * np.exp(rng.uniform(-9, 9, (rows, 1)))
        hot = int(rng.integers(64))
        A[:, hot] *= big
        A = A.astype(np.float32)
        W = rng.uniform(-1, 1, (64, 64)).astype(np.float32)
        W[hot, :32] = 0
        ref = A.astype(np.float64) @ W.astype(np.float64)
        sA, sW = pow2_scale(np.abs(A).max(axis=1, keepdims=True)), pow2_scale(np.abs(W).max())

        def run(low):
            Ah, Al = split(A * sA, low=low)
            Wh, Wl = split(W * sW, low=low)
            cross = (Al.astype(np.float64) @ Wh + Ah.astype(np.float64) @ Wl).astype(np.float32) / low
            return ((cross.astype(np.float64) + Ah.astype(np.float64) @ Wh).astype(np.float32) / (sA * sW)).astype(np.float64)

        def err(x):   # per (row, 32-column half), relative to that half's own largest output
            d = np.abs(x - ref).reshape(rows, 2, 32).max(axis=2)
            return float((d / np.abs(ref).reshape(rows, 2, 32).max(axis=2)).max())

        assert err(run(np.float32(1.0))) > 5e-5      # the unscaled low half: the window of rounds 2-3
        assert err(run(LOW)) <= 2e-6, (big, err(run(LOW)))
