"""NumPy model of the (hi, lo) f16 operand pairs of the matrix-core kernels (irbfn_amd/csrc/f16_split.h): what
precision a pair carries across the dynamic range of a weight column / cotangent batch, for the scheme that
ships (lo pre-scaled by 2^11, own accumulator) and for the round-1 scheme (lo unscaled) that it replaced.
The GPU counterpart is tests/test_gpu_f16.py::test_forward_f16_ill_conditioned_columns."""
import numpy as np


def split_static(v):
    """f16_split.h::split_static_f16 -- |v| <= 1 -> (hi, lo) with 2^15 v = hi + 2^-11 lo."""
    w = (v.astype(np.float32) * np.float32(32768.0)).astype(np.float32)
    hi = w.astype(np.float16)
    lo = ((w - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return hi, lo


def split_legacy(v):
    hi = v.astype(np.float32).astype(np.float16)
    lo = (v.astype(np.float32) - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def split_phi(phi):
    """split_pair_f16<3>: P = 2^14 phi -> hi = top 11 bits, lo = 2^11 (P - hi)."""
    P = (phi.astype(np.float32) * np.float32(16384.0)).astype(np.float32)
    h = (P.view(np.uint32) & np.uint32(0xFFFFE000)).view(np.float32)
    lo = ((P - h) * np.float32(2048.0)).astype(np.float16)
    return h.astype(np.float16), lo


def test_static_pair_keeps_22_bits_over_28_binades():
    rng = np.random.default_rng(0)
    for e in range(0, 29):
        v = (rng.uniform(0.5, 1.0, size=4096) * 2.0 ** -e * rng.choice([-1, 1], size=4096)).astype(np.float32)
        hi, lo = split_static(v)
        back = (hi.astype(np.float64) + lo.astype(np.float64) / 2048.0) / 32768.0
        rel = np.abs(back - v.astype(np.float64)) / np.abs(v)
        assert rel.max() <= 2.0 ** -21, (e, rel.max())
    # the round-1 pair degrades one bit per binade below 2^-3 (3.1e-5 at 2^-10)
    v = (rng.uniform(0.5, 1.0, size=4096) * 2.0 ** -10).astype(np.float32)
    hi, lo = split_legacy(v)
    rel = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - v) / v
    assert rel.max() > 1e-5


def test_phi_pair_keeps_21_bits_down_to_2pow_minus_28():
    rng = np.random.default_rng(1)
    for e in range(0, 28):
        phi = (rng.uniform(0.5, 1.0, size=4096) * 2.0 ** -e).astype(np.float32)
        hi, lo = split_phi(phi)
        back = (hi.astype(np.float64) + lo.astype(np.float64) / 2048.0) / 16384.0
        rel = np.abs(back - phi.astype(np.float64)) / phi
        assert rel.max() <= 2.0 ** -20, (e, rel.max())
        assert np.isfinite(lo.astype(np.float32)).all() and np.abs(lo.astype(np.float32)).max() < 2.0 ** 15


def test_reduction_model_on_the_outlier_column():
    """One |W| = 1e4 outlier on a centre no query sees, bulk O(1): sum_k phi_k W_k through the pairs, two
    accumulators, against float64 -- 1e-6 relative with the scaled lo halves, > 1e-5 with the round-1 pairs."""
    rng = np.random.default_rng(2)
    K, B = 512, 256
    phi = rng.uniform(0.0, 1.0, size=(B, K)) ** 4
    phi[:, 3] = 0.0
    W = np.abs(rng.normal(size=K)) + 0.05
    W[3] = 1.0e4
    s = 2.0 ** np.ceil(np.log2(W.max() * (1 + 1e-9)))
    ref = phi @ W
    ph, pl = split_phi(phi.astype(np.float32))
    wh, wl = split_static((W / s).astype(np.float32))
    ph, pl, wh, wl = (a.astype(np.float64) for a in (ph, pl, wh, wl))
    a1, a2 = ph @ wh, pl @ wh + ph @ wl
    new = (a1 + a2 / 2048.0) * s / (16384.0 * 32768.0)
    assert (np.abs(new - ref) / ref).max() <= 1e-6
    lh, ll = split_legacy((W / s).astype(np.float32))
    P = phi.astype(np.float32) * np.float32(16384.0)
    h0 = (P.view(np.uint32) & np.uint32(0xFFFFE000)).view(np.float32)
    l0 = (P - h0).astype(np.float16).astype(np.float64)
    h0 = h0.astype(np.float64)
    old = (h0 @ lh.astype(np.float64) + l0 @ lh.astype(np.float64) + h0 @ ll.astype(np.float64)) * s / 16384.0
    assert (np.abs(old - ref) / ref).max() > 1e-5
