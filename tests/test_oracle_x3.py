"""CPU: the defining properties of the three-plane bf16 split and the six-product sum (oracle/x3_ref.py), which the device
kernels (csrc/gemm_x3.hip, the bf16-pipe forward recurrence of csrc/lstm_persist.hip) are compared with in tests/test_gpu_x3.py."""
import numpy as np

from oracle import x3_ref


def _values(rs, n):
    x = (rs.randn(n) * np.exp(rs.randn(n) * 12)).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.0 + 2.0 ** -23, 257.0 / 256.0]
    x[8:12] = [2.0 ** -100, -(2.0 ** -100) * (1 + 2.0 ** -23), 0.1, 1.0 - 2.0 ** -24]
    return x


def test_split_is_exact_and_planes_are_bf16():
    rs = np.random.RandomState(0)
    x = _values(rs, 20000)
    h = x3_ref.split3(x)
    p = x3_ref.planes(x)
    for hp in h:
        assert int(hp.max()) < 65536
    assert np.array_equal(p[0].astype(np.float64) + p[1].astype(np.float64) + p[2].astype(np.float64), x.astype(np.float64))
    # each term is the nearest bfloat16 of what is left: |rest| <= half an ulp of the term's bf16 grid
    rest = x.astype(np.float64)
    for k in range(3):
        rest = rest - p[k].astype(np.float64)
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(p[k].astype(np.float64)), 1e-300))) - 7)
        assert np.all(np.abs(rest) <= 0.5 * ulp + 0.0)
    # ties go to even: 1 + 2^-8 sits between the bf16 neighbours 1 and 1 + 2^-7
    assert x3_ref.bf16_rne_bits(np.float32([1.0 + 2.0 ** -8]))[0] == 0x3F80
    assert x3_ref.bf16_rne_bits(np.float32([1.0 + 3 * 2.0 ** -8]))[0] == 0x3F82


def test_six_products_error_bound():
    """Dropped terms a2 b3 + a3 b2 + a3 b3: |error| <= 2^-24 sum |a| |b| (measured ~2^-26), far below the sqrt(K) 2^-24 rounding
    noise of an fp32 accumulation -- so the x3 GEMM may be held to the fp32 GEMM's tolerance."""
    rs = np.random.RandomState(1)
    A = (rs.randn(37, 300) * np.exp(rs.randn(37, 300) * 3)).astype(np.float32)
    B = (rs.randn(29, 300) * np.exp(rs.randn(29, 300) * 3)).astype(np.float32)
    exact = A.astype(np.float64) @ B.astype(np.float64).T
    bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    err = np.abs(x3_ref.six_products(A, B) - exact)
    assert np.all(err <= 2.0 ** -24 * bound)
    assert err.max() / bound.max() < 2.0 ** -25
    # with only the first plane of each operand (plain bf16 GEMM) the error is 2^-9-grade: the split is what buys fp32 grade
    p1 = x3_ref.planes(A)[0].astype(np.float64) @ x3_ref.planes(B)[0].astype(np.float64).T
    assert np.abs(p1 - exact).max() / bound.max() > 2.0 ** -14


def test_image_layout():
    """[row group][K block][plane][row][slot][8]: slot of the low half = (row >> 3) & 1; padding is zero; the scaled image equals
    the image of the fp32 product x * scale."""
    rs = np.random.RandomState(2)
    R, K = 70, 37
    x = rs.randn(R, K).astype(np.float32)
    img = x3_ref.image(x)
    assert img.shape == (3, 3, 3, 32, 2, 8) and img.dtype == np.uint16
    h = np.stack(x3_ref.split3(x))
    for r, k in [(0, 0), (7, 15), (8, 0), (8, 8), (31, 36), (32, 16), (69, 36), (45, 9)]:
        half, j = (k % 16) // 8, k % 8
        slot = half ^ ((r % 32 >> 3) & 1)
        for p in range(3):
            assert img[r // 32, k // 16, p, r % 32, slot, j] == h[p, r, k]
    assert not img[2, :, :, 6:, :, :].any()                       # rows 70..95
    flat = img.reshape(3, 3, 3, 32, 16)
    sw = ((np.arange(32) >> 3) & 1).astype(bool)
    kcols = np.where(sw[:, None], np.r_[8:16, 0:8][None, :], np.arange(16)[None, :])   # logical k of each stored position
    assert not flat[:, 2][..., kcols + 32 >= K].any()             # columns 37..47 of the last K block
    s = np.float32(4.0 / 255.0)
    assert np.array_equal(x3_ref.image(x, scale=float(s)), x3_ref.image((x * s).astype(np.float32)))


def test_h2_split_properties():
    """The two-half split of the h2 products (round 5): hi + lo reproduces scale * x to 2^-23 relative while lo is a normal half, the
    kept three products are within 2^-21 sum |a||b| of the exact product, and values beyond the half range clamp instead of
    overflowing."""
    from oracle import x3_ref
    rs = np.random.RandomState(5)
    x = (rs.randn(64, 96) * np.exp(rs.randn(64, 96) * 1.5)).astype(np.float32)
    S = np.float32(2.0 ** (13 - np.ceil(np.log2(np.abs(x).max()))))
    hi, lo = x3_ref.split_h2(x, S)
    v = (x * S).astype(np.float64)
    r = hi.astype(np.float64) + lo.astype(np.float64) - v
    big = np.abs(v) >= 2.0 ** -3
    assert np.all(np.abs(r[big]) <= 2.0 ** -22 * np.abs(v[big])) and np.all(np.abs(r[~big]) <= 2.0 ** -24)
    assert np.isfinite(hi.astype(np.float32)).all() and np.abs(hi.astype(np.float32)).max() < 2.0 ** 14
    A = (rs.randn(40, 300)).astype(np.float32)
    B = (rs.randn(24, 300) * 0.05).astype(np.float32)
    got = x3_ref.three_products(A, B, 2.0 ** 11, 2.0 ** 15)
    exact = A.astype(np.float64) @ B.astype(np.float64).T
    bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    assert np.all(np.abs(got - exact) <= 2.0 ** -21 * bound)
    h, _ = x3_ref.split_h2(np.array([1e9, -1e9, np.float32(70000.0)], dtype=np.float32))
    assert np.all(np.abs(h.astype(np.float32)) == 65504.0)
    assert x3_ref.image_h2(x, S).shape == (2, 6, 2, 32, 2, 8)
