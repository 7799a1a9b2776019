"""The arithmetic claims behind the split-bf16 fp32 GEMM (argsim_amd/csrc/gemm_f32s.hip), checked on the CPU in
numpy: the 3-way RNE split is exact to 2^-26, every bf16 x bf16 product is exact in fp32, the six kept partial
products reproduce a*b to 2^-27 rms (2^-23 worst case, zero mean), and a long dot product accumulated in fp32 from them is as close
to the float64 result as the plain fp32 dot product is."""
import numpy as np


def bf16_rne(x):
    """round-to-nearest-even of float32 to bfloat16, returned as float32 (what v_cvt_pk_bf16_f32 does)"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x):
    hi = bf16_rne(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16_rne(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = bf16_rne(r2)
    return hi, mid, lo, r1, r2


def test_split_is_exact_to_2_pow_minus_26():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200_000) * np.exp(4 * rng.standard_normal(200_000))).astype(np.float32)
    hi, mid, lo, r1, r2 = split3(x)
    # both residual subtractions are exact in fp32 (checked in float64)
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - hi.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - mid.astype(np.float64))
    err = np.abs(x.astype(np.float64) - (hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)))
    assert (err <= 2.0 ** -26 * np.abs(x)).all()
    # the pieces shrink by 2^-8 each (RNE): the weights quoted in the kernel's header
    nz = x != 0
    assert (np.abs(mid[nz]) <= 2.0 ** -8 * np.abs(x[nz]) * 1.0001).all() and (np.abs(lo[nz]) <= 2.0 ** -16 * np.abs(x[nz]) * 1.0001).all()


def test_bf16_products_are_exact_in_fp32_and_six_terms_suffice():
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(100_000) * np.exp(rng.standard_normal(100_000))).astype(np.float32)
    b = (rng.standard_normal(100_000) * np.exp(rng.standard_normal(100_000))).astype(np.float32)
    A, B = split3(a)[:3], split3(b)[:3]
    for p in A:
        for q in B:                      # 8-bit x 8-bit significands: the fp32 product carries no rounding
            assert np.array_equal((p * q).astype(np.float64), p.astype(np.float64) * q.astype(np.float64))
    kept = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]            # lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
    s = sum(A[i].astype(np.float64) * B[j].astype(np.float64) for i, j in kept)
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(s - exact) / np.abs(exact)
    # dropped terms (mid*lo, lo*mid, lo*lo): <= 2^-23 in the worst case (|mid| <= 2^-8, |lo| <= 2^-16 of the operand),
    # ~2^-27 rms and zero mean -- below the rounding of the fp32 accumulation they are added into (2^-24 per add)
    assert rel.max() <= 2.0 ** -23 and np.sqrt((rel ** 2).mean()) <= 2.0 ** -26
    assert abs(((s - exact) / np.abs(exact)).mean()) <= 2.0 ** -30


def test_fp32_accumulated_dot_products_are_as_accurate_as_plain_fp32():
    rng = np.random.default_rng(2)
    K, N = 4096, 64
    a = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal((N, K)).astype(np.float32)
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    scale = (np.abs(a).astype(np.float64) * np.abs(b)).sum(1)
    plain = np.zeros(N, np.float32)
    split = np.zeros(N, np.float32)
    A, B = split3(a)[:3], split3(b)[:3]
    kept = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]
    for k in range(K):                   # sequential fp32 accumulation, one K step at a time
        plain = (plain + a[:, k] * b[:, k]).astype(np.float32)
        for i, j in kept:
            split = (split + A[i][:, k] * B[j][:, k]).astype(np.float32)
    e_plain = np.abs(plain - ref) / scale
    e_split = np.abs(split - ref) / scale
    assert e_split.max() <= 2.0 * max(e_plain.max(), 2.0 ** -24) and e_split.mean() <= 2.0 * e_plain.mean()
