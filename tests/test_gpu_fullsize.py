"""Every BASELINE.json config AT ITS STATED SIZE under the driver's `-m gpu` run: the HIP path through the
C ABI against the oracle on a fixed row subset (a fixed random subset of rows over ALL columns when n is too
large for a full fp64 oracle).  The subset products come from the C restatement (oracle/covgram_oracle.c, strict IEEE build,
fp64, OpenMP) — the numpy oracle needs ~50 s for 512 rows of C3 on 8 cores — which tests/test_oracle.py pins to the numpy one.
Tolerances (BASELINE.json): 1e-5 fp32 (norm-wise, and row-wise as tests/test_gpu_parity.py::rowwise_err), 1e-12 fp64 dense /
gradient, 1e-10 fp64 Toeplitz."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED0 = 0xC0F                    # seed = 0xC0F + config index


def relerr(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return float(np.linalg.norm(b - ref) / np.linalg.norm(ref))


def subset_products(oracle, X, Y, a, rows):
    """fp64 products of the rows `rows` of gramian(EQ, X, Y) with a and with |a| (the row-wise scale), and L_i = -ln max_j k_ij."""
    import c_oracle
    ko = oracle.Kernel(oracle.EQ)
    Xr = X[rows].astype(np.float64); Y64 = Y.astype(np.float64)
    ref = c_oracle.mvm(ko, Xr, Y64, a.astype(np.float64))
    absref = c_oracle.mvm(ko, Xr, Y64, np.abs(a).astype(np.float64))
    L = np.empty(len(rows))
    yy = (Y64 * Y64).sum(1)
    for i0 in range(0, len(rows), 32):
        xb = Xr[i0:i0 + 32]
        d2 = (xb * xb).sum(1)[:, None] + yy[None, :] - 2.0 * xb @ Y64.T
        L[i0:i0 + 32] = np.maximum(d2.min(1), 0.0) / 2.0
    return ref, absref, L


def rowwise(b, ref, absref, L):
    return float(np.max(np.abs(np.asarray(b, np.float64) - ref) / absref / np.maximum(1.0, L / 10.0)))


def test_config2_eq_d3_n131072_f32_general_and_symmetric(cg, oracle):
    """C2: EQ dense Gramian mul!, d = 3, n = 131072, fp32 — the kernel bench.py times (all n*m entries) and the symmetric default."""
    n, d = 131072, 3
    rng = np.random.default_rng(SEED0 + 1)
    Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    rows = np.sort(np.random.default_rng(1).choice(n, 512, replace=False))
    ref, absref, L = subset_products(oracle, Xh, Xh, ah, rows)
    G = cg.gramian(cg.EQ(), torch.from_numpy(Xh).cuda()); a = torch.from_numpy(ah).cuda()
    try:
        for sym in (0, -1):
            cg.set_option("mfma_sym", sym)
            b = (G @ a).cpu().numpy()
            assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_sym") == (0 if sym == 0 else 1)
            # ... and it is the INSTANCE bench.py times: the fp16 two-way split, for sym = 0 dense_mfma_eq_kernel<K2 = 1, RT = 2, WPB = 8, LDS = 1, STAMP = 0, FMT = 1>
            assert cg.get_info("last_mfma_f16") == 1
            assert cg.get_info("last_mfma_instance") == (128101 if sym == 0 else -11), cg.get_info("last_mfma_instance")
            assert np.isfinite(b).all()
            assert relerr(b[rows], ref) <= 1e-5 and rowwise(b[rows], ref, absref, L) <= 1e-5, (sym, relerr(b[rows], ref), rowwise(b[rows], ref, absref, L))
    finally:
        cg.set_option("mfma_sym", -1)


def test_config3_eq_d8_n524288_f32_shard_and_symmetric_partials(cg, oracle):
    """C3: EQ dense Gramian mul!, d = 8, n = 524288, fp32 — what ONE of the 8 ranks computes: (a) its row shard, 65536 rows x
    all 524288 columns (rank 3's, so that the shard offset matters), and (b) the symmetric form: the partial products of the 8
    ranks' cyclic panels, summed here on one GPU in place of the all-reduce."""
    n, d, world = 524288, 8, 8
    rng = np.random.default_rng(SEED0 + 2)
    Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    per = n // world
    lo, hi = 3 * per, 4 * per
    rows = np.sort(np.random.default_rng(2).choice(per, 512, replace=False)) + lo
    ref, absref, L = subset_products(oracle, Xh, Xh, ah, rows)
    G = cg.gramian(cg.EQ(), X[lo:hi], X)
    b = (G @ a).cpu().numpy()
    assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_sym") == 0 and b.shape == (per,)
    # the kernel bench.py's C3_shard line times: fp16 split, d = 8 -> two MFMAs per tile: dense_mfma_eq_kernel<K2 = 2, RT = 2, WPB = 8, LDS = 1, 0, FMT = 1>
    assert cg.get_info("last_mfma_f16") == 1 and cg.get_info("last_mfma_instance") == 228101, cg.get_info("last_mfma_instance")
    assert np.isfinite(b).all()
    assert relerr(b[rows - lo], ref) <= 1e-5 and rowwise(b[rows - lo], ref, absref, L) <= 1e-5, (relerr(b[rows - lo], ref), rowwise(b[rows - lo], ref, absref, L))
    # (b) symmetric partials of all 8 ranks
    Gf = cg.gramian(cg.EQ(), X)
    assert Gf.sym_partial_supported()
    tot = torch.zeros(n, dtype=torch.float32, device="cuda"); part = torch.empty_like(tot)
    for r in range(world):
        Gf.sym_partial_(part, a, r, world); tot += part
    assert cg.get_info("last_mfma_f16") == 1 and cg.get_info("last_mfma_instance") == -21     # the symmetric kernel, K2 = 2, fp16 split
    bs = tot.cpu().numpy()
    assert np.isfinite(bs).all()
    assert relerr(bs[rows], ref) <= 1e-5 and rowwise(bs[rows], ref, absref, L) <= 1e-5, (relerr(bs[rows], ref), rowwise(bs[rows], ref, absref, L))
    # the shard and the summed partials are two evaluations of the same rows
    assert relerr(bs[lo:hi], b) <= 5e-6


def test_config4_gradient_eq_d32_n16384_f64(cg, oracle):
    """C4: GradientKernel(EQ) mul!, d = 32, n = 16384, fp64 (a in R^{n d}), 128 block rows against the C oracle."""
    import c_oracle
    n, d = 16384, 32
    rng = np.random.default_rng(SEED0 + 3)
    Xh = rng.standard_normal((n, d)); ah = rng.standard_normal(n * d)
    K = cg.gramian(cg.GradientKernel(cg.EQ()), torch.from_numpy(Xh).cuda())
    y = (K @ torch.from_numpy(ah).cuda()).cpu().numpy()
    rows = np.sort(np.random.default_rng(3).choice(n, 128, replace=False))
    ref = c_oracle.grad_mvm(oracle.Kernel(oracle.EQ), Xh[rows], Xh, ah)
    assert relerr(y.reshape(n, d)[rows].reshape(-1), ref) <= 1e-12
    # block-wise as well: no single block row off
    e = np.linalg.norm(y.reshape(n, d)[rows] - ref.reshape(-1, d), axis=1) / np.linalg.norm(ref.reshape(-1, d), axis=1)
    assert e.max() <= 1e-11, e.max()


def test_config5_exponential_toeplitz_n4194304_f64(cg, oracle):
    """C5: Exponential on range(-1, 1, length = 2^22), fp64: the Toeplitz MVM (embedding N = 2^23) against numpy's FFT and
    against 64 explicit dense rows."""
    n = 1 << 22
    T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n))
    ah = np.random.default_rng(SEED0 + 4).standard_normal(n)
    y = (T @ torch.from_numpy(ah).cuda()).cpu().numpy()
    vc = T.vc.cpu().numpy()
    xs = oracle.srange_points(oracle.srange(-1, 1, n)).reshape(-1)
    assert relerr(vc[:4096], np.exp(-np.abs(xs[:4096] - xs[0]))) <= 1e-14           # k.(x[1], x), src/gramian.jl:174
    assert relerr(y, oracle.toeplitz_mul(None, vc, None, ah)) <= 1e-10
    idx = np.random.default_rng(5).choice(n, 64, replace=False)
    direct = np.array([np.dot(vc[np.abs(i - np.arange(n))], ah) for i in idx])
    assert relerr(y[idx], direct) <= 1e-10


def test_fp64_symmetric_direct_kernel_at_readme_and_large_sizes(cg, oracle):
    """The reference README's own dense case (MaternP(2), d = 3, n = 16384, Float64: README.md:26-38) and a larger one (d = 8, n = 40000,
    not a multiple of the 64-row blocks) as gramian(k, x) * a — the library's default path there is the fp64 symmetric direct-difference
    kernel (upper triangle once) — against the C oracle on a fixed row subset over ALL columns, norm-wise and row by row, and against
    the all-entries kernel on every row."""
    import c_oracle
    for idx, (n, d, scale) in enumerate(((16384, 3, 1.0), (40000, 8, 0.35))):
        rng = np.random.default_rng(SEED0 + 40 + idx)
        X = rng.standard_normal((n, d)) * scale; a = rng.standard_normal(n)
        Xd = torch.from_numpy(X).cuda(); ad = torch.from_numpy(a).cuda()
        G = cg.gramian(cg.MaternP(2), Xd)
        b = (G @ ad)
        assert cg.get_info("last_dense_sym") == 1 and cg.get_info("last_dense_path") == 1
        rows = np.sort(np.random.default_rng(3).choice(n, 512, replace=False))
        rows[:3] = (0, n // 2, n - 1)                     # first / middle / last row block (only column sums / both / only row sums)
        ko = oracle.Kernel(oracle.MATERNP, p=2)
        ref = c_oracle.mvm(ko, X[rows], X, a)
        absref = c_oracle.mvm(ko, X[rows], X, np.abs(a))
        got = b.cpu().numpy()
        assert relerr(got[rows], ref) <= 1e-12, relerr(got[rows], ref)
        assert float(np.max(np.abs(got[rows] - ref) / absref)) <= 1e-12
        try:
            cg.set_option("dense_sym", 0)
            ball = (G @ ad).cpu().numpy()
            assert cg.get_info("last_dense_sym") == 0
        finally:
            cg.set_option("dense_sym", -1)
        assert relerr(got, ball) <= 1e-13


def test_matern_class_gramians_at_contract_size_f32(cg, oracle):
    """gramian(k, x), d = 3, n = 131072, fp32 (the contract cloud) for the Matérn-class profiles the round-5 kernels serve — MaternP(2), MaternP(1), RQ,
    Cauchy on the two-row-tile symmetric matrix-core kernel (csrc/dense_mfma_sym2.hpp), the composite of bench.py's F2 line one MVM per term, a Sum of
    three in one pass — against 256 fp64 rows of the C oracle, norm-wise and row-wise at 1e-5; the instance that ran is pinned."""
    import c_oracle
    o = oracle
    n, d = 131072, 3
    rng = np.random.default_rng(SEED0 + 1)
    Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); y = torch.empty_like(a)
    rows = np.sort(np.random.default_rng(11).choice(n, 256, replace=False))
    Xr = Xh[rows].astype(np.float64); Xd = Xh.astype(np.float64); ad = ah.astype(np.float64)
    L = cg.Lengthscale
    cases = [("MaternP(2)", cg.MaternP(2), [(1.0, o.Kernel(o.MATERNP, p=2))], 2), ("MaternP(1)", cg.MaternP(1), [(1.0, o.Kernel(o.MATERNP, p=1))], 2),
             ("RQ(1.5)", cg.RQ(1.5), [(1.0, o.Kernel(o.RQ, param=1.5))], 2), ("Cauchy", cg.Cauchy(), [(1.0, o.Kernel(o.CAUCHY))], 2),
             ("F2", 1.5 * L(cg.MaternP(2), 0.7) + 0.5 * L(cg.EQ(), 2.0), [(1.5, o.Kernel(o.MATERNP, p=2, lengthscale=0.7)), (0.5, o.Kernel(o.EQ, lengthscale=2.0))], None),
             ("three", L(cg.EQ(), 1.4) + 0.7 * L(cg.RQ(0.8), 0.9) + 0.2 * cg.MaternP(1),
              [(1.0, o.Kernel(o.EQ, lengthscale=1.4)), (0.7, o.Kernel(o.RQ, param=0.8, lengthscale=0.9)), (0.2, o.Kernel(o.MATERNP, p=1))], 1)]
    for name, k, terms, rt in cases:
        G = cg.gramian(k, X)
        G.mul_(y, a)
        assert cg.get_info("last_mfma_sym") == 1, name
        if rt is not None: assert cg.get_info("last_mfma_sym_rt") == rt, (name, cg.get_info("last_mfma_sym_rt"))
        if name == "three": assert cg.get_info("last_sum_fused") == 1
        if name == "F2": assert cg.get_info("last_sum_fused") == 0
        ref = sum(c * c_oracle.mvm(kk, Xr, Xd, ad) for c, kk in terms)
        absref = sum(abs(c) * c_oracle.mvm(kk, Xr, Xd, np.abs(ad)) for c, kk in terms)
        got = y.cpu().numpy()[rows].astype(np.float64)
        assert relerr(got, ref) <= 1e-5, (name, relerr(got, ref))
        assert float(np.max(np.abs(got - ref) / absref)) <= 1e-5, (name, float(np.max(np.abs(got - ref) / absref)))
