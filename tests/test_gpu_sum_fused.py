"""A Sum of single-profile isotropic kernels in ONE pass of the fp32 matrix-core kernels (round 5; csrc/common.hpp: SumParams,
csrc/dense_mfma.hpp: mfma_sum_term) — the reference evaluates a Sum per pair, every term on the same (x, y)
(src/algebra.jl:27-47, call :36); round 4 ran one full MVM per term.  Checked three ways on the same seeded inputs: the one-pass
kernels against one MVM per term (option "sum_fused" = 0) and both against the fp64 oracle, norm-wise and row-wise at
BASELINE.json's 1e-5 — on the symmetric (upper triangle once) forms, the general form (two point sets, row shards), 2 and 3
terms, every admitted family, alpha / beta and a NaN-filled output.  The packed profile arithmetic of the single-profile
matrix-core kernels (MaternP, RQ, Cauchy, IMQ on register pairs; 6-wave symmetric panels) is held to the same bar."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return np.linalg.norm(b - ref) / np.linalg.norm(ref)


def rowwise(b, ref, absref):
    return float(np.max(np.abs(np.asarray(b, dtype=np.float64) - ref) / absref))


def _sums(cg, o):
    """(name, covgram kernel, oracle terms [(coef, oracle kernel)])"""
    L = cg.Lengthscale
    return [
        ("M2+EQ", 1.5 * L(cg.MaternP(2), 0.7) + 0.5 * L(cg.EQ(), 2.0), [(1.5, o.Kernel(o.MATERNP, p=2, lengthscale=0.7)), (0.5, o.Kernel(o.EQ, lengthscale=2.0))]),
        ("EQ+EQ", L(cg.EQ(), 0.9) + 0.25 * L(cg.EQ(), 2.5), [(1.0, o.Kernel(o.EQ, lengthscale=0.9)), (0.25, o.Kernel(o.EQ, lengthscale=2.5))]),
        ("M1+M3", 0.8 * L(cg.MaternP(1), 1.1) + 1.2 * L(cg.MaternP(3), 0.8), [(0.8, o.Kernel(o.MATERNP, p=1, lengthscale=1.1)), (1.2, o.Kernel(o.MATERNP, p=3, lengthscale=0.8))]),
        ("RQ+Cauchy", L(cg.RQ(1.5), 1.3) + 2.0 * cg.Cauchy(), [(1.0, o.Kernel(o.RQ, param=1.5, lengthscale=1.3)), (2.0, o.Kernel(o.CAUCHY))]),
        ("IMQ+M2+EQ", 0.5 * cg.InverseMultiQuadratic(1.2) + L(cg.MaternP(2), 1.5) + 0.3 * L(cg.EQ(), 0.8),
         [(0.5, o.Kernel(o.IMQ, param=1.2)), (1.0, o.Kernel(o.MATERNP, p=2, lengthscale=1.5)), (0.3, o.Kernel(o.EQ, lengthscale=0.8))]),
        ("EQ+RQ+M1", L(cg.EQ(), 1.4) + 0.7 * L(cg.RQ(0.8), 0.9) + 0.2 * cg.MaternP(1),
         [(1.0, o.Kernel(o.EQ, lengthscale=1.4)), (0.7, o.Kernel(o.RQ, param=0.8, lengthscale=0.9)), (0.2, o.Kernel(o.MATERNP, p=1))]),
    ]


def _oracle_mul(o, terms, Xd, Yd, ad):
    return sum(c * o.mul(None, k, Xd, Yd, ad) for c, k in terms)


def _oracle_abs(o, terms, Xd, Yd, ad):
    return sum(abs(c) * (np.abs(o.matrix(k, Xd, Yd)) @ np.abs(ad)) for c, k in terms)


@pytest.mark.parametrize("n,d", [(1500, 3), (777, 1), (1100, 2), (1300, 5), (1000, 8), (900, 12)])
def test_one_pass_sum_symmetric_matches_termwise_and_oracle(cg, oracle, n, d):
    o = oracle
    rng = np.random.default_rng(900 + n + d)
    Xh = (0.8 * rng.standard_normal((n, d))).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    Xd, ad = Xh.astype(np.float64), ah.astype(np.float64)
    try:
        cg.set_option("mfma_sym", 1)                         # the symmetric forms below their automatic size
        for name, k, terms in _sums(cg, o):
            G = cg.gramian(k, X)
            cg.set_option("sum_fused", 1)                    # two-term Sums too (the automatic rule keeps those on one symmetric MVM per term)
            y = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
            G.mul_(y, a)
            assert cg.get_info("last_sum_fused") == 1 and cg.get_info("last_mfma_sym") == 1 and cg.get_info("last_dense_path") == 2, name
            cg.set_option("sum_fused", 0)
            y0 = torch.empty_like(y); G.mul_(y0, a)
            ref = _oracle_mul(o, terms, Xd, Xd, ad)
            absref = _oracle_abs(o, terms, Xd, Xd, ad)
            b, b0 = y.cpu().numpy(), y0.cpu().numpy()
            assert np.isfinite(b).all(), name
            assert relerr(b, ref) <= 1e-5 and rowwise(b, ref, absref) <= 1e-5, (name, n, d, relerr(b, ref), rowwise(b, ref, absref))
            assert rowwise(b0, ref, absref) <= 1e-5, name
            assert rowwise(b, b0.astype(np.float64), absref) <= 4e-6, (name, rowwise(b, b0.astype(np.float64), absref))   # fused vs one MVM per term (both within 1e-5 of the oracle above)
            cg.set_option("sum_fused", 1)
            y2 = torch.from_numpy(ah[::-1].copy()).cuda()
            G.mul_(y2, a, -0.7, 1.3)
            assert relerr(y2.cpu().numpy(), -0.7 * ref + 1.3 * ad[::-1]) <= 1e-5, name
    finally:
        cg.set_option("mfma_sym", -1); cg.set_option("sum_fused", -1)


@pytest.mark.parametrize("n,m,d", [(700, 1300, 3), (333, 2049, 6), (1200, 640, 2), (500, 900, 16)])
def test_one_pass_sum_two_point_sets_matches_termwise_and_oracle(cg, oracle, n, m, d):
    o = oracle
    rng = np.random.default_rng(1900 + n + d)
    Xh = (0.8 * rng.standard_normal((n, d))).astype(np.float32); Yh = (0.8 * rng.standard_normal((m, d)) + 0.1).astype(np.float32)
    ah = rng.standard_normal(m).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); Y = torch.from_numpy(Yh).cuda(); a = torch.from_numpy(ah).cuda()
    Xd, Yd, ad = Xh.astype(np.float64), Yh.astype(np.float64), ah.astype(np.float64)
    try:
        cg.set_option("dense_variant", 2)                    # matrix cores whatever the routing rule says at this size
        for name, k, terms in _sums(cg, o):
            G = cg.gramian(k, X, Y)
            cg.set_option("sum_fused", -1)                   # the automatic rule: three terms in one pass, two terms one MVM per term (tools/sum_general_ab.py)
            y = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
            G.mul_(y, a)
            assert cg.get_info("last_sum_fused") == (1 if len(terms) >= 3 else 0), name
            cg.set_option("sum_fused", 1)
            y = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
            G.mul_(y, a)
            assert cg.get_info("last_sum_fused") == 1 and cg.get_info("last_dense_path") == 2, name
            cg.set_option("sum_fused", 0)
            y0 = torch.empty_like(y); G.mul_(y0, a)
            ref = _oracle_mul(o, terms, Xd, Yd, ad)
            absref = _oracle_abs(o, terms, Xd, Yd, ad)
            b = y.cpu().numpy()
            assert np.isfinite(b).all(), name
            assert relerr(b, ref) <= 1e-5 and rowwise(b, ref, absref) <= 1e-5, (name, relerr(b, ref), rowwise(b, ref, absref))
            assert rowwise(b, y0.cpu().numpy().astype(np.float64), absref) <= 4e-6, name
            # three right-hand sides through the 4-column instance
            cg.set_option("sum_fused", 1)
            A3 = torch.from_numpy(rng.standard_normal((3, m)).astype(np.float32)).cuda()
            B3 = (G @ A3.T).cpu().numpy()
            for c in range(3):
                refc = _oracle_mul(o, terms, Xd, Yd, A3[c].cpu().numpy().astype(np.float64))
                assert relerr(B3[:, c], refc) <= 1e-5, (name, c)
    finally:
        cg.set_option("dense_variant", 0); cg.set_option("sum_fused", -1)


def test_sums_the_one_pass_kernels_do_not_take_stay_termwise(cg, oracle):
    """A Power wrapper, a product term, a constant term, Exponential (not smooth at 0), fp64, and sum_fused = 0: one MVM per term
    (or the interpreter) as before; the result still matches the oracle."""
    o = oracle
    n, d = 600, 3
    rng = np.random.default_rng(5)
    Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    Xd, ad = Xh.astype(np.float64), ah.astype(np.float64)
    cases = [(cg.EQ() ** 2 + cg.MaternP(2), [(1.0, o.Kernel(o.EQ, power=2)), (1.0, o.Kernel(o.MATERNP, p=2))]),
             (cg.Exp() + cg.EQ(), [(1.0, o.Kernel(o.EXP)), (1.0, o.Kernel(o.EQ))])]
    for k, terms in cases:
        b = (cg.gramian(k, X) @ a).cpu().numpy()
        assert cg.get_info("last_sum_fused") == 0
        assert relerr(b, _oracle_mul(o, terms, Xd, Xd, ad)) <= 1e-5
    Xd64 = torch.from_numpy(Xd).cuda(); a64 = torch.from_numpy(ad).cuda()
    k = cg.MaternP(2) + 0.5 * cg.EQ()
    b = (cg.gramian(k, Xd64) @ a64).cpu().numpy()
    assert cg.get_info("last_sum_fused") == 0
    assert relerr(b, o.mul(None, o.Kernel(o.MATERNP, p=2), Xd, Xd, ad) + 0.5 * o.mul(None, o.Kernel(o.EQ), Xd, Xd, ad)) <= 1e-12


@pytest.mark.parametrize("n,d", [(1400, 3), (1000, 7), (800, 12)])
def test_packed_profiles_on_the_symmetric_and_general_matrix_core_kernels(cg, oracle, n, d):
    """MaternP(1, 2, 3, 4), RQ, Cauchy, IMQ, EQ^2: the tile's profile arithmetic runs on register pairs (v_pk_*_f32) since round 5;
    symmetric 6- / 8-wave panels and the general kernel, both against the oracle row by row."""
    o = oracle
    rng = np.random.default_rng(3000 + n + d)
    Xh = (0.7 * rng.standard_normal((n, d))).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    Xd, ad = Xh.astype(np.float64), ah.astype(np.float64)
    L = cg.Lengthscale
    cases = [("M1", L(cg.MaternP(1), 0.9), o.Kernel(o.MATERNP, p=1, lengthscale=0.9)), ("M2", cg.MaternP(2), o.Kernel(o.MATERNP, p=2)),
             ("M3", L(cg.MaternP(3), 1.2), o.Kernel(o.MATERNP, p=3, lengthscale=1.2)), ("M4", cg.MaternP(4), o.Kernel(o.MATERNP, p=4)),
             ("RQ", L(cg.RQ(2.5), 1.1), o.Kernel(o.RQ, param=2.5, lengthscale=1.1)), ("Cauchy", cg.Cauchy(), o.Kernel(o.CAUCHY)),
             ("IMQ", cg.InverseMultiQuadratic(0.9), o.Kernel(o.IMQ, param=0.9)), ("EQ^2", L(cg.EQ(), 1.5) ** 2, o.Kernel(o.EQ, lengthscale=1.5, power=2))]
    try:
        cg.set_option("dense_variant", 2)
        for name, k, ko in cases:
            ref = o.mul(None, ko, Xd, Xd, ad)
            absref = np.abs(o.matrix(ko, Xd, Xd)) @ np.abs(ad)
            for sym in (1, 0):
                cg.set_option("mfma_sym", sym)
                y = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
                cg.gramian(k, X).mul_(y, a)
                assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_sym") == sym, (name, sym)
                b = y.cpu().numpy()
                assert np.isfinite(b).all(), (name, sym)
                assert relerr(b, ref) <= 1e-5 and rowwise(b, ref, absref) <= 1e-5, (name, sym, relerr(b, ref), rowwise(b, ref, absref))
    finally:
        cg.set_option("dense_variant", 0); cg.set_option("mfma_sym", -1)
