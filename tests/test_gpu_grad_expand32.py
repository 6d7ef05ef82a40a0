"""fp32 GradientKernel / ValueGradientKernel Gramians in the expanded form (round 5; csrc/grad_mvm.hpp "Expanded form"): |x - y|^2 and
c2 (x - y) from cached norms and column scalars, 4 instead of 5-6 VALU instructions per dimension and pair, inside gamma^2 R^2 <= 128.
The reference's block is src/gradient.jl:86-92 in the data's own element type (src/gramian.jl:27-33); checked against the fp64 oracle
norm-wise and per block row at BASELINE.json's fp32 tolerance 1e-5, at the C4 shape's cloud (d = 32, x ~ N(0, I): R^2 ~ 90) and AT the
gate, against the direct-difference kernel (option grad_expand = 0), with one and two right-hand sides, alpha / beta, value-gradient blocks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return float(np.linalg.norm(b - ref) / np.linalg.norm(ref))


def blockrows(b, ref, d):
    e = np.linalg.norm(np.asarray(b, np.float64).reshape(-1, d) - ref.reshape(-1, d), axis=1) / np.linalg.norm(ref.reshape(-1, d), axis=1)
    return float(e.max())


@pytest.mark.parametrize("n,d,scale", [(1500, 32, 1.0), (2000, 8, 1.0), (1200, 16, 1.0), (900, 12, 2.9), (1000, 48, 0.8)])
def test_fp32_gradient_expanded_form_matches_the_oracle(cg, oracle, n, d, scale):
    o = oracle
    rng = np.random.default_rng(7000 + n + d)
    Xh = (scale * rng.standard_normal((n, d))).astype(np.float32); ah = rng.standard_normal(n * d).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    Xd, ad = Xh.astype(np.float64), ah.astype(np.float64)
    for name, k, ko in (("EQ", cg.EQ(), o.Kernel(o.EQ)), ("MaternP2", cg.Lengthscale(cg.MaternP(2), 1.5), o.Kernel(o.MATERNP, p=2, lengthscale=1.5)),
                        ("RQ", cg.RQ(1.5), o.Kernel(o.RQ, param=1.5)), ("Cauchy", cg.Lengthscale(cg.Cauchy(), 2.0), o.Kernel(o.CAUCHY, lengthscale=2.0))):
        ref = o.grad_mul(None, ko, Xd, Xd, ad)
        K = cg.gramian(cg.GradientKernel(k), X)
        try:
            cg.set_option("grad_expand", 1)
            y = torch.full((n * d,), float("nan"), dtype=torch.float32, device="cuda")
            K.mul_(y, a)
            assert cg.get_info("last_grad_expand") == 1, name
            cg.set_option("grad_expand", 0)
            y0 = torch.empty_like(y); K.mul_(y0, a)
            assert cg.get_info("last_grad_expand") == 0
        finally:
            cg.set_option("grad_expand", -1)
        b, b0 = y.cpu().numpy(), y0.cpu().numpy()
        assert np.isfinite(b).all(), name
        assert rel(b, ref) <= 1e-5 and blockrows(b, ref, d) <= 1e-4, (name, n, d, rel(b, ref), blockrows(b, ref, d), rel(b0, ref), blockrows(b0, ref, d))
        assert rel(b0, ref) <= 1e-5, name


def test_fp32_gradient_expanded_default_rule_alpha_beta_two_columns_and_value_gradient(cg, oracle):
    """The automatic rule takes the form at the C4 cloud (d = 32, R^2 ~ 90 <= 128) and leaves a wide cloud (R^2 > 128) and the profiles that are
    singular at 0 on direct differences; alpha / beta, a two-column right-hand side (one pass, two accumulators) and ValueGradientKernel blocks."""
    o = oracle
    n, d = 1100, 32
    rng = np.random.default_rng(81)
    Xh = rng.standard_normal((n, d)).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); Xd = Xh.astype(np.float64)
    K = cg.gramian(cg.GradientKernel(cg.EQ()), X)
    A = rng.standard_normal((n * d, 2)).astype(np.float32)
    Y0 = rng.standard_normal((n * d, 2)).astype(np.float32)
    Yt = torch.from_numpy(Y0.copy()).cuda()
    K.mul_(Yt, torch.from_numpy(A).cuda(), -0.7, 1.3)
    assert cg.get_info("last_grad_expand") == 1
    for c in range(2):
        ref = -0.7 * o.grad_mul(None, o.Kernel(o.EQ), Xd, Xd, A[:, c].astype(np.float64)) + 1.3 * Y0[:, c].astype(np.float64)
        assert rel(Yt.cpu().numpy()[:, c], ref) <= 1e-5, c
    # a cloud beyond the gate, and Exponential (NaN diagonal blocks by the reference's own arithmetic): direct differences
    Xw = torch.from_numpy((2.5 * Xh)).cuda()
    (cg.gramian(cg.GradientKernel(cg.EQ()), Xw) @ torch.from_numpy(A[:, 0].copy()).cuda())
    assert cg.get_info("last_grad_expand") == 0
    (cg.gramian(cg.GradientKernel(cg.Exp()), X) @ torch.from_numpy(A[:, 0].copy()).cuda())
    assert cg.get_info("last_grad_expand") == 0
    # value-gradient blocks of d + 1
    av = rng.standard_normal(n * (d + 1)).astype(np.float32)
    V = cg.gramian(cg.ValueGradientKernel(cg.MaternP(2)), X)
    bv = (V @ torch.from_numpy(av).cuda()).cpu().numpy()
    assert cg.get_info("last_grad_expand") == 1
    refv = o.valgrad_mul(None, o.Kernel(o.MATERNP, p=2), Xd, Xd, av.astype(np.float64))
    assert rel(bv, refv) <= 1e-5
