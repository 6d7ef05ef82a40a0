"""Round 4: the fp64 expanded-form gradient / value-gradient MVM with the column records in VECTOR registers (csrc/grad_bcast.hpp:
v_fmac_f64_dpp row_newbcast, counted vector loads) — the block mul! of src/gradient.jl:86-92 / :319-351 under blockmul!
(src/gramian.jl:241-253).  Against the fp64 oracle at BASELINE's 1e-12 and against the scalar-stream kernel it replaces (same arithmetic
up to the order of two partial sums), every padded dimension it is compiled for, both workgroup shapes, ragged sizes, two point sets,
alpha / beta, a translated cloud, and the automatic rule (from padded d = 24)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return np.linalg.norm(b - ref) / np.linalg.norm(ref)


def _kernels(cg, o):
    return [(cg.EQ(), o.Kernel(o.EQ)), (cg.Lengthscale(cg.MaternP(2), 1.7), o.Kernel(o.MATERNP, p=2, lengthscale=1.7)),
            (1.5 * cg.RQ(2.0), o.Kernel(o.RQ, param=2.0, scale=1.5)), (cg.Cauchy(), o.Kernel(o.CAUCHY)),
            (cg.Lengthscale(cg.MaternP(3), 2.0), o.Kernel(o.MATERNP, p=3, lengthscale=2.0)), (cg.InverseMultiQuadratic(0.9), o.Kernel(o.IMQ, param=0.9))]


@pytest.mark.parametrize("d", [7, 12, 16, 23, 32, 41, 48])
def test_broadcast_gradient_kernel_matches_the_oracle(cg, oracle, d):
    o = oracle
    rng = np.random.default_rng(1200 + d)
    n, m = 333, 211
    X = rng.standard_normal((n, d)) / np.sqrt(d) * 2.0; Y = rng.standard_normal((m, d)) / np.sqrt(d) * 2.0 + 0.1
    a = rng.standard_normal(m * d); y0 = rng.standard_normal(n * d)
    av = rng.standard_normal(m * (d + 1)); yv0 = rng.standard_normal(n * (d + 1))
    Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    try:
        cg.set_option("grad_expand", 1)
        for k, ko in _kernels(cg, o):
            G = cg.gramian(cg.GradientKernel(k), Xd, Yd)
            Gv = cg.gramian(cg.ValueGradientKernel(k), Xd, Yd)
            ref = o.grad_mul(y0, ko, X, Y, a, 0.7, -1.1)
            refv = o.valgrad_mul(yv0, ko, X, Y, av, -0.4, 0.9)
            outs, outv = {}, {}
            for bc in (0, 1, 4):
                cg.set_option("grad_bcast", bc)
                yd = torch.from_numpy(y0.copy()).cuda(); cg.mul_(yd, G, torch.from_numpy(a).cuda(), 0.7, -1.1)
                assert cg.get_info("last_grad_expand") == 1 and cg.get_info("last_grad_bcast") == bc, (d, bc, cg.get_info("last_grad_bcast"))
                outs[bc] = yd.cpu().numpy()
                assert relerr(outs[bc], ref) <= 1e-12, (type(k).__name__, d, bc, relerr(outs[bc], ref))
                yv = torch.from_numpy(yv0.copy()).cuda(); cg.mul_(yv, Gv, torch.from_numpy(av).cuda(), -0.4, 0.9)
                assert cg.get_info("last_grad_bcast") == bc
                outv[bc] = yv.cpu().numpy()
                assert relerr(outv[bc], refv) <= 1e-12, (type(k).__name__, d, bc, relerr(outv[bc], refv))
            assert relerr(outs[1], outs[0]) <= 1e-13 and relerr(outs[4], outs[0]) <= 1e-13 and relerr(outv[4], outv[0]) <= 1e-13
        # the automatic rule: broadcast kernel from padded d = 24, inside the expanded form's radius gate only
        cg.set_option("grad_expand", -1); cg.set_option("grad_bcast", -1)
        G = cg.gramian(cg.GradientKernel(cg.EQ()), Xd, Yd)
        b = (G @ torch.from_numpy(a).cuda()).cpu().numpy()
        assert cg.get_info("last_grad_bcast") == (4 if d > 16 else 0), d
        assert relerr(b, o.grad_mul(None, o.Kernel(o.EQ), X, Y, a)) <= 1e-12
        # a translated cloud is centred first
        Gs = cg.gramian(cg.GradientKernel(cg.EQ()), Xd + 1.0e4, Yd + 1.0e4)
        bs = (Gs @ torch.from_numpy(a).cuda()).cpu().numpy()
        assert relerr(bs, o.grad_mul(None, o.Kernel(o.EQ), X + 1.0e4, Y + 1.0e4, a)) <= 1e-11
        # what the kernel does not serve falls back, it is not computed some other way: Power wrapper, the gate
        cg.set_option("grad_bcast", 4)
        (cg.gramian(cg.GradientKernel(cg.EQ() ** 2), Xd, Yd) @ torch.from_numpy(a).cuda()); assert cg.get_info("last_grad_bcast") == 0
        A2 = torch.from_numpy(rng.standard_normal((m * d, 2))).cuda()
        B2 = (G @ A2).cpu().numpy()
        assert cg.get_info("last_grad_bcast") == 4                  # matrix right-hand sides: column by column on the broadcast kernel
        assert relerr(B2[:, 1], o.grad_mul(None, o.Kernel(o.EQ), X, Y, A2[:, 1].cpu().numpy())) <= 1e-12
        cg.set_option("grad_expand", -1)
        Gw = cg.gramian(cg.GradientKernel(cg.Lengthscale(cg.EQ(), 0.02)), Xd, Yd)
        (Gw @ torch.from_numpy(a).cuda()); assert cg.get_info("last_grad_expand") == 0 and cg.get_info("last_grad_bcast") == 0
    finally:
        cg.set_option("grad_expand", -1); cg.set_option("grad_bcast", -1)


def test_broadcast_gradient_kernel_nan_and_square_case(cg, oracle):
    """gramian(GradientKernel(EQ), x) on one point set (the diagonal blocks: s rounds to ~0, clamped at 0), a NaN coordinate poisons its
    own row and every row through its column exactly as on the scalar-stream kernel, several column splits."""
    o = oracle
    rng = np.random.default_rng(5)
    n, d = 1500, 32
    X = rng.standard_normal((n, d)) * 0.6
    a = rng.standard_normal(n * d)
    Xd = torch.from_numpy(X).cuda(); ad = torch.from_numpy(a).cuda()
    try:
        G = cg.gramian(cg.GradientKernel(cg.EQ()), Xd)
        b = (G @ ad).cpu().numpy()
        assert cg.get_info("last_grad_bcast") == 4
        assert relerr(b, o.grad_mul(None, o.Kernel(o.EQ), X, X, a)) <= 1e-12
        for js in (1, 3, 7):
            cg.set_option("jsplit", js)
            assert relerr((G @ ad).cpu().numpy(), b) <= 1e-14
        cg.set_option("jsplit", 0)
        Xn = X.copy(); Xn[17, 5] = np.nan
        Gn = cg.gramian(cg.GradientKernel(cg.EQ()), torch.from_numpy(Xn).cuda())
        bn = (Gn @ ad).cpu().numpy()
        cg.set_option("grad_bcast", 0)
        bn0 = (Gn @ ad).cpu().numpy()
        assert np.array_equal(np.isnan(bn), np.isnan(bn0)) and np.isnan(bn).all()
    finally:
        cg.set_option("jsplit", 0); cg.set_option("grad_bcast", -1)
