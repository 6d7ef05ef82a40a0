"""Round 4: fp64 dense Gramian MVM of WIDE points on dense_bcast_kernel (csrc/dense_bcast.hpp): |x - y|^2 expanded around cached norms,
one v_fmac_f64 per dimension and pair with the column operand taken by DPP broadcast from records held in VGPRs — `mul!(b, gramian(k, x, y), a)`
of src/gramian.jl:78-87 for Float64 points with d >= 16 (the reference README's second dense case is EQ, d = 32, n = 16384: README.md:369-395).
Against the fp64 oracle at 1e-12 norm-wise and row-wise, against the direct-difference kernel it replaces, every compiled dimension, ragged
sizes, two point sets, alpha / beta, NaN-in-y with beta = 0, a translated cloud, the radius gate and the profiles it must refuse."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return np.linalg.norm(b - ref) / np.linalg.norm(ref)


def _kernels(cg, o):
    return [("EQ", cg.EQ(), o.Kernel(o.EQ)), ("MaternP2_l", cg.Lengthscale(cg.MaternP(2), 1.7), o.Kernel(o.MATERNP, p=2, lengthscale=1.7)),
            ("RQ", 1.5 * cg.RQ(2.0), o.Kernel(o.RQ, param=2.0, scale=1.5)), ("Cauchy", cg.Cauchy(), o.Kernel(o.CAUCHY)),
            ("MaternP1", cg.MaternP(1), o.Kernel(o.MATERNP, p=1)), ("IMQ", cg.InverseMultiQuadratic(0.9), o.Kernel(o.IMQ, param=0.9)),
            ("MaternP5", cg.MaternP(5), o.Kernel(o.MATERNP, p=5))]


@pytest.mark.parametrize("d", [8, 11, 16, 23, 32, 40, 48, 57, 64])
def test_dense_bcast_kernel_matches_the_oracle(cg, oracle, d):
    o = oracle
    rng = np.random.default_rng(2200 + d)
    n, m = 517, 1203
    X = rng.standard_normal((n, d)) / np.sqrt(d) * 1.7; Y = rng.standard_normal((m, d)) / np.sqrt(d) * 1.7 + 0.1
    a = rng.standard_normal(m); y0 = rng.standard_normal(n)
    Xd, Yd, ad = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda(), torch.from_numpy(a).cuda()
    try:
        for name, k, ko in _kernels(cg, o):
            G = cg.gramian(k, Xd, Yd)
            ref = o.mul(y0, ko, X, Y, a, 0.7, -1.1)
            absref = np.abs(0.7) * (np.abs(o.matrix(ko, X, Y)) @ np.abs(a)) + 1.1 * np.abs(y0)
            outs = {}
            for bc in (0, 1):
                cg.set_option("dense_bcast", bc)
                yd = torch.from_numpy(y0.copy()).cuda(); cg.mul_(yd, G, ad, 0.7, -1.1)
                assert cg.get_info("last_dense_bcast") == bc and cg.get_info("last_dense_path") == 1, (name, d, bc)
                outs[bc] = yd.cpu().numpy()
                assert relerr(outs[bc], ref) <= 1e-12, (name, d, bc, relerr(outs[bc], ref))
                assert float(np.max(np.abs(outs[bc] - ref) / absref)) <= 1e-12, (name, d, bc)
            # beta = 0 never reads y (NaN-safe, src/gramian.jl:80)
            cg.set_option("dense_bcast", 1)
            yn = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda"); cg.mul_(yn, G, ad)
            assert relerr(yn.cpu().numpy(), o.mul(None, ko, X, Y, a)) <= 1e-12
        # automatic rule: from padded d = 16, inside the gate; a translated cloud is centred first
        cg.set_option("dense_bcast", -1)
        G = cg.gramian(cg.EQ(), Xd + 1.0e4, Yd + 1.0e4)
        b = (G @ ad).cpu().numpy()
        assert cg.get_info("last_dense_bcast") == (1 if d > 12 else 0), d
        assert relerr(b, o.mul(None, o.Kernel(o.EQ), X + 1.0e4, Y + 1.0e4, a)) <= 1e-11
        # outside the gate (a cloud ~100 lengthscales wide): direct differences, still accurate
        Gw = cg.gramian(cg.Lengthscale(cg.EQ(), 0.02), Yd); bw = (Gw @ ad).cpu().numpy()      # (one point set: the diagonal keeps the product away from 0)
        assert cg.get_info("last_dense_bcast") == 0
        assert relerr(bw, o.mul(None, o.Kernel(o.EQ, lengthscale=0.02), Y, Y, a)) <= 1e-12
        # refused, not computed some other way: profiles that are not smooth in s at 0, Power wrappers, matrix right-hand sides, fp32
        cg.set_option("dense_bcast", 1)
        for kk in (cg.Exp(), cg.GammaExp(1.3), cg.MaternP(0), cg.EQ() ** 2, cg.Dot() ** 2):
            (cg.gramian(kk, Xd, Yd) @ ad); assert cg.get_info("last_dense_bcast") == 0, type(kk).__name__
        A3 = torch.from_numpy(rng.standard_normal((m, 3))).cuda()
        B3 = (cg.gramian(cg.EQ(), Xd, Yd) @ A3).cpu().numpy()
        assert cg.get_info("last_dense_bcast") == 0 and relerr(B3[:, 2], o.mul(None, o.Kernel(o.EQ), X, Y, A3[:, 2].cpu().numpy())) <= 1e-12
        (cg.gramian(cg.Lengthscale(cg.EQ(), 30.0), Xd.float(), Yd.float()) @ ad.float()); assert cg.get_info("last_dense_bcast") == 0
    finally:
        cg.set_option("dense_bcast", -1)


def test_dense_bcast_square_case_splits_and_nan(cg, oracle):
    """gramian(k, x) on one point set (s rounds to ~0 on the diagonal: clamped at 0, k = 1), several column splits, a NaN coordinate."""
    o = oracle
    rng = np.random.default_rng(7)
    n, d = 3000, 32
    X = rng.standard_normal((n, d)) * 0.5; a = rng.standard_normal(n)
    Xd, ad = torch.from_numpy(X).cuda(), torch.from_numpy(a).cuda()
    try:
        G = cg.gramian(cg.MaternP(2), Xd)
        b = (G @ ad).cpu().numpy()
        assert cg.get_info("last_dense_bcast") == 1
        assert relerr(b, o.mul(None, o.Kernel(o.MATERNP, p=2), X, X, a)) <= 1e-12
        for js in (1, 3, 7, 32):
            cg.set_option("jsplit", js)
            assert relerr((G @ ad).cpu().numpy(), b) <= 1e-14, js
        cg.set_option("jsplit", 0)
        Xn = X.copy(); Xn[5, 3] = np.nan
        Gn = cg.gramian(cg.EQ(), torch.from_numpy(Xn).cuda())
        bn = (Gn @ ad).cpu().numpy()
        cg.set_option("dense_bcast", 0)
        bn0 = (Gn @ ad).cpu().numpy()
        assert np.array_equal(np.isnan(bn), np.isnan(bn0)) and np.isnan(bn).all()
    finally:
        cg.set_option("jsplit", 0); cg.set_option("dense_bcast", -1)


@pytest.mark.parametrize("n,d", [(9000, 32), (8200, 17), (12345, 48)])
def test_dense_bcast_symmetric_form(cg, oracle, n, d):
    """gramian(k, x) from n = 8192: the upper triangle once WITH the broadcast distance (dense_bcast_sym_kernel) — row sums and column sums
    of the same evaluations, the column sums reduced four columns at a time; single launch and the partial (multi-GPU) form; 384 oracle rows."""
    o = oracle
    import c_oracle
    rng = np.random.default_rng(n + d)
    X = rng.standard_normal((n, d)) / np.sqrt(d) * 1.6; a = rng.standard_normal(n)
    Xd, ad = torch.from_numpy(X).cuda(), torch.from_numpy(a).cuda()
    rows = np.sort(rng.choice(n, 384, replace=False))
    try:
        for k, ko in ((cg.EQ(), o.Kernel(o.EQ)), (cg.Lengthscale(cg.MaternP(2), 1.3), o.Kernel(o.MATERNP, p=2, lengthscale=1.3))):
            G = cg.gramian(k, Xd)
            ref = c_oracle.mvm(ko, X[rows], X, a)
            y0 = rng.standard_normal(n)
            res = {}
            for sym, bc in ((0, 0), (1, 0), (0, 1), (1, 1)):
                cg.set_option("dense_sym", sym); cg.set_option("dense_bcast", bc)
                y = torch.from_numpy(y0.copy()).cuda(); G.mul_(y, ad, 1.3, -0.4)
                assert cg.get_info("last_dense_sym") == sym and cg.get_info("last_dense_bcast") == bc, (sym, bc)
                res[(sym, bc)] = y.cpu().numpy()
                assert relerr(res[(sym, bc)][rows], 1.3 * ref - 0.4 * y0[rows]) <= 1e-12, (n, d, sym, bc)
            assert relerr(res[(1, 1)], res[(0, 0)]) <= 1e-13
            # the default takes both
            cg.set_option("dense_sym", -1); cg.set_option("dense_bcast", -1)
            b = (G @ ad).cpu().numpy()
            assert cg.get_info("last_dense_sym") == 1 and cg.get_info("last_dense_bcast") == 1
            assert relerr(b[rows], ref) <= 1e-12
            # partial form: the cyclic 64-row blocks of rank r of 3 add up to G a
            total = torch.zeros(n, dtype=torch.float64, device="cuda"); part = torch.empty_like(total)
            for r in range(3):
                part.fill_(float("nan")); G.sym_partial_(part, ad, r, 3)
                assert cg.get_info("last_dense_bcast") == 1 and torch.isfinite(part).all()
                total += part
            assert relerr(total.cpu().numpy()[rows], ref) <= 1e-12
    finally:
        cg.set_option("dense_sym", -1); cg.set_option("dense_bcast", -1)
