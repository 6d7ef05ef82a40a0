"""gramian(k, x) for the isotropic single profiles on the two-row-tile symmetric matrix-core kernel (round 5; csrc/dense_mfma_sym2.hpp, generic form):
4 waves x 2 row tiles per 256-row panel at one or two MFMAs per tile (d <= 3 with the bf16 split, d <= 6 with the fp16 split).  The reference evaluates
every entry of the symmetric Gramian (src/gramian.jl:78-87); this kernel evaluates each tile on or above the diagonal once.  Checked against the fp64
oracle norm- and row-wise at 1e-5, against the one-row-tile panels on the same inputs (option "mfma_sym_rt" = 1), on ragged sizes (n not a multiple of
the 256-row panel, of 32, smaller than a panel), with alpha / beta and a NaN-filled output, and as the cyclic partial products of a multi-GPU MVM
(covgram_mvm_sym_partial: the partials of all ranks add up to G a)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return np.linalg.norm(b - ref) / np.linalg.norm(ref)


def rowwise(b, ref, absref):
    return float(np.max(np.abs(np.asarray(b, dtype=np.float64) - ref) / absref))


def _profiles(cg, o):
    L = cg.Lengthscale
    return [("MaternP(2)", cg.MaternP(2), o.Kernel(o.MATERNP, p=2)), ("MaternP(1;l=1.3)", L(cg.MaternP(1), 1.3), o.Kernel(o.MATERNP, p=1, lengthscale=1.3)),
            ("MaternP(3)", cg.MaternP(3), o.Kernel(o.MATERNP, p=3)), ("MaternP(4;l=2)", L(cg.MaternP(4), 2.0), o.Kernel(o.MATERNP, p=4, lengthscale=2.0)),
            ("RQ(1.5)", cg.RQ(1.5), o.Kernel(o.RQ, param=1.5)), ("Cauchy", cg.Cauchy(), o.Kernel(o.CAUCHY)), ("IMQ(1.2)", cg.InverseMultiQuadratic(1.2), o.Kernel(o.IMQ, param=1.2)),
            ("EQ^2", cg.EQ() ** 2, o.Kernel(o.EQ, power=2))]


@pytest.mark.parametrize("n,d", [(2048, 3), (1537, 1), (1000, 2), (777, 3), (200, 3), (2300, 5), (1291, 6), (31, 2)])
def test_two_row_tile_symmetric_kernel_generic_profiles(cg, oracle, n, d):
    o = oracle
    rng = np.random.default_rng(7000 + n + d)
    Xh = ((0.5 if d >= 5 else 0.9) * rng.standard_normal((n, d))).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); Xd, ad = Xh.astype(np.float64), ah.astype(np.float64)
    try:
        cg.set_option("mfma_sym", 1)
        for name, k, ko in _profiles(cg, o):
            G = cg.gramian(k, X)
            ref = o.mul(None, ko, Xd, Xd, ad); absref = np.abs(o.matrix(ko, Xd, Xd)) @ np.abs(ad)
            cg.set_option("mfma_sym_rt", -1)
            y = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda"); G.mul_(y, a)
            assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_sym") == 1, name
            f16 = cg.get_info("last_mfma_f16")
            k2 = (d + 2 + 3) // 4 if f16 else (d + 1 + 1) // 2
            assert cg.get_info("last_mfma_sym_rt") == (2 if k2 <= 2 else 1), (name, d, f16)
            b = y.cpu().numpy()
            assert np.isfinite(b).all(), name
            assert relerr(b, ref) <= 1e-5 and rowwise(b, ref, absref) <= 1e-5, (name, n, d, relerr(b, ref), rowwise(b, ref, absref))
            cg.set_option("mfma_sym_rt", 1)
            y1 = torch.empty_like(y); G.mul_(y1, a)
            assert cg.get_info("last_mfma_sym_rt") == 1
            assert rowwise(b, y1.cpu().numpy().astype(np.float64), absref) <= 2e-6, name        # same sums, another association across row tiles
            cg.set_option("mfma_sym_rt", -1)
            y2 = torch.from_numpy(ah[::-1].copy()).cuda(); G.mul_(y2, a, -0.7, 1.3)
            assert relerr(y2.cpu().numpy(), -0.7 * ref + 1.3 * ad[::-1]) <= 1e-5, name
    finally:
        cg.set_option("mfma_sym", -1); cg.set_option("mfma_sym_rt", -1)


@pytest.mark.parametrize("n,d,world", [(3000, 3, 3), (2049, 2, 8), (1500, 5, 2)])
def test_two_row_tile_generic_partials_add_up(cg, oracle, n, d, world):
    """rank r of `world` takes the panels p = r (mod world): the partial products add up to G a (ONE all-reduce on real ranks)"""
    o = oracle
    rng = np.random.default_rng(7100 + n)
    Xh = ((0.5 if d >= 5 else 0.9) * rng.standard_normal((n, d))).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); Xd, ad = Xh.astype(np.float64), ah.astype(np.float64)
    try:
        cg.set_option("mfma_sym", 1)
        for name, k, ko in _profiles(cg, o)[:5:2]:
            G = cg.gramian(k, X)
            if not G.sym_partial_supported(world): pytest.skip("no symmetric partial form here")
            tot = np.zeros(n)
            part = torch.empty(n, dtype=torch.float32, device="cuda")
            for r in range(world):
                part.fill_(float("nan")); G.sym_partial_(part, a, r, world)
                assert cg.get_info("last_mfma_sym_rt") == 2, name
                tot += part.cpu().numpy().astype(np.float64)
            ref = o.mul(None, ko, Xd, Xd, ad); absref = np.abs(o.matrix(ko, Xd, Xd)) @ np.abs(ad)
            assert relerr(tot, ref) <= 1e-5 and rowwise(tot, ref, absref) <= 1e-5, (name, relerr(tot, ref))
    finally:
        cg.set_option("mfma_sym", -1)


@pytest.mark.parametrize("n,d,f16", [(1500, 20, -1), (2049, 24, -1), (777, 17, -1), (1300, 12, 0), (1025, 11, 0), (600, 32, 16)])
def test_eq_symmetric_six_mfmas_per_tile_staged_kernel(cg, oracle, n, d, f16):
    """gramian(EQ, x) at six MFMAs per tile (d = 17 .. 24 with the fp16 split, 11 .. 12 with bf16): the staged 8-wave symmetric kernel (round 5; the
    one-tile-per-stage 4-wave kernel before) — against the oracle, as cyclic partials, and (f16 = 16: option mfma_sym_st = 16) the diagnostic
    eight-MFMA form."""
    o = oracle
    rng = np.random.default_rng(7300 + n + d)
    Xh = (2.0 / np.sqrt(d) * rng.standard_normal((n, d))).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); Xd, ad = Xh.astype(np.float64), ah.astype(np.float64)
    ko = o.Kernel(o.EQ)
    ref = o.mul(None, ko, Xd, Xd, ad); absref = np.abs(o.matrix(ko, Xd, Xd)) @ np.abs(ad)
    try:
        cg.set_option("mfma_sym", 1)
        if f16 == 16: cg.set_option("mfma_sym_st", 16)
        else: cg.set_option("mfma_f16", f16)
        G = cg.gramian(cg.EQ(), X)
        y = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda"); G.mul_(y, a)
        assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_sym") == 1
        b = y.cpu().numpy()
        assert relerr(b, ref) <= 1e-5 and rowwise(b, ref, absref) <= 1e-5, (n, d, relerr(b, ref), rowwise(b, ref, absref))
        y2 = torch.from_numpy(ah[::-1].copy()).cuda(); G.mul_(y2, a, -0.7, 1.3)
        assert relerr(y2.cpu().numpy(), -0.7 * ref + 1.3 * ad[::-1]) <= 1e-5
        tot = np.zeros(n); part = torch.empty(n, dtype=torch.float32, device="cuda")
        for r in range(3):
            part.fill_(float("nan")); G.sym_partial_(part, a, r, 3); tot += part.cpu().numpy().astype(np.float64)
        assert relerr(tot, ref) <= 1e-5 and rowwise(tot, ref, absref) <= 1e-5
    finally:
        cg.set_option("mfma_sym", -1); cg.set_option("mfma_f16", -1); cg.set_option("mfma_sym_st", 0)
