"""GPU tier: seeded random sweep over shapes / kernels / right-hand sides / alpha-beta for the dense, gradient and value-gradient
entry points — ragged sizes around every tile edge (32-row MFMA tiles, 64-row waves, 512-column chunks, 4-RHS passes), both dense
paths (matrix cores and direct differences), against the fp64 oracle."""
import numpy as np
import pytest
import torch

import kernel_cases

pytestmark = pytest.mark.gpu


def relerr(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    den = np.linalg.norm(ref)
    return np.linalg.norm(b - ref) / (den if den > 0 else 1.0)


@pytest.mark.parametrize("seed", range(6))
def test_random_dense_cases(cg, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    cases = kernel_cases.cases(cg) + kernel_cases.composite_cases(cg)
    try:
        for _ in range(14):
            name, k, ko = cases[rng.integers(len(cases))]
            dt = [np.float32, np.float64][rng.integers(2)]
            d = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 70]))
            n = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 129, 255, 257, 300]))
            m = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 513, 1025]))
            p = int(rng.choice([1, 1, 1, 2, 3, 4, 5, 9]))
            alpha, beta = [(1.0, 0.0), (-0.7, 1.3), (2.0, 0.0), (0.0, 0.5)][rng.integers(4)]
            variant = int(rng.choice([0, 1, 2]))
            X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(dt); Y = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(dt)
            A = rng.standard_normal((m, p)).astype(dt); Y0 = rng.standard_normal((n, p)).astype(dt)
            if p == 1: A, Y0 = A[:, 0], Y0[:, 0]
            cg.set_option("dense_variant", variant)
            G = cg.gramian(k, torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
            yd = torch.from_numpy(Y0.copy()).cuda()
            if beta == 0.0:
                yd.fill_(float("nan"))                                   # beta == 0 must not read y
            cg.mul_(yd, G, torch.from_numpy(A).cuda(), alpha, beta)
            ref = oracle.mul(Y0, ko, X, Y, A, alpha, beta, dt)
            tol = 1e-5 if dt == np.float32 else 1e-12
            e = relerr(yd.cpu().numpy(), ref)
            assert e <= tol or np.linalg.norm(ref) < 1e-30, (name, dt.__name__, d, n, m, p, alpha, beta, variant, e)
    finally:
        cg.set_option("dense_variant", 0)


@pytest.mark.parametrize("seed", range(4))
def test_random_symmetric_cases(cg, oracle, seed):
    """gramian(k, x) with the symmetric upper-triangle kernels forced on (mfma_sym = 1, dense_sym = 1, dense_variant = 2 where the shape allows):
    every kernel case, random n around the 32-row tiles / 256-row panels / 64-tile chunks, random chunk splits, alpha / beta, the
    multi-GPU partial form for a random world size — whatever path the library ends up taking must match the fp64 oracle."""
    rng = np.random.default_rng(5000 + seed)
    cases = kernel_cases.cases(cg) + kernel_cases.composite_cases(cg)
    try:
        for _ in range(14):
            name, k, ko = cases[rng.integers(len(cases))]
            dt = [np.float32, np.float32, np.float64][rng.integers(3)]
            d = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 17, 24, 33, 48, 64]))      # (from 16: the fp64 broadcast kernels of round 4, symmetric form)
            n = int(rng.choice([1, 2, 31, 33, 255, 256, 257, 511, 513, 1000, 2047, 2049, 2600]))
            alpha, beta = [(1.0, 0.0), (-0.7, 1.3), (2.0, 0.0)][rng.integers(3)]
            X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(dt)
            a = rng.standard_normal(n).astype(dt); y0 = rng.standard_normal(n).astype(dt)
            cg.set_option("mfma_sym", 1); cg.set_option("dense_variant", int(rng.choice([0, 2]))); cg.set_option("jsplit", int(rng.choice([0, 0, 2, 5])))
            cg.set_option("dense_sym", 1)                     # the direct-difference symmetric kernels (fp64, fp32) wherever they are eligible
            cg.set_option("mfma_f16", int(rng.choice([-1, 0, 1])))      # fp32 EQ on the matrix cores: fp16 two-way / bf16 three-way split of the coordinates
            cg.set_option("dense_bcast", int(rng.choice([-1, 0, 1])))   # fp64: expanded distance with broadcast records — by rule, never, from d = 8
            Xd = torch.from_numpy(X).cuda(); ad = torch.from_numpy(a).cuda()
            G = cg.gramian(k, Xd)
            yd = torch.from_numpy(y0.copy()).cuda()
            if beta == 0.0:
                yd.fill_(float("nan"))
            cg.mul_(yd, G, ad, alpha, beta)
            ref = oracle.mul(y0, ko, X, X, a, alpha, beta, dt)
            tol = 1e-5 if dt == np.float32 else 1e-12
            e = relerr(yd.cpu().numpy(), ref)
            assert e <= tol or np.linalg.norm(ref) < 1e-30, (name, dt.__name__, d, n, alpha, beta, cg.get_info("last_mfma_sym"), e)
            if hasattr(G, "sym_partial_supported") and G.sym_partial_supported():
                world = int(rng.choice([2, 3, 5]))
                tot = torch.zeros(n, dtype=Xd.dtype, device="cuda"); part = torch.empty_like(tot)   # (fp32 matrix-core panels or fp64 direct-difference blocks)
                for r in range(world):
                    G.sym_partial_(part, ad, r, world); tot += part
                e = relerr(tot.cpu().numpy(), oracle.mul(None, ko, X, X, a, dtype=dt))
                assert e <= tol, (name, d, n, world, e)
    finally:
        cg.set_option("mfma_sym", -1); cg.set_option("dense_variant", 0); cg.set_option("jsplit", 0); cg.set_option("dense_sym", -1); cg.set_option("dense_bcast", -1); cg.set_option("mfma_f16", -1)


@pytest.mark.parametrize("seed", range(4))
def test_random_gradient_cases(cg, oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    cases = kernel_cases.grad_cases(cg) + kernel_cases.composite_cases(cg)
    for _ in range(10):
        name, k, ko = cases[rng.integers(len(cases))]
        dt = [np.float32, np.float64][rng.integers(2)]
        d = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 31, 32, 48, 49, 64, 65, 80]))
        n = int(rng.choice([1, 2, 33, 64, 65, 130]))
        m = int(rng.choice([1, 2, 31, 64, 65, 200, 513]))
        vg = bool(rng.integers(2))
        bd = d + (1 if vg else 0)
        alpha, beta = [(1.0, 0.0), (-0.7, 1.3), (0.5, 0.0)][rng.integers(3)]
        X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(dt); Y = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(dt)
        a = rng.standard_normal(m * bd).astype(dt); y0 = rng.standard_normal(n * bd).astype(dt)
        K = cg.gramian((cg.ValueGradientKernel if vg else cg.GradientKernel)(k), torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
        cg.set_option("grad_bcast", int(rng.choice([-1, 0, 1, 4])))      # fp64 expanded form: scalar stream / broadcast records (round 4), by rule or forced
        yd = torch.from_numpy(y0.copy()).cuda()
        if beta == 0.0:
            yd.fill_(float("nan"))
        cg.mul_(yd, K, torch.from_numpy(a).cuda(), alpha, beta)
        ref = (oracle.valgrad_mul if vg else oracle.grad_mul)(y0, ko, X, Y, a, alpha, beta, dt)
        tol = 1e-5 if dt == np.float32 else 1e-12
        e = relerr(yd.cpu().numpy(), ref)
        cg.set_option("grad_bcast", -1)
        assert e <= tol, (name, dt.__name__, d, n, m, vg, alpha, beta, e)


@pytest.mark.parametrize("seed", range(3))
def test_random_structured_cases(cg, oracle, seed):
    """Toeplitz (rectangular, around the four-step / fused thresholds), circulant, Kronecker (ragged factor shapes), low rank."""
    import ctypes as C
    rng = np.random.default_rng(3000 + seed)
    f = cg._ffi
    for _ in range(6):
        dt = [np.float32, np.float64][rng.integers(2)]
        tdt = torch.float32 if dt == np.float32 else torch.float64
        tol = 1e-5 if dt == np.float32 else 1e-10
        n = int(rng.choice([1, 2, 5, 100, 4095, 4096, 8191, 8193, 16385, 40000, 70001]))
        m = int(rng.choice([1, 3, 64, 4097, 8192, 30000, 65537]))
        vc = rng.standard_normal(n).astype(dt); vr = rng.standard_normal(m).astype(dt); vr[0] = vc[0]
        a = rng.standard_normal(m).astype(dt); y0 = rng.standard_normal(n).astype(dt)
        T = cg.Toeplitz(torch.from_numpy(vc).cuda(), torch.from_numpy(vr).cuda())
        yd = torch.from_numpy(y0.copy()).cuda()
        cg.mul_(yd, T, torch.from_numpy(a).cuda(), 0.5, -1.5)
        ref = oracle.toeplitz_mul(y0, vc, vr, a, 0.5, -1.5)
        assert relerr(yd.cpu().numpy(), ref) <= tol, ("toeplitz", dt.__name__, n, m, relerr(yd.cpu().numpy(), ref))
    for _ in range(3):
        nc = int(rng.choice([1, 2, 33, 1000, 4099]))
        vc = rng.standard_normal(nc); a = rng.standard_normal(nc)
        Cc = cg.Circulant(torch.from_numpy(vc).cuda())
        assert relerr((Cc @ torch.from_numpy(a).cuda()).cpu().numpy(), oracle.toeplitz_mul(None, vc, None, a, circulant=True)) <= 1e-10
    for _ in range(4):
        q = int(rng.integers(1, 5))
        shapes = [(int(rng.integers(1, 9)), int(rng.integers(1, 9))) for _ in range(q)]
        Fs = [rng.standard_normal(s) for s in shapes]
        a = rng.standard_normal(int(np.prod([s[1] for s in shapes]))); y0 = rng.standard_normal(int(np.prod([s[0] for s in shapes])))
        Kp = cg.kronecker(*[torch.from_numpy(Fm).cuda() for Fm in Fs])
        yd = torch.from_numpy(y0.copy()).cuda()
        cg.mul_(yd, Kp, torch.from_numpy(a).cuda(), -0.3, 0.9)
        dense = Fs[0]
        for Fm in Fs[1:]: dense = np.kron(dense, Fm)
        assert relerr(yd.cpu().numpy(), -0.3 * dense @ a + 0.9 * y0) <= 1e-12, ("kron", shapes)
    for _ in range(4):
        n, m, r = int(rng.choice([1, 7, 1023, 1025, 5000])), int(rng.choice([1, 9, 1024, 3001])), int(rng.choice([1, 2, 31, 32, 33, 70]))
        dt = [np.float32, np.float64][rng.integers(2)]
        tdt = torch.float32 if dt == np.float32 else torch.float64
        U = rng.standard_normal((n, r)).astype(dt); V = rng.standard_normal((m, r)).astype(dt)
        a = rng.standard_normal(m).astype(dt); y0 = rng.standard_normal(n).astype(dt)
        Lp = cg.LazyMatrixProduct(torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda())
        yd = torch.from_numpy(y0.copy()).cuda()
        cg.mul_(yd, Lp, torch.from_numpy(a).cuda(), 1.5, -0.5)
        assert relerr(yd.cpu().numpy(), oracle.lowrank_mul(y0, U, V, a, 1.5, -0.5)) <= (1e-5 if dt == np.float32 else 1e-12), ("lowrank", n, m, r)
