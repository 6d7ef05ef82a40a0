"""Seeded random sweep over the dense MVM's routing space: kernel family x precision x d x (n, m) x one / two point sets x alpha, beta x the options that
pick among the round-5 kernels (symmetric forms forced on below their automatic size, row tiles per wave, fp16 / bf16 split, one-pass Sum).  Every
case is checked against the fp64 oracle of src/gramian.jl:78-87 norm-wise and row-wise; sizes are small (the oracle is O(n m)) and deliberately
ragged around the 32-row tiles and 256-row panels of the matrix-core kernels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 31, 32, 33, 63, 65, 255, 256, 257, 300, 511, 513, 777, 1025, 1300]


def _kernels(cg, o, rng):
    L = cg.Lengthscale
    l = float(rng.uniform(0.7, 2.5)); al = float(rng.uniform(0.6, 2.5))
    fams = [
        ("EQ", L(cg.EQ(), l), [(1.0, o.Kernel(o.EQ, lengthscale=l))]),
        ("MaternP1", L(cg.MaternP(1), l), [(1.0, o.Kernel(o.MATERNP, p=1, lengthscale=l))]),
        ("MaternP2", L(cg.MaternP(2), l), [(1.0, o.Kernel(o.MATERNP, p=2, lengthscale=l))]),
        ("MaternP3", L(cg.MaternP(3), l), [(1.0, o.Kernel(o.MATERNP, p=3, lengthscale=l))]),
        ("RQ", L(cg.RQ(al), l), [(1.0, o.Kernel(o.RQ, param=al, lengthscale=l))]),
        ("Cauchy", L(cg.Cauchy(), l), [(1.0, o.Kernel(o.CAUCHY, lengthscale=l))]),
        ("IMQ", cg.InverseMultiQuadratic(al), [(1.0, o.Kernel(o.IMQ, param=al))]),
        ("Exp", L(cg.Exp(), l), [(1.0, o.Kernel(o.EXP, lengthscale=l))]),
        ("Dot", cg.Dot(), [(1.0, o.Kernel(o.DOT))]),
        ("M2+EQ", 1.5 * L(cg.MaternP(2), l) + 0.5 * cg.EQ(), [(1.5, o.Kernel(o.MATERNP, p=2, lengthscale=l)), (0.5, o.Kernel(o.EQ))]),
        ("EQ+RQ+M1", L(cg.EQ(), l) + 0.7 * cg.RQ(al) + 0.2 * cg.MaternP(1), [(1.0, o.Kernel(o.EQ, lengthscale=l)), (0.7, o.Kernel(o.RQ, param=al)), (0.2, o.Kernel(o.MATERNP, p=1))]),
    ]
    return fams[int(rng.integers(len(fams)))]


@pytest.mark.parametrize("seed", range(48))
def test_random_routing_cases_match_the_oracle(cg, oracle, seed):
    o = oracle
    rng = np.random.default_rng(90000 + seed)
    opts = {"mfma_sym": int(rng.choice([-1, 1, 1, 0])), "mfma_sym_rt": int(rng.choice([-1, 1, 2])), "mfma_f16": int(rng.choice([-1, 0, -1])),
            "sum_fused": int(rng.choice([-1, 0, 1])), "dense_variant": int(rng.choice([0, 0, 0, 2, 1])), "dense_sym": int(rng.choice([-1, 1]))}
    try:
        for key, v in opts.items(): cg.set_option(key, v)
        for rep in range(4):
            name, k, terms = _kernels(cg, o, rng)
            dtype = torch.float32 if rng.random() < 0.7 else torch.float64
            d = int(rng.choice([1, 2, 3, 3, 4, 5, 6, 8, 11, 14, 16]))
            n = int(rng.choice(SIZES)); same = rng.random() < 0.55
            m = n if same else int(rng.choice(SIZES))
            scale = float(rng.choice([0.3, 0.8, 1.5]))
            npd = np.float32 if dtype == torch.float32 else np.float64
            Xh = (scale * rng.standard_normal((n, d)) + 0.2).astype(npd); Yh = Xh if same else (scale * rng.standard_normal((m, d))).astype(npd)
            ah = rng.standard_normal(m).astype(npd); y0 = rng.standard_normal(n).astype(npd)
            alpha, beta = (1.0, 0.0) if rng.random() < 0.5 else (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)))
            X = torch.from_numpy(Xh).cuda(); Y = X if same else torch.from_numpy(Yh).cuda()
            G = cg.gramian(k, X) if same else cg.gramian(k, X, Y)
            y = torch.from_numpy(y0.copy()).cuda() if beta != 0.0 else torch.full((n,), float("nan"), dtype=dtype, device="cuda")
            cg.mul_(y, G, torch.from_numpy(ah).cuda(), alpha, beta)
            Xd, Yd, ad = Xh.astype(np.float64), Yh.astype(np.float64), ah.astype(np.float64)
            Gab = sum(c * o.mul(None, kk, Xd, Yd, ad) for c, kk in terms)
            absG = sum(abs(c) * (np.abs(o.matrix(kk, Xd, Yd)) @ np.abs(ad)) for c, kk in terms)
            ref = alpha * Gab + (beta * y0.astype(np.float64) if beta != 0.0 else 0.0)
            scale_r = abs(alpha) * absG + (abs(beta) * np.abs(y0.astype(np.float64)) if beta != 0.0 else 0.0) + 1e-300
            got = y.cpu().numpy().astype(np.float64)
            tol = 1e-5 if dtype == torch.float32 else 1e-12
            info = (seed, rep, name, str(dtype), d, n, m, same, alpha, beta, opts, cg.get_info("last_dense_path"), cg.get_info("last_mfma_sym"), cg.get_info("last_mfma_f16"))
            assert np.isfinite(got).all(), info
            assert np.linalg.norm(got - ref) <= tol * max(np.linalg.norm(ref), np.linalg.norm(scale_r) * 1e-2), info
            assert float(np.max(np.abs(got - ref) / scale_r)) <= tol, info + (float(np.max(np.abs(got - ref) / scale_r)),)
    finally:
        for key, v in {"mfma_sym": -1, "mfma_sym_rt": -1, "mfma_f16": -1, "sum_fused": -1, "dense_variant": 0, "dense_sym": -1}.items(): cg.set_option(key, v)


@pytest.mark.parametrize("seed", range(24))
def test_random_gradient_cases_match_the_oracle(cg, oracle, seed):
    """GradientKernel / ValueGradientKernel block MVMs (src/gramian.jl:241-253, src/gradient.jl:86-92, 319-351): family x precision x d x sizes x
    one / two point sets x alpha, beta x expanded / direct form x broadcast kernels on or off, against the fp64 oracle."""
    o = oracle
    rng = np.random.default_rng(91000 + seed)
    opts = {"grad_expand": int(rng.choice([-1, 0, 1])), "grad_bcast": int(rng.choice([-1, 0, 1])), "grad_keep_r": int(rng.choice([-1, -1, 0, 1]))}
    try:
        for key, v in opts.items(): cg.set_option(key, v)
        for rep in range(3):
            l = float(rng.uniform(0.8, 2.0)); al = float(rng.uniform(0.7, 2.0))
            name, kin, ko = [("EQ", cg.Lengthscale(cg.EQ(), l), o.Kernel(o.EQ, lengthscale=l)), ("RQ", cg.RQ(al), o.Kernel(o.RQ, param=al)),
                             ("MaternP2", cg.Lengthscale(cg.MaternP(2), l), o.Kernel(o.MATERNP, p=2, lengthscale=l)), ("MaternP3", cg.MaternP(3), o.Kernel(o.MATERNP, p=3)),
                             ("Cauchy", cg.Cauchy(), o.Kernel(o.CAUCHY)), ("Dot^3", cg.Dot() ** 3, o.Kernel(o.DOT, power=3))][int(rng.integers(6))]
            vg = rng.random() < 0.3
            dtype = torch.float64 if rng.random() < 0.6 else torch.float32
            npd = np.float64 if dtype == torch.float64 else np.float32
            d = int(rng.choice([1, 2, 3, 5, 8, 16, 24, 32])); blk = d + 1 if vg else d
            n = int(rng.choice([1, 2, 33, 64, 65, 130, 257])); same = rng.random() < 0.6; m = n if same else int(rng.choice([1, 17, 64, 100, 200]))
            sc = 0.6 / np.sqrt(d) if name == "Dot^3" else 0.8
            Xh = (sc * rng.standard_normal((n, d))).astype(npd); Yh = Xh if same else (sc * rng.standard_normal((m, d))).astype(npd)
            ah = rng.standard_normal(m * blk).astype(npd); y0 = rng.standard_normal(n * blk).astype(npd)
            alpha, beta = (1.0, 0.0) if rng.random() < 0.5 else (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)))
            X = torch.from_numpy(Xh).cuda(); Y = X if same else torch.from_numpy(Yh).cuda()
            K = cg.ValueGradientKernel(kin) if vg else cg.GradientKernel(kin)
            G = cg.gramian(K, X) if same else cg.gramian(K, X, Y)
            y = torch.from_numpy(y0.copy()).cuda() if beta != 0.0 else torch.full((n * blk,), float("nan"), dtype=dtype, device="cuda")
            cg.mul_(y, G, torch.from_numpy(ah).cuda(), alpha, beta)
            f = o.valgrad_mul if vg else o.grad_mul
            ref = np.asarray(f(y0 if beta != 0.0 else None, ko, Xh, Yh, ah, alpha, beta)).reshape(-1)
            got = y.cpu().numpy().astype(np.float64)
            tol = 2e-5 if dtype == torch.float32 else 1e-11
            info = (seed, rep, name, "valgrad" if vg else "grad", str(dtype), d, n, m, same, alpha, beta, opts)
            assert np.isfinite(got).all(), info
            assert np.linalg.norm(got - ref) <= tol * max(np.linalg.norm(ref), 1e-30) + (1e-6 if dtype == torch.float32 else 1e-13) * np.linalg.norm(ah), info + (np.linalg.norm(got - ref) / np.linalg.norm(ref),)
    finally:
        for key in opts: cg.set_option(key, -1)
