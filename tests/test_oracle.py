"""CPU tier (`-m "not gpu"`): the oracle against its golden vectors, the reference's own test relations and
closed-form known answers; the C oracle against the numpy oracle.  (The reference ships no golden vectors —
SURVEY.md §4 — so the pins are relations + closed forms + the mpmath-checked fixtures of oracle/make_golden.py.)"""
import math

import numpy as np
import pytest

import covgram_oracle as o
import c_oracle

GOLD = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden")


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def kernel_from_fields(f):
    return o.Kernel(int(f[0]), p=int(f[1]), power=int(f[2]), param=float(f[3]), lengthscale=float(f[4]), scale=float(f[5]))


def test_golden_dense_numpy_and_c():
    g = np.load(f"{GOLD}/dense.npz")
    names = list(g["kernel_names"])
    for d in (1, 2, 3, 8, 32):
        for (n, m) in ((8, 16), (257, 129)):
            tag = f"d{d}_n{n}_m{m}"
            X, Y, a, A3, y0, Y3 = (g[f"{tag}_{s}"] for s in ("X", "Y", "a", "A3", "y0", "Y3"))
            for name, f in zip(names, g["kernel_fields"]):
                k = kernel_from_fields(f)
                assert rel(o.mul(None, k, X, Y, a), g[f"{tag}_{name}_b10"]) < 1e-14
                assert rel(o.mul(y0, k, X, Y, a, -0.7, 1.3), g[f"{tag}_{name}_b2"]) < 1e-14
                assert rel(o.mul(Y3, k, X, Y, A3, -0.7, 1.3), g[f"{tag}_{name}_B3"]) < 1e-14
                # C restatement (sequential accumulation like gramian.jl:82-83)
                assert rel(c_oracle.mvm(k, X, Y, a), g[f"{tag}_{name}_b10"]) < 1e-13
                assert rel(c_oracle.mvm(k, X, Y, a, y0, -0.7, 1.3), g[f"{tag}_{name}_b2"]) < 1e-13
                assert rel(c_oracle.mvm(k, X, Y, A3, Y3, -0.7, 1.3), g[f"{tag}_{name}_B3"]) < 1e-13


def test_golden_gradient_numpy_and_c():
    g = np.load(f"{GOLD}/gradient.npz")
    for d in (1, 5, 32):
        for n in (2, 33):
            tag = f"d{d}_n{n}"
            X, a, b0 = g[f"{tag}_X"], g[f"{tag}_a"], g[f"{tag}_b0"]
            alpha, beta = g[f"{tag}_ab"]
            for name, f in zip(g["kernel_names"], g["kernel_fields"]):
                k = kernel_from_fields(f)
                assert rel(o.grad_mul(b0, k, X, X, a, alpha, beta), g[f"{tag}_{name}"]) < 1e-14
                assert rel(c_oracle.grad_mvm(k, X, X, a, b0, alpha, beta), g[f"{tag}_{name}"]) < 1e-12


def test_golden_toeplitz_and_kronecker():
    g = np.load(f"{GOLD}/toeplitz.npz")
    for n in (32, 1000, 4096):
        for name in ("EQ", "Exp"):
            vc, a = g[f"n{n}_{name}_vc"], g[f"n{n}_a"]
            assert rel(o.toeplitz_mul(None, vc, None, a), g[f"n{n}_{name}_b"]) < 1e-10
            assert rel(c_oracle.toeplitz_mvm(vc, None, a), g[f"n{n}_{name}_b"]) < 1e-10
            assert rel(o.toeplitz_mul(g[f"n{n}_y0"], vc, None, a, 0.3, -1.1), g[f"n{n}_{name}_b_ab"]) < 1e-10
    assert rel(o.toeplitz_mul(None, g["n32_EQ_shift_vc"], g["n32_EQ_shift_vr"], g["n32_a"]), g["n32_EQ_shift_b"]) < 1e-12
    assert rel(c_oracle.toeplitz_mvm(g["n32_EQ_shift_vc"], g["n32_EQ_shift_vr"], g["n32_a"]), g["n32_EQ_shift_b"]) < 1e-12
    assert rel(o.toeplitz_mul(None, g["n32_Exp_vc"], None, g["n32_a"], circulant=True), g["n32_Exp_circ_b"]) < 1e-12
    kr = np.load(f"{GOLD}/kronecker.npz")
    F = o.matrix(o.Kernel(o.EQ), kr["x"], kr["y"])
    assert rel(o.kron_mul(None, [F, F, F], kr["a"]), kr["b"]) < 1e-13
    assert rel(o.kron_mul(None, [kr["f1"], kr["f2"], kr["f3"]], kr["av"]), kr["bv"]) < 1e-13


# --- the reference's own test relations, re-asserted on the oracle -------------------------------------------
def test_relations_test_gramian_jl():
    rng = np.random.default_rng(1)
    n = 8
    x = rng.standard_normal(n); k = o.Kernel(o.EQ)
    M = o.matrix(k, x)
    assert np.allclose(M, np.exp(-(x[:, None] - x[None, :]) ** 2 / 2), rtol=1e-15)          # test/gramian.jl:17-21
    assert np.allclose(M, M.T) and np.all(np.linalg.eigvalsh(M) > -1e-12)
    y = rng.standard_normal(2 * n); a = rng.standard_normal(2 * n)
    assert rel(o.mul(None, k, x, y, a), o.matrix(k, x, y) @ a) < 1e-15                        # :55-63
    A = rng.standard_normal((2 * n, 3))
    assert rel(o.mul(None, k, x, y, A), o.matrix(k, x, y) @ A) < 1e-15                        # :65-72
    yn = np.full(n, np.nan)
    assert np.all(np.isfinite(o.mul(yn, k, x, y, a, 1.0, 0.0)))                               # gramian.jl:80
    # gramian(x, y) defaults to the Euclidean dot product (test/gramian.jl:36-37)
    assert np.allclose(o.matrix(o.Kernel(o.DOT), x, x), np.outer(x, x))


def test_relations_test_stationary_jl():
    rng = np.random.default_rng(2)
    # phi(0) = 1 for every isotropic profile; MaternP(0) = exp(-r)
    for k in (o.Kernel(o.EQ), o.Kernel(o.EXP), o.Kernel(o.RQ, param=0.7), o.Kernel(o.GAMMAEXP, param=1.3), o.Kernel(o.CAUCHY),
              *(o.Kernel(o.MATERNP, p=p) for p in range(9))):
        assert abs(float(o.profile(k, 0.0)) - 1.0) < 1e-15
    s = rng.random(50) * 5
    assert np.allclose(o.profile(o.Kernel(o.MATERNP, p=0), s), np.exp(-np.sqrt(s)), rtol=1e-15)
    # Taylor-guarded MaternP(p) vs the naive closed form for r² = 10^(1:16) eps (test/stationary.jl:62-69)
    r2 = 10.0 ** np.arange(1, 17) * np.finfo(float).eps
    for p in (2, 3):
        naive = []
        for sv in r2:
            r = math.sqrt((2 * p + 1) * sv)
            val = sum(math.factorial(p + i) // (math.factorial(p - i) * math.factorial(i)) * (2 * r) ** (p - i) for i in range(p + 1))
            naive.append(val * math.exp(-r) / (math.factorial(2 * p) // math.factorial(p)))
        assert np.allclose(o.profile(o.Kernel(o.MATERNP, p=p), r2), naive, rtol=1e-8)       # isapprox default rtol
    # MaternP(2) near-zero Taylor: 1 - (5/6) s + (25/24) s²
    assert o.maternp_derivatives_at_zero(2) == [-5 / 6 + 0, 25 / 12] or [float(v) for v in o.maternp_derivatives_at_zero(2)] == [-5 / 6, 25 / 12]
    tiny = 1e-9
    assert abs(float(o.profile(o.Kernel(o.MATERNP, p=2), tiny)) - (1 - 5 / 6 * tiny + 25 / 24 * tiny ** 2)) < 1e-16
    # PSD-ness of every profile (test/stationary.jl:45-47)
    x = rng.standard_normal(16)
    for k in (o.Kernel(o.EQ), o.Kernel(o.EXP), o.Kernel(o.RQ, param=1.1), o.Kernel(o.GAMMAEXP, param=1.0), o.Kernel(o.CAUCHY),
              o.Kernel(o.IMQ, param=1.0), *(o.Kernel(o.MATERNP, p=p) for p in range(9))):
        assert np.all(np.linalg.eigvalsh(o.matrix(k, x)) > -1e-12)
    # Lengthscale: kl(r) ≈ k(|r|²/l²)  (test/stationary.jl:118-130)
    l = math.exp(rng.standard_normal())
    for d in (1, 2, 3):
        r = rng.standard_normal(d)
        for fam in (o.EQ, o.EXP, o.CAUCHY):
            assert np.isclose(float(o.profile(o.Kernel(fam, lengthscale=l), r @ r)), float(o.profile(o.Kernel(fam), r @ r / l ** 2)), rtol=1e-14)


def test_relations_test_gradient_jl():
    """test/gradient.jl:26-53: symmetric PSD block matrix, specialised ≈ generic (here: finite-difference mixed
    partials of the scalar kernel), mul!(Kab, K, a, α, β) ≈ α MK a + β b."""
    rng = np.random.default_rng(3)
    n, d = 2, 5
    X = rng.standard_normal((n, d)) / math.sqrt(d)
    a = rng.standard_normal(n * d); b = rng.standard_normal(n * d)
    for k in (o.Kernel(o.MATERNP, p=3), o.Kernel(o.DOT, power=3), o.Kernel(o.EQ), o.Kernel(o.RQ, param=1.0)):
        MK = o.grad_matrix(k, X)
        assert np.max(np.abs(MK - MK.T)) < 1e4 * np.finfo(float).eps
        assert np.all(np.linalg.eigvalsh((MK + MK.T) / 2) >= -1e-12)
        alpha, beta = rng.standard_normal(2)
        assert rel(o.grad_mul(b, k, X, X, a, alpha, beta), alpha * MK @ a + beta * b) < 1e-14
        # generic check: block (0,1) by central differences of k(x, y)
        h = 1e-5
        f = lambda x, y: float(o.profile(k, o.pair_arg(k, x[None], y[None])[0, 0]))
        blk = np.zeros((d, d))
        for p_ in range(d):
            for q_ in range(d):
                ep = np.zeros(d); ep[p_] = h; eq = np.zeros(d); eq[q_] = h
                blk[p_, q_] = (f(X[0] + ep, X[1] + eq) - f(X[0] + ep, X[1] - eq) - f(X[0] - ep, X[1] + eq) + f(X[0] - ep, X[1] - eq)) / (4 * h * h)
        assert np.allclose(o.grad_block(k, X[0], X[1]), blk, rtol=1e-5, atol=1e-6)
    # EQ block closed form k (I - r r')
    r = X[0] - X[1]
    assert np.allclose(o.grad_block(o.Kernel(o.EQ), X[0], X[1]), math.exp(-r @ r / 2) * (np.eye(d) - np.outer(r, r)), rtol=1e-14)
    # symmetric Gramian ⇒ aᵀ(G b) = bᵀ(G a)
    k = o.Kernel(o.MATERNP, p=2)
    assert np.isclose(a @ o.grad_mul(None, k, X, X, b), b @ o.grad_mul(None, k, X, X, a), rtol=1e-12)


def test_golden_composite_and_value_gradient():
    """Composite (Sum/Product/Power) kernels and ValueGradientKernel blocks: fixtures (mpmath-checked at generation),
    the block MVM against the explicit block matrix (test/gradient.jl:87-125 relations), and every block entry against
    numerical differentiation of k(x, y) itself."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    import make_golden as mg
    g = np.load(f"{GOLD}/composite.npz")
    kernels = dict(mg.COMPOSITES); kernels.update({nm: mg.KERNELS[nm] for nm in mg.VALGRAD_KERNELS}); kernels.update(mg.MATERN_KERNELS)
    # Matern(p + 1/2) with the Bessel function == MaternP(p) with the closed form (src/stationary.jl:85, 117-131)
    ss = np.array([0.0, 1e-9, 1e-3, 0.5, 3.0, 20.0, 400.0])
    for p in (0, 1, 2, 3, 5):
        assert np.allclose(o.profile(o.Kernel(o.MATERN, param=p + 0.5), ss), o.profile(o.Kernel(o.MATERNP, p=p), ss), rtol=1e-13, atol=1e-300)
    for d in (1, 3, 8):
        for (n, m) in ((4, 6), (65, 33)):
            tag = f"d{d}_n{n}_m{m}"
            X, Y, a, y0, ag, yg0, av, yv0 = (g[f"{tag}_{s}"] for s in ("X", "Y", "a", "y0", "ag", "yg0", "av", "yv0"))
            alpha, beta = g[f"{tag}_ab"]
            for name, k in kernels.items():
                assert rel(o.valgrad_mul(yv0, k, X, Y, av, alpha, beta), g[f"{tag}_{name}_bv"]) < 1e-14
                if name in mg.COMPOSITES or name in mg.MATERN_KERNELS:
                    assert rel(o.mul(y0, k, X, Y, a, alpha, beta), g[f"{tag}_{name}_b"]) < 1e-14
                    assert rel(o.grad_mul(yg0, k, X, Y, ag, alpha, beta), g[f"{tag}_{name}_bg"]) < 1e-14
                if n == 4:
                    M = o.valgrad_matrix(k, X, Y)
                    assert rel(alpha * (M @ av) + beta * yv0, g[f"{tag}_{name}_bv"]) < 1e-13
                    # the gradient-gradient sub-blocks are the GradientKernel Gramian
                    b = d + 1
                    idx = np.array([i * b + 1 + l for i in range(n) for l in range(d)]); jdx = np.array([j * b + 1 + l for j in range(m) for l in range(d)])
                    assert rel(M[np.ix_(idx, jdx)], o.grad_matrix(k, X, Y)) < 1e-15
    # entries of a block by central differences of the kernel function (independent of every closed form)
    rng = np.random.default_rng(3)
    h = 1e-5
    for name, k in kernels.items():
        x, y = rng.standard_normal(3) * 0.6, rng.standard_normal(3) * 0.6
        kf = lambda u, v: float(o.matrix(k, u[None], v[None])[0, 0])
        B = o.valgrad_block(k, x, y)
        E = np.eye(3)
        gx = np.array([(kf(x + h * e, y) - kf(x - h * e, y)) / (2 * h) for e in E])
        gy = np.array([(kf(x, y + h * e) - kf(x, y - h * e)) / (2 * h) for e in E])
        gxy = np.array([[(kf(x + h * e, y + h * f) - kf(x + h * e, y - h * f) - kf(x - h * e, y + h * f) + kf(x - h * e, y - h * f)) / (4 * h * h)
                         for f in E] for e in E])
        sc = max(1.0, np.abs(B).max())
        assert abs(B[0, 0] - kf(x, y)) < 1e-14 * sc and np.abs(B[1:, 0] - gx).max() < 1e-8 * sc
        assert np.abs(B[0, 1:] - gy).max() < 1e-8 * sc and np.abs(B[1:, 1:] - gxy).max() < 1e-4 * sc, name
    # symmetric case: the value-gradient Gramian is symmetric (test/gradient.jl:100-104)
    Xs = rng.standard_normal((5, 2))
    for k in kernels.values():
        if getattr(k, "family", None) == o.MATERN and k.param <= 2:
            continue                                        # phi'' (nu <= 2) / phi' (nu <= 1) are singular at r = 0, like Exp
        M = o.valgrad_matrix(k, Xs, Xs)
        assert np.abs(M - M.T).max() < 1e-12 * max(1.0, np.abs(M).max())


def test_toeplitz_direct_solvers_and_pivoted_cholesky():
    """test/toeplitz.jl:7-42 on the restated durbin / trench / levinson (src/toeplitz.jl:12-111), plus the diagonal
    normalisation the reference gets wrong (r_0 != 1), and the pivoted-Cholesky restatement (src/gramian.jl:192-199)."""
    n = 16
    x = np.linspace(-1, 1, n + 1)
    rng = np.random.default_rng(2)
    for a in (rng.random(n + 1) / n, np.exp(-np.abs(x[0] - x)), np.exp(-np.abs(x[0] - x) ** 2) / 2):
        a = a.copy(); a[0] = 1.0
        r = a[1:]
        K = o.toeplitz_dense(np.concatenate([[1.0], r[:-1]]))
        assert np.allclose(-np.linalg.solve(K, r), o.durbin(r), rtol=1e-9, atol=1e-12)              # test/toeplitz.jl:21-25
        assert np.allclose(o.trench(r[:-1]), np.linalg.inv(K), rtol=1e-8, atol=1e-10)               # :27-29
        r2 = a[:-1][1:]; b = rng.standard_normal(n)
        K2 = o.toeplitz_dense(np.concatenate([[1.0], r2]))
        assert np.allclose(np.linalg.solve(K2, b), o.levinson(r2, b), rtol=1e-9, atol=1e-12)        # :31-40
        vc = 2.5 * np.concatenate([[1.0], r2])                                                      # r_0 != 1
        assert np.allclose(np.linalg.solve(2.5 * K2, b), o.levinson_toeplitz(vc, b), rtol=1e-9, atol=1e-12)
    X = rng.standard_normal((60, 2))
    G = o.matrix(o.Kernel(o.EQ), X)
    L, piv, rank = o.pivoted_cholesky(G, tol=1e-6)
    assert rank < 60 and sorted(piv.tolist()) == list(range(60))
    assert np.abs(G - L @ L.T).max() <= 60 * 1e-6                                                  # every remaining diagonal <= tol
    assert np.allclose(np.triu(L[piv][:rank], 1), 0)                                                # P'L is lower trapezoidal
    Gd = o.matrix(o.Kernel(o.DOT), rng.standard_normal((30, 3)))                                   # exact rank 3
    assert o.pivoted_cholesky(Gd, tol=1e-10)[2] == 3
    Lf, _, rf = o.pivoted_cholesky(G + 1e-3 * np.eye(60), tol=0.0)                                  # full rank: plain Cholesky up to P
    assert rf == 60 and np.allclose(Lf @ Lf.T, G + 1e-3 * np.eye(60), atol=1e-12)


def test_profile_derivatives_against_mpmath():
    import mpmath as mp
    mp.mp.dps = 40
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    import make_golden as mg
    for k in (o.Kernel(o.EQ), o.Kernel(o.EXP), o.Kernel(o.RQ, param=1.3), o.Kernel(o.GAMMAEXP, param=1.5), o.Kernel(o.CAUCHY),
              o.Kernel(o.IMQ, param=0.8), o.Kernel(o.MATERNP, p=1), o.Kernel(o.MATERNP, p=2), o.Kernel(o.MATERNP, p=4),
              o.Kernel(o.MATERNP, p=8), o.Kernel(o.DOT, power=3), o.Kernel(o.EXPDOT), o.Kernel(o.EQ, lengthscale=0.7, scale=2.5),
              o.Kernel(o.MATERNP, p=2, power=2, lengthscale=1.7)):
        for s in (1e-3, 0.4, 2.7, 11.0):
            v, d1, d2 = o.profile_derivatives(k, np.array(s))
            assert np.isclose(float(v), float(mg.mp_phi(k, s)), rtol=1e-13)
            assert np.isclose(float(d1), float(mp.diff(lambda t: mg.mp_phi(k, t), s)), rtol=1e-12, atol=1e-300)
            assert np.isclose(float(d2), float(mp.diff(lambda t: mg.mp_phi(k, t), s, 2)), rtol=1e-11, atol=1e-13)


def test_relations_toeplitz_kronecker_lowrank():
    rng = np.random.default_rng(4)
    n = 32
    rg = o.srange(-1, 1, n); x = o.srange_points(rg)
    k = o.Kernel(o.EQ)
    vc, vr = o.toeplitz_vectors(k, rg)
    assert vr is None and rel(o.toeplitz_dense(vc), o.matrix(k, x)) < 1e-15                   # test/gramian.jl:152
    assert rel(o.toeplitz_mul(None, vc, None, x), o.matrix(k, x) @ x) < 1e-13                 # :150
    with pytest.raises(ValueError):
        o.toeplitz_vectors(k, rg, o.srange(-1, 1, n // 2))                                    # :175-178 different step
    # Kronecker: identical factors make LazyGrid's two enumerations coincide (test/algebra.jl:70-89) ...
    xs = rng.standard_normal(4); ys = rng.standard_normal(8)
    F = o.matrix(k, xs, ys)
    K3 = o.kron_dense([F, F, F])
    gx, gy = o.lazy_grid_points([xs] * 3), o.lazy_grid_points([ys] * 3)
    assert rel(K3, o.matrix(k, gx, gy)) < 1e-14
    # ... but for NON-identical factors kronecker(G1..Gq) is the row-major enumeration (reference quirk, kept)
    x1, x2 = rng.standard_normal(3), rng.standard_normal(4)
    k1, k2 = o.Kernel(o.EQ), o.Kernel(o.EXP)
    Kq = o.kron_dense([o.matrix(k1, x1), o.matrix(k2, x2)])
    def sep(P):
        return o.matrix(k1, P[:, 0]) * o.matrix(k2, P[:, 1])
    assert rel(Kq, sep(o.lazy_grid_points([x1, x2], rowmajor=True))) < 1e-14
    assert rel(Kq, sep(o.lazy_grid_points([x1, x2]))) > 1e-3
    # FiniteBasis low rank (test/mercer.jl:23-38)
    xx = rng.standard_normal(16)
    U = np.stack([np.sin(xx), np.cos(xx), xx], axis=1)
    a = rng.standard_normal(16)
    assert rel(o.lowrank_mul(None, U, U, a), (U @ U.T) @ a) < 1e-14


def test_c_oracle_f32_and_baseline_kernels():
    rng = np.random.default_rng(5)
    X = rng.standard_normal((500, 3)).astype(np.float32); a = rng.standard_normal(500).astype(np.float32)
    ref = o.mul(None, o.Kernel(o.EQ), X, X, a, dtype=np.float32)
    assert rel(c_oracle.mvm(o.Kernel(o.EQ), X, X, a), ref) < 2e-6
    import ctypes
    lib = c_oracle._load("libcovgram_cpubaseline.so")
    assert rel(c_oracle.eq_rows(X, X, a, 100, 300, lib), ref[100:300]) < 2e-6
    Xd = X.astype(np.float64); ad = a.astype(np.float64)
    assert rel(c_oracle.eq_rows(Xd, Xd, ad, 0, 500, lib), o.mul(None, o.Kernel(o.EQ), Xd, Xd, ad)) < 1e-13
    assert c_oracle.num_threads(lib) >= 1


def test_cpu_baseline_object_leaves_denormals_alone():
    """bench.py dlopen()s the -ffast-math build of the C restatement as its timed CPU baseline and afterwards runs oracle spot checks in
    the same process.  A shared object LINKED with -ffast-math carries gcc's crtfastmath constructor, which sets FTZ / DAZ in MXCSR for
    the loading thread (round 3's bench ran its checks that way).  oracle/Makefile therefore compiles with the flag and links without it:
    loading the object must leave subnormal arithmetic intact."""
    tiny = np.array([1e-310])
    assert (tiny * 1.0)[0] != 0.0
    lib = c_oracle._load("libcovgram_cpubaseline.so")
    X = np.random.default_rng(0).standard_normal((64, 3)).astype(np.float32)
    c_oracle.eq_rows(X, X, X[:, 0].copy(), 0, 64, lib)
    assert (tiny * 1.0)[0] != 0.0, "the CPU-baseline object switched on flush-to-zero"
    assert np.finfo(np.float32).smallest_subnormal > 0
