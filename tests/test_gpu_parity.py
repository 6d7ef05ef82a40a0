"""GPU parity: the HIP path (through the C ABI) against the fp64 oracle on seeded inputs.

Tolerances (stated by BASELINE.json / SURVEY.md §8d): rel-err = ||b - b_ref||_2 / ||b_ref||_2 against the
fp64 oracle: <= 1e-5 for fp32, <= 1e-12 for fp64 dense/gradient, <= 1e-10 for fp64 Toeplitz."""
import numpy as np
import pytest
import torch

import kernel_cases

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-5, torch.float64: 1e-12}


def relerr(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    den = np.linalg.norm(ref)
    return np.linalg.norm(b - ref) / (den if den > 0 else 1.0)


def npdt(dt):
    return np.float32 if dt == torch.float32 else np.float64


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("d", [1, 2, 3, 5, 8, 32])
def test_dense_mvm_all_kernels(cg, oracle, dtype, d):
    rng = np.random.default_rng(0xC0F + d)
    n, m = 257, 129
    X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(npdt(dtype))
    Y = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(npdt(dtype))
    a = rng.standard_normal(m).astype(npdt(dtype))
    y0 = rng.standard_normal(n).astype(npdt(dtype))
    Xd, Yd, ad = (torch.from_numpy(v).cuda() for v in (X, Y, a))
    for name, k, ko in kernel_cases.cases(cg):
        G = cg.gramian(k, Xd, Yd)
        assert tuple(G.shape) == (n, m)
        ref = oracle.mul(None, ko, X, Y, a, 1.0, 0.0, npdt(dtype))
        b = (G @ ad).cpu().numpy()
        assert relerr(b, ref) <= TOL[dtype], (name, d, relerr(b, ref))
        # 5-argument mul!, beta != 0
        alpha, beta = -0.7, 1.3
        yd = torch.from_numpy(y0.copy()).cuda()
        out = cg.mul_(yd, G, ad, alpha, beta)
        assert out is yd
        ref2 = oracle.mul(y0, ko, X, Y, a, alpha, beta, npdt(dtype))
        assert relerr(yd.cpu().numpy(), ref2) <= TOL[dtype], (name, d)
        # beta == 0 must ignore NaNs in y (src/gramian.jl:80)
        yn = torch.full((n,), float("nan"), dtype=dtype, device="cuda")
        cg.mul_(yn, G, ad, 1.0, 0.0)
        assert relerr(yn.cpu().numpy(), ref) <= TOL[dtype], (name, d)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_dense_matrix_rhs_and_matrix(cg, oracle, dtype):
    """test/gramian.jl:55-72: G*a ≈ Matrix(G)*a, G*A ≈ Matrix(G)*A for an n×2n Gramian, p = 3 (and p = 6)."""
    rng = np.random.default_rng(7)
    n, d = 8, 1
    x = rng.standard_normal(n).astype(npdt(dtype)); y = rng.standard_normal(2 * n).astype(npdt(dtype))
    k, ko = cg.EQ(), oracle.Kernel(oracle.EQ)
    G = cg.gramian(k, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    assert tuple(G.shape) == (n, 2 * n)
    M = G.to_dense().cpu().numpy()
    assert relerr(M, oracle.matrix(ko, x, y)) <= TOL[dtype]
    for p in (3, 6):
        A = rng.standard_normal((2 * n, p)).astype(npdt(dtype))
        B = (G @ torch.from_numpy(A).cuda()).cpu().numpy()
        assert B.shape == (n, p)
        assert relerr(B, oracle.mul(None, ko, x, y, A)) <= TOL[dtype]
        assert relerr(B, M.astype(np.float64) @ A) <= 10 * TOL[dtype]
    # indexing (test/gramian.jl:75-95)
    assert abs(float(G[2, 5]) - k(x[2], y[5])) <= 1e-6
    sub = G[2:n - 1, 3:2 * n - 4].cpu().numpy()
    assert relerr(sub, oracle.matrix(ko, x[2:n - 1], y[3:2 * n - 4])) <= TOL[dtype]


@pytest.mark.parametrize("shape", [(1, 1), (1, 700), (700, 1), (1023, 1025), (2049, 513)])
def test_dense_ragged_shapes(cg, oracle, shape):
    n, m = shape
    rng = np.random.default_rng(n * 7919 + m)
    X = rng.standard_normal((n, 3)); Y = rng.standard_normal((m, 3)); a = rng.standard_normal(m)
    for dtype in (torch.float32, torch.float64):
        G = cg.gramian(cg.MaternP(2), torch.from_numpy(X.astype(npdt(dtype))).cuda(), torch.from_numpy(Y.astype(npdt(dtype))).cuda())
        b = (G @ torch.from_numpy(a.astype(npdt(dtype))).cuda()).cpu().numpy()
        ref = oracle.mul(None, oracle.Kernel(oracle.MATERNP, p=2), X.astype(npdt(dtype)), Y.astype(npdt(dtype)), a.astype(npdt(dtype)), dtype=npdt(dtype))
        assert relerr(b, ref) <= TOL[dtype], (shape, dtype)


def test_config1_maternp2_d3_n4096_f64(cg, oracle):
    """BASELINE.json configs[0]: MaternP(2), d=3, n=4096, fp64."""
    rng = np.random.default_rng(0xC0F + 0)
    X = rng.standard_normal((4096, 3)); a = rng.standard_normal(4096)
    G = cg.gramian(cg.MaternP(2), torch.from_numpy(X).cuda())
    assert G.issymmetric() and G.isposdef()
    b = (G @ torch.from_numpy(a).cuda()).cpu().numpy()
    ref = oracle.mul(None, oracle.Kernel(oracle.MATERNP, p=2), X, X, a)
    assert relerr(b, ref) <= 1e-12


@pytest.mark.parametrize("m", [4999, 5000, 5001, 511, 513])
def test_dense_odd_column_counts_and_split_options(cg, oracle, m):
    """fp32 streams two columns per packed instruction: odd m exercises the zero-weight pad column, and the J-split
    options exercise the partial-slab reduction."""
    rng = np.random.default_rng(11 + m)
    X = rng.standard_normal((3000, 3)).astype(np.float32); Y = rng.standard_normal((m, 3)).astype(np.float32)
    a = rng.standard_normal(m).astype(np.float32)
    G = cg.gramian(cg.EQ(), torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
    ad = torch.from_numpy(a).cuda()
    ref = oracle.mul(None, oracle.Kernel(oracle.EQ), X, Y, a, dtype=np.float32)
    for js in (0, 1, 3):
        try:
            cg.set_option("jsplit", js)
            assert relerr((G @ ad).cpu().numpy(), ref) <= 1e-5, (m, js)
        finally:
            cg.set_option("jsplit", 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("d", [1, 5, 32])
def test_gradient_mvm(cg, oracle, dtype, d):
    """test/gradient.jl:26-53: mul!(Kab, K, a, α, β) ≈ α MK a + β b, K symmetric, for n ∈ {2, 33}."""
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    for n in (2, 33):
        rng = np.random.default_rng(100 * d + n)
        X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(npdt(dtype))
        a = rng.standard_normal(n * d).astype(npdt(dtype)); b0 = rng.standard_normal(n * d).astype(npdt(dtype))
        alpha, beta = rng.standard_normal(2)
        Xd = torch.from_numpy(X).cuda(); ad = torch.from_numpy(a).cuda()
        for name, k, ko in kernel_cases.grad_cases(cg):
            K = cg.gramian(cg.GradientKernel(k), Xd)
            assert isinstance(K, cg.BlockGramian) and tuple(K.shape) == (n * d, n * d)
            ref = oracle.grad_mul(b0, ko, X, X, a, alpha, beta, npdt(dtype))
            bd = torch.from_numpy(b0.copy()).cuda()
            cg.mul_(bd, K, ad, alpha, beta)
            assert relerr(bd.cpu().numpy(), ref) <= tol, (name, d, n, relerr(bd.cpu().numpy(), ref))
            # dense check against the explicit block matrix (Matrix(K) = K * I)
            if n == 2 and d <= 5:
                MK = oracle.grad_matrix(ko, X, X, npdt(dtype))
                assert np.max(np.abs(MK - MK.T)) < 1e-10
                assert relerr((K @ ad).cpu().numpy(), MK @ a.astype(np.float64)) <= 10 * tol


@pytest.mark.parametrize("d", [1, 3, 8, 32, 48])
def test_gradient_expanded_form_fp64(cg, oracle, d):
    """fp64 isotropic gradient Gramians run the block mul! of src/gradient.jl:86-92 in the expanded form (|x - y|^2 from cached
    norms, 4 instead of 6 fp64 instructions per dimension and pair; csrc/grad_mvm.hpp) while the pre-scaled clouds stay inside
    the radius gate: both forms against the oracle at 1e-12, the gate sends wide / short-lengthscale data to direct
    differences, dot-product kernels never expand; value-gradient blocks too."""
    rng = np.random.default_rng(990 + d)
    n, m = 300, 211
    X = rng.standard_normal((n, d)); Y = rng.standard_normal((m, d)) + 0.3
    a = rng.standard_normal(m * d); y0 = rng.standard_normal(n * d)
    av = rng.standard_normal(m * (d + 1))
    Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    try:
        for k, ko in ((cg.EQ(), oracle.Kernel(oracle.EQ)), (cg.Lengthscale(cg.MaternP(2), 1.7), oracle.Kernel(oracle.MATERNP, p=2, lengthscale=1.7)),
                      (1.5 * cg.RQ(2.0), oracle.Kernel(oracle.RQ, param=2.0, scale=1.5)), (cg.EQ() ** 2, oracle.Kernel(oracle.EQ, power=2))):
            G = cg.gramian(cg.GradientKernel(k), Xd, Yd)
            ref = oracle.grad_mul(y0, ko, X, Y, a, 0.7, -1.1)
            outs = {}
            for ex in (0, 1, -1):
                cg.set_option("grad_expand", ex)
                yd = torch.from_numpy(y0.copy()).cuda(); cg.mul_(yd, G, torch.from_numpy(a).cuda(), 0.7, -1.1)
                auto = d >= 7                                                 # the automatic rule: padded d >= 8 (singular profiles: below)
                assert cg.get_info("last_grad_expand") == (0 if ex == 0 else (1 if ex == 1 else int(auto))), (ex, d)
                outs[ex] = yd.cpu().numpy()
                assert relerr(outs[ex], ref) <= 1e-12, (type(k).__name__, d, ex, relerr(outs[ex], ref))
            assert relerr(outs[1], outs[0]) <= 1e-13
            cg.set_option("grad_expand", 1)
            Gv = cg.gramian(cg.ValueGradientKernel(k), Xd, Yd)
            bv = (Gv @ torch.from_numpy(av).cuda()).cpu().numpy()
            assert cg.get_info("last_grad_expand") == 1
            assert relerr(bv, oracle.valgrad_mul(None, ko, X, Y, av)) <= 1e-12
        cg.set_option("grad_expand", -1)
        # outside the gate: a cloud 100 lengthscales wide falls back to direct differences (and stays accurate)
        Gw = cg.gramian(cg.GradientKernel(cg.Lengthscale(cg.EQ(), 0.02)), Xd, Yd)
        bw = (Gw @ torch.from_numpy(a).cuda()).cpu().numpy()
        assert cg.get_info("last_grad_expand") == 0
        assert relerr(bw, oracle.grad_mul(None, oracle.Kernel(oracle.EQ, lengthscale=0.02), X, Y, a)) <= 1e-12
        # a translated cloud is centred first: still inside the gate, still accurate
        Gs = cg.gramian(cg.GradientKernel(cg.EQ()), Xd + 1.0e4, Yd + 1.0e4)
        bs = (Gs @ torch.from_numpy(a).cuda()).cpu().numpy()
        assert cg.get_info("last_grad_expand") == (1 if d >= 7 else 0)
        assert relerr(bs, oracle.grad_mul(None, oracle.Kernel(oracle.EQ), X + 1.0e4, Y + 1.0e4, a)) <= 1e-11
        # profiles that are singular at s = 0 keep direct differences: the NaN diagonal blocks of gramian(GradientKernel(Exp), x)
        # (inf * 0 in the reference, src/gradient.jl:86-92 with ForwardDiff's derivatives of exp(-sqrt(s)) at 0) stay NaN
        for ks in (cg.Exp(), cg.GammaExponential(1.3), cg.MaternP(0)):
            Gx = cg.gramian(cg.GradientKernel(ks), Xd)
            bx = (Gx @ torch.from_numpy(rng.standard_normal(n * d)).cuda()).cpu().numpy()
            assert cg.get_info("last_grad_expand") == 0 and np.all(np.isnan(bx)), type(ks).__name__
        # dot-product kernels never take it; fp32 does since round 5, inside its own gate of 128 (tests/test_gpu_grad_expand32.py), from d = 8
        (cg.gramian(cg.GradientKernel(cg.EQ()), Xd.float(), Yd.float()) @ torch.from_numpy(a).cuda().float()); assert cg.get_info("last_grad_expand") == (1 if d >= 7 else 0)
        (cg.gramian(cg.GradientKernel(cg.Dot() ** 2), Xd, Yd) @ torch.from_numpy(a).cuda()); assert cg.get_info("last_grad_expand") == 0
    finally:
        cg.set_option("grad_expand", -1)


def test_gradient_eq_block_closed_form(cg):
    """EQ gradient block = k (I - r r') (SURVEY §3.3 known answer)."""
    x = np.array([[0.3, -0.2, 0.5]]); y = np.array([[-0.1, 0.4, 0.2]])
    K = cg.gramian(cg.GradientKernel(cg.EQ()), torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    M = K.to_dense().cpu().numpy()
    r = (x - y)[0]
    assert np.allclose(M, np.exp(-r @ r / 2) * (np.eye(3) - np.outer(r, r)), rtol=1e-13, atol=1e-14)


def test_config4_gradient_eq_d32_f64_subset(cg, oracle):
    """BASELINE.json configs[3] shape (GradientKernel(EQ), d=32, fp64) at an oracle-sized n."""
    rng = np.random.default_rng(0xC0F + 3)
    n, d = 1024, 32
    X = rng.standard_normal((n, d)); a = rng.standard_normal(n * d)
    K = cg.gramian(cg.GradientKernel(cg.EQ()), torch.from_numpy(X).cuda())
    b = (K @ torch.from_numpy(a).cuda()).cpu().numpy()
    assert relerr(b, oracle.grad_mul(None, oracle.Kernel(oracle.EQ), X, X, a)) <= 1e-12


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_toeplitz(cg, oracle, dtype):
    """test/gramian.jl:143-178."""
    tol = 1e-5 if dtype == torch.float32 else 1e-10
    for n in (32, 1000, 4096):
        x = cg.srange(-1, 1, n, dtype)
        for k, ko in ((cg.EQ(), oracle.Kernel(oracle.EQ)), (cg.Exp(), oracle.Kernel(oracle.EXP))):
            G = cg.gramian(k, x)
            assert isinstance(G, cg.SymmetricToeplitz) and tuple(G.shape) == (n, n)
            xs = oracle.srange_points(oracle.srange(-1, 1, n))
            a = np.random.default_rng(n).standard_normal(n).astype(npdt(dtype))
            ref = oracle.matrix(ko, xs, xs) @ a.astype(np.float64)
            b = (G @ torch.from_numpy(a).cuda()).cpu().numpy()
            assert relerr(b, ref) <= tol, (n, relerr(b, ref))
            if n == 32:
                assert relerr(G.to_dense().cpu().numpy(), oracle.matrix(ko, xs, xs)) <= tol
                # periodic boundary conditions -> Circulant
                Cc = cg.gramian(k, x, cg.PeriodicInput())
                assert isinstance(Cc, cg.Circulant)
                vc, _ = oracle.toeplitz_vectors(ko, oracle.srange(-1, 1, n))
                refc = oracle.toeplitz_dense(vc, circulant=True) @ a.astype(np.float64)
                assert relerr((Cc @ torch.from_numpy(a).cuda()).cpu().numpy(), refc) <= tol
                # shifted y with the same step -> non-symmetric Toeplitz
                y = x + 0.37
                Tn = cg.gramian(k, x, y)
                assert isinstance(Tn, cg.Toeplitz)
                ys = xs + 0.37
                assert relerr((Tn @ torch.from_numpy(a).cuda()).cpu().numpy(), oracle.matrix(ko, xs, ys) @ a.astype(np.float64)) <= tol
                # different step -> plain Gramian
                assert isinstance(cg.gramian(k, x, cg.srange(-1, 1, n // 2, dtype)), cg.Gramian)
            # alpha/beta
            y0 = np.random.default_rng(n + 1).standard_normal(n).astype(npdt(dtype))
            yd = torch.from_numpy(y0.copy()).cuda()
            cg.mul_(yd, G, torch.from_numpy(a).cuda(), 0.3, -1.1)
            assert relerr(yd.cpu().numpy(), 0.3 * ref - 1.1 * y0) <= tol


def test_kronecker_and_separable(cg, oracle):
    """test/algebra.jl:70-89, test/separable.jl:9-28 (+ the MVM the reference never asserts)."""
    rng = np.random.default_rng(5)
    n, d = 4, 3
    x = rng.standard_normal(n); y = rng.standard_normal(2 * n)
    gx, gy = cg.LazyGrid(torch.from_numpy(x), d), cg.LazyGrid(torch.from_numpy(y), d)
    p3 = cg.separable("*", *(cg.EQ() for _ in range(d)))
    G = cg.gramian(p3, gx, gy)
    assert isinstance(G, cg.KroneckerProduct) and tuple(G.shape) == (n ** d, (2 * n) ** d)
    ko = oracle.Kernel(oracle.EQ)
    F = oracle.matrix(ko, x, y)
    a = rng.standard_normal((2 * n) ** d)
    ref = oracle.kron_mul(None, [F, F, F], a)
    assert relerr((G @ torch.from_numpy(a).cuda()).cpu().numpy(), ref) <= 1e-12
    assert relerr(G.to_dense().cpu().numpy(), oracle.kron_dense([F, F, F])) <= 1e-12
    # EQ is separable in dimension: the Kronecker Gramian equals the isotropic Gramian on the grid points
    Gx = cg.gramian(cg.EQ(), gx.points("cuda"), gy.points("cuda"))
    assert relerr((Gx @ torch.from_numpy(a).cuda()).cpu().numpy(), ref) <= 1e-11
    # non-identical factors: standard Kronecker order (first factor slowest)
    f1 = rng.standard_normal((3, 5)); f2 = rng.standard_normal((4, 2)); f3 = rng.standard_normal((2, 6))
    Kp = cg.kronecker(*(torch.from_numpy(f).cuda() for f in (f1, f2, f3)))
    av = rng.standard_normal(5 * 2 * 6)
    assert relerr((Kp @ torch.from_numpy(av).cuda()).cpu().numpy(), np.kron(np.kron(f1, f2), f3) @ av) <= 1e-12
    # SeparableKernel: kronecker(K) = gramian(k, x) ⊗ B
    B = rng.standard_normal((3, 3)); B = B.T @ B
    xs = rng.standard_normal(3)
    S = cg.gramian(cg.Separable(cg.EQ(), B), torch.from_numpy(xs).cuda())
    assert tuple(S.shape) == (9, 9)
    MK = np.kron(oracle.matrix(ko, xs, xs), B)
    assert relerr(cg.kronecker(S).to_dense().cpu().numpy(), MK) <= 1e-12
    v = rng.standard_normal(9)
    assert relerr((S @ torch.from_numpy(v).cuda()).cpu().numpy(), MK @ v) <= 1e-12


def test_lowrank_finite_basis(cg, oracle):
    """test/mercer.jl:23-38."""
    rng = np.random.default_rng(6)
    x = rng.standard_normal(16)
    k = cg.FiniteBasis([torch.sin, torch.cos, lambda t: t])
    G = cg.gramian(k, torch.from_numpy(x).cuda())
    assert isinstance(G, cg.LazyMatrixProduct)
    U = np.stack([np.sin(x), np.cos(x), x], axis=1)
    assert relerr(G.to_dense().cpu().numpy(), U @ U.T) <= 1e-13
    a = rng.standard_normal(16); y0 = rng.standard_normal(16)
    yd = torch.from_numpy(y0.copy()).cuda()
    cg.mul_(yd, G, torch.from_numpy(a).cuda(), 1.7, -0.4)
    assert relerr(yd.cpu().numpy(), oracle.lowrank_mul(y0, U, U, a, 1.7, -0.4)) <= 1e-12
    # larger, rectangular, fp32
    n, m, r = 5000, 3000, 7
    xs = rng.standard_normal(n).astype(np.float32); ys = rng.standard_normal(m).astype(np.float32)
    fns = [lambda t, i=i: torch.cos(i * t) for i in range(r)]
    G2 = cg.gramian(cg.FiniteBasis(fns), torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda())
    Un = np.stack([np.cos(i * xs.astype(np.float64)) for i in range(r)], 1); Vn = np.stack([np.cos(i * ys.astype(np.float64)) for i in range(r)], 1)
    av = rng.standard_normal(m).astype(np.float32)
    assert relerr((G2 @ torch.from_numpy(av).cuda()).cpu().numpy(), oracle.lowrank_mul(None, Un, Vn, av)) <= 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_lowrank_gemv_pair_many_slabs(cg, oracle, dtype):
    """The GEMV form over MANY row slabs (csrc/lowrank.hip): for r <= 128 the last slab's workgroup to arrive adds the slab partials inside the
    first-pass kernel (round 4: ticket, fixed order — repeated calls are bit-identical and the counter is back at zero for the next column), for a
    longer z the separate slab-sum kernel runs; one and several columns (each its own pair of passes), ragged n / m / r."""
    rng = np.random.default_rng(71)
    dt = npdt(dtype)
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    for (n, m, r, p) in ((2000, 600011, 40, 1), (3000, 300000, 150, 2), (1537, 1200003, 7, 3), (70000, 70000, 128, 1), (5, 2049, 129, 1)):
        U = rng.standard_normal((n, r)).astype(dt); V = U if n == m else rng.standard_normal((m, r)).astype(dt)
        A = rng.standard_normal((m, p)).astype(dt)
        Ud = torch.from_numpy(U).cuda(); Vd = Ud if V is U else torch.from_numpy(V).cuda()
        L = cg.LazyMatrixProduct(Ud, Vd)
        Ad = torch.from_numpy(A if p > 1 else A[:, 0].copy()).cuda()
        y1 = (L @ Ad).cpu().numpy(); y2 = (L @ Ad).cpu().numpy()
        ref = oracle.lowrank_mul(None, U, V, A if p > 1 else A[:, 0])
        assert relerr(y1, ref) <= tol * (10 if dtype == torch.float32 else 1), (n, m, r, p, relerr(y1, ref))
        assert np.array_equal(y1, y2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_lowrank_matrix_rhs_on_the_matrix_cores(cg, oracle, dtype):
    """LazyMatrixProduct(U, V') with a matrix right-hand side (src/lazy_linear_algebra.jl:78-85): from 8 columns on both
    tall-skinny products run on the matrix cores in the data's own precision (csrc/lowrank.hip); fewer columns take the GEMV
    pair per column.  Ragged n / m / r / p (tiles end past every edge), U is V and U is not V, alpha / beta, NaN in y with
    beta = 0, and the raw C ABI with padded leading dimensions."""
    from covgram import _ffi as f
    import ctypes as C
    rng = np.random.default_rng(61)
    dt = npdt(dtype)
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    for (n, m, r, p) in ((1000, 777, 5, 8), (4099, 3001, 33, 12), (2500, 2500, 64, 40), (1031, 999, 130, 70), (300, 311, 200, 9), (5000, 40000, 17, 3)):
        U = rng.standard_normal((n, r)).astype(dt); V = U if n == m else rng.standard_normal((m, r)).astype(dt)
        A = rng.standard_normal((m, p)).astype(dt); Y0 = rng.standard_normal((n, p)).astype(dt)
        Ud = torch.from_numpy(U).cuda(); Vd = Ud if V is U else torch.from_numpy(V).cuda()
        L = cg.LazyMatrixProduct(Ud, Vd)
        Yd = torch.from_numpy(Y0.copy()).cuda()
        out = cg.mul_(Yd, L, torch.from_numpy(A).cuda(), 1.7, -0.4)
        assert out is Yd
        ref = oracle.lowrank_mul(Y0, U, V, A, 1.7, -0.4)
        for c in range(p):
            assert relerr(Yd.cpu().numpy()[:, c], ref[:, c]) <= tol, (n, m, r, p, c, relerr(Yd.cpu().numpy()[:, c], ref[:, c]))
        Yn = torch.full((n, p), float("nan"), dtype=dtype, device="cuda")
        cg.mul_(Yn, L, torch.from_numpy(A).cuda(), 1.0, 0.0)
        assert relerr(Yn.cpu().numpy(), oracle.lowrank_mul(None, U, V, A)) <= tol
        assert relerr((L @ torch.from_numpy(A).cuda()).cpu().numpy(), oracle.lowrank_mul(None, U, V, A)) <= tol
    # raw ABI, host pointers, leading dimensions larger than the matrices
    n, m, r, p, ldu, ldv, lda, ldy = 700, 520, 19, 11, 704, 528, 523, 777
    U = np.zeros((r, ldu), dt); V = np.zeros((r, ldv), dt); A = np.zeros((p, lda), dt); Y = np.full((p, ldy), 7.0, dt)     # column-major storage
    U[:, :n] = rng.standard_normal((r, n)); V[:, :m] = rng.standard_normal((r, m)); A[:, :m] = rng.standard_normal((p, m)); Y[:, :n] = rng.standard_normal((p, n))
    Y0 = Y.copy()
    P = lambda a: C.c_void_p(a.ctypes.data)
    ctx = cg.get_ctx().bind_stream()
    f.check(f.lib().covgram_lowrank_mvm(ctx, P(U), ldu, P(V), ldv, n, m, r, f.F32 if dtype == torch.float32 else f.F64, P(A), lda, P(Y), ldy, p, -0.8, 0.3, f.HOST))
    ref = oracle.lowrank_mul(Y0[:, :n].T, U[:, :n].T, V[:, :m].T, A[:, :m].T, -0.8, 0.3)
    assert relerr(Y[:, :n].T, ref) <= tol
    assert np.all(Y[:, n:] == 7.0)                                  # the padding rows of y are not touched
    with pytest.raises(f.DimensionMismatch):
        f.check(f.lib().covgram_lowrank_mvm(ctx, P(U), ldu, P(V), ldv, n, m, r, f.F32 if dtype == torch.float32 else f.F64, P(A), m - 1, P(Y), ldy, p, 1.0, 0.0, f.HOST))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("d", [1, 3, 4, 8, 12, 16, 17, 32, 64, 70])
def test_dot_gramian_is_applied_as_x_yt_a(cg, oracle, dtype, d):
    """Gramian(Dot(), x, y) (src/gramian.jl:23,150-151; src/mercer.jl:6-9) is X Y': the library applies it as X (Y' a) — two
    streaming passes instead of the reference's O(n m d) entry loop — for vectors and matrices, alpha / beta, a scaled kernel;
    dense_variant = 1 keeps the entry-by-entry kernel and both agree with the oracle.  Dot()^2 is not a product of thin factors.
    The d list covers the one-right-hand-side kernels of round 5: scalar rows (1, 3, 17, 70), whole-vector rows (4, 8, 12), and the lane-group
    kernels for rows of more than 64 bytes (16 fp64, 32, 64; 16 fp32 stays on the vector rows)."""
    rng = np.random.default_rng(70 + d)
    dt = npdt(dtype)
    tol = TOL[dtype]
    n, m = 1300, 2111
    X = rng.standard_normal((n, d)).astype(dt); Y = rng.standard_normal((m, d)).astype(dt)
    Xd = torch.from_numpy(X).cuda(); Yd = torch.from_numpy(Y).cuda()
    try:
        for k, ko in ((cg.Dot(), oracle.Kernel(oracle.DOT)), (2.5 * cg.Dot(), oracle.Kernel(oracle.DOT, scale=2.5))):
            G = cg.gramian(k, Xd, Yd)
            for p in (1, 3, 9):
                a = rng.standard_normal((m, p) if p > 1 else m).astype(dt); y0 = rng.standard_normal((n, p) if p > 1 else n).astype(dt)
                ref = oracle.mul(y0, ko, X, Y, a, -0.7, 1.3, dt)
                got = {}
                for variant in (0, 1):
                    cg.set_option("dense_variant", variant)
                    yd = torch.from_numpy(y0.copy()).cuda()
                    cg.mul_(yd, G, torch.from_numpy(a).cuda(), -0.7, 1.3)
                    assert cg.get_info("last_dense_path") == (4 if variant == 0 else (1 if d <= 64 else 3)), (variant, d, cg.get_info("last_dense_path"))
                    got[variant] = yd.cpu().numpy()
                    assert relerr(got[variant], ref) <= tol, (variant, d, p, relerr(got[variant], ref))
                cg.set_option("dense_variant", 0)
                yn = torch.full(y0.shape, float("nan"), dtype=dtype, device="cuda")
                cg.mul_(yn, G, torch.from_numpy(a).cuda(), 1.0, 0.0)
                assert relerr(yn.cpu().numpy(), oracle.mul(None, ko, X, Y, a, 1.0, 0.0, dt)) <= tol
        # gramian(x, y) == Gramian(Dot(), x, y) and the symmetric case
        G1 = cg.gramian(Xd, Yd); a = rng.standard_normal(m).astype(dt)
        assert relerr((G1 @ torch.from_numpy(a).cuda()).cpu().numpy(), oracle.mul(None, oracle.Kernel(oracle.DOT), X, Y, a, dtype=dt)) <= tol
        assert cg.get_info("last_dense_path") == 4
        G2 = cg.gramian(cg.Dot() ** 2, Xd, Yd); (G2 @ torch.from_numpy(a).cuda())
        assert cg.get_info("last_dense_path") != 4
    finally:
        cg.set_option("dense_variant", 0)


def test_errors_and_traits(cg):
    x = torch.randn(10, 3, device="cuda", dtype=torch.float64)
    with pytest.raises(cg.DimensionMismatch):
        cg.gramian(cg.EQ(), x, torch.randn(10, 2, device="cuda", dtype=torch.float64))
    G = cg.gramian(cg.EQ(), x)
    with pytest.raises(cg.DimensionMismatch):
        G @ torch.randn(11, device="cuda", dtype=torch.float64)
    Gg = cg.gramian(lambda a, b: 1.0, x)       # GenericInput closure: no device path, no CPU fallback
    with pytest.raises(cg.UnsupportedKernel):
        Gg @ torch.randn(10, device="cuda", dtype=torch.float64)
    assert isinstance(G + torch.ones(10, device="cuda", dtype=torch.float64), cg.LazyMatrixSum)


def test_lazy_sum_with_diagonal(cg, oracle):
    rng = np.random.default_rng(9)
    X = rng.standard_normal((300, 2)); a = rng.standard_normal(300)
    G = cg.gramian(cg.EQ(), torch.from_numpy(X).cuda())
    S = G + 1e-2 * torch.ones(300, device="cuda", dtype=torch.float64)
    b = (S @ torch.from_numpy(a).cuda()).cpu().numpy()
    assert relerr(b, oracle.mul(None, oracle.Kernel(oracle.EQ), X, X, a) + 1e-2 * a) <= 1e-12


def test_cg_solve_through_the_hot_path(cg, oracle):
    """SURVEY §8f-1 / test/gradient.jl:56-63: ‖K (K \\ Ka) - Ka‖ / ‖Ka‖ < 1e-6, for a noisy scalar Gramian and a
    gradient-kernel Gramian; every iteration's MVM runs on the device."""
    rng = np.random.default_rng(21)
    X = rng.standard_normal((400, 2))
    G = cg.gramian(cg.MaternP(2), torch.from_numpy(X).cuda())
    S = G + 1e-2 * torch.ones(400, device="cuda", dtype=torch.float64)            # G + sigma^2 I stays lazy
    bvec = torch.from_numpy(rng.standard_normal(400)).cuda()
    x, info = cg.cg(S, bvec, reltol=1e-10)
    assert info["converged"]
    M = oracle.matrix(oracle.Kernel(oracle.MATERNP, p=2), X) + 1e-2 * np.eye(400)
    assert relerr(x.cpu().numpy(), np.linalg.solve(M, bvec.cpu().numpy())) < 1e-7
    # the same solve with the iteration body replayed as a HIP graph (residual read back every 4 iterations), also
    # with a preconditioner and a start vector
    xg, ig = cg.cg(S, bvec, reltol=1e-10, graph=True, check_every=4)
    # (same recurrence; the iteration count may differ by the read-back granularity and by rounding of this ill-conditioned solve)
    assert ig["converged"] and ig["graph"] and abs(ig["iterations"] - info["iterations"]) <= 4 + info["iterations"] // 8
    assert relerr(xg.cpu().numpy(), np.linalg.solve(M, bvec.cpu().numpy())) < 1e-7
    dinv = 1.0 / torch.from_numpy(np.diag(M).copy()).cuda()
    xg2, ig2 = cg.cg(S, bvec, x0=0.5 * xg, reltol=1e-10, graph=True, precond=lambda r: dinv * r)
    assert ig2["converged"] and relerr(xg2.cpu().numpy(), np.linalg.solve(M, bvec.cpu().numpy())) < 1e-7
    n, d = 16, 3
    Xg = rng.standard_normal((n, d)) / np.sqrt(d)
    K = cg.gramian(cg.GradientKernel(cg.EQ()), torch.from_numpy(Xg).cuda())
    a = torch.from_numpy(rng.standard_normal(n * d)).cuda()
    Ka = K @ a
    xs = cg.solve(K + 1e-8 * torch.ones(n * d, device="cuda", dtype=torch.float64), Ka, reltol=1e-10, maxiter=2000)
    assert float(torch.linalg.norm(K @ xs - Ka) / torch.linalg.norm(Ka)) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [8192, 16384, 20000, 65536, 100003])
def test_toeplitz_four_step_fft_path(cg, oracle, dtype, n):
    """Embeddings N >= 16384 take the transposition-free four-step FFT (csrc/toeplitz.hip): symmetric, shifted
    (non-symmetric) and alpha/beta forms against the numpy circulant-embedding oracle and explicit dense rows."""
    tol = 1e-5 if dtype == torch.float32 else 1e-10
    rng = np.random.default_rng(n)
    x = cg.srange(-1, 1, n, dtype)
    xs = oracle.srange_points(oracle.srange(-1, 1, n))
    a = rng.standard_normal(n).astype(npdt(dtype)); y0 = rng.standard_normal(n).astype(npdt(dtype))
    ad = torch.from_numpy(a).cuda()
    for k, ko in ((cg.Exp(), oracle.Kernel(oracle.EXP)), (cg.Lengthscale(cg.EQ(), 0.05), oracle.Kernel(oracle.EQ, lengthscale=0.05))):
        G = cg.gramian(k, x)
        assert isinstance(G, cg.SymmetricToeplitz)
        vc, _ = oracle.toeplitz_vectors(ko, oracle.srange(-1, 1, n))
        ref = oracle.toeplitz_mul(None, vc, None, a.astype(np.float64))
        b = (G @ ad).cpu().numpy()
        assert relerr(b, ref) <= tol, (n, relerr(b, ref))
        rows = rng.choice(n, 16, replace=False)
        dense_rows = np.array([np.dot(vc[np.abs(i - np.arange(n))], a.astype(np.float64)) for i in rows])
        assert relerr(b[rows], dense_rows) <= tol
        yd = torch.from_numpy(y0.copy()).cuda()
        cg.mul_(yd, G, ad, 0.3, -1.1)
        assert relerr(yd.cpu().numpy(), 0.3 * ref - 1.1 * y0) <= tol
        yn = torch.full((n,), float("nan"), dtype=dtype, device="cuda")       # beta == 0 ignores NaN
        cg.mul_(yn, G, ad, 1.0, 0.0)
        assert relerr(yn.cpu().numpy(), ref) <= tol
    # non-symmetric: shifted y, same step
    Tn = cg.gramian(cg.Exp(), x, x + 0.123)
    assert isinstance(Tn, cg.Toeplitz)
    rg = oracle.srange(-1, 1, n)
    vc2, vr2 = oracle.toeplitz_vectors(oracle.Kernel(oracle.EXP), rg, (rg[0] + 0.123, rg[1], rg[2]))
    assert relerr((Tn @ ad).cpu().numpy(), oracle.toeplitz_mul(None, vc2, vr2, a.astype(np.float64))) <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("d", [65, 100, 256])
def test_dense_wide_dimensions(cg, oracle, dtype, d):
    """d > 64 takes the chunked kernel (csrc/dense_wide.hpp): same direct-difference arithmetic, any d, odd m, matrix RHS."""
    rng = np.random.default_rng(1000 + d)
    n, m = 300, 257
    X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(npdt(dtype)); Y = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(npdt(dtype))
    a = rng.standard_normal(m).astype(npdt(dtype)); A3 = rng.standard_normal((m, 3)).astype(npdt(dtype))
    Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    for name, k, ko in kernel_cases.cases(cg):
        if name not in ("EQ", "MaternP(2)", "RQ(1.0)", "Dot()^3", "ExponentialDot", "2.5*Lengthscale(MaternP(2),1.3)"):
            continue
        G = cg.gramian(k, Xd, Yd)
        assert relerr((G @ torch.from_numpy(a).cuda()).cpu().numpy(), oracle.mul(None, ko, X, Y, a, dtype=npdt(dtype))) <= TOL[dtype], (name, d)
        assert relerr((G @ torch.from_numpy(A3).cuda()).cpu().numpy(), oracle.mul(None, ko, X, Y, A3, dtype=npdt(dtype))) <= TOL[dtype], (name, d)
    # README-sized case: EQ, d = 1024 (README.md:369-395 uses d = 32; the gradient README case has d = 1024)
    if d == 256:
        Xb = rng.standard_normal((2048, 1024)).astype(npdt(dtype)) / 32; ab = rng.standard_normal(2048).astype(npdt(dtype))
        Gb = cg.gramian(cg.EQ(), torch.from_numpy(Xb).cuda())
        rows = rng.choice(2048, 64, replace=False)
        assert relerr((Gb @ torch.from_numpy(ab).cuda()).cpu().numpy()[rows], oracle.mul(None, oracle.Kernel(oracle.EQ), Xb[rows], Xb, ab, dtype=npdt(dtype))) <= TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("d", [65, 100, 256])
def test_gradient_wide_dimensions(cg, oracle, dtype, d):
    """d beyond the lane-owned limit takes the two-kernel panel path (csrc/grad_wide.hpp): rectangular, odd m, alpha/beta."""
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    rng = np.random.default_rng(2000 + d)
    n, m = 70, 45
    X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(npdt(dtype)); Y = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(npdt(dtype))
    a = rng.standard_normal(m * d).astype(npdt(dtype)); b0 = rng.standard_normal(n * d).astype(npdt(dtype))
    Xd, Yd, ad = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda(), torch.from_numpy(a).cuda()
    for name, k, ko in kernel_cases.grad_cases(cg):
        if name not in ("EQ", "MaternP(2)", "RQ(1.0)", "Dot()^3", "ExponentialDot", "2.5*Lengthscale(MaternP(2),1.3)", "EQ^2"):
            continue
        K = cg.gramian(cg.GradientKernel(k), Xd, Yd)
        assert tuple(K.shape) == (n * d, m * d)
        assert relerr((K @ ad).cpu().numpy(), oracle.grad_mul(None, ko, X, Y, a, dtype=npdt(dtype))) <= tol, (name, d)
        bd = torch.from_numpy(b0.copy()).cuda()
        cg.mul_(bd, K, ad, -0.6, 1.4)
        assert relerr(bd.cpu().numpy(), oracle.grad_mul(b0, ko, X, Y, a, -0.6, 1.4, npdt(dtype))) <= tol, (name, d)
    if d == 256:   # the README gradient case shape (README.md:231-245): MaternP(2), d = 1024, n = 1024 (here checked on a row subset)
        nb, db = 1024, 1024
        Xb = (rng.standard_normal((nb, db)) / np.sqrt(db)).astype(npdt(dtype)); ab = rng.standard_normal(nb * db).astype(npdt(dtype))
        Kb = cg.gramian(cg.GradientKernel(cg.MaternP(2)), torch.from_numpy(Xb).cuda())
        got = (Kb @ torch.from_numpy(ab).cuda()).cpu().numpy().reshape(nb, db)
        rows = np.sort(rng.choice(nb, 8, replace=False))
        ref = oracle.grad_mul(None, oracle.Kernel(oracle.MATERNP, p=2), Xb[rows], Xb, ab, dtype=npdt(dtype)).reshape(8, db)
        assert relerr(got[rows], ref) <= tol


# ---- (f)-2: composite kernels and ValueGradientKernel -------------------------------------------------------------------
GOLD = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("d", [1, 3, 8, 70])
def test_composite_kernels_dense_gradient_matrix(cg, oracle, dtype, d):
    """Sum / Product / Power of same-trait kernels (src/algebra.jl:5-63) through every entry point: mul! (vector, 3 RHS),
    Matrix(G), the GradientKernel Gramian; d = 70 takes the wide kernels."""
    tol = TOL[dtype]
    gtol = 1e-5 if dtype == torch.float32 else 1e-12
    rng = np.random.default_rng(0xC0F + 50 + d)
    n, m = 131, 77
    X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(npdt(dtype)); Y = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(npdt(dtype))
    a = rng.standard_normal(m).astype(npdt(dtype)); A3 = rng.standard_normal((m, 3)).astype(npdt(dtype)); y0 = rng.standard_normal(n).astype(npdt(dtype))
    ag = rng.standard_normal(m * d).astype(npdt(dtype)); yg0 = rng.standard_normal(n * d).astype(npdt(dtype))
    Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    for name, k, ko in kernel_cases.composite_cases(cg):
        assert isinstance(cg.device_spec(k), cg._ffi.covgram_kernel_composite)
        G = cg.gramian(k, Xd, Yd)
        yd = torch.from_numpy(y0.copy()).cuda()
        cg.mul_(yd, G, torch.from_numpy(a).cuda(), -0.7, 1.3)
        assert relerr(yd.cpu().numpy(), oracle.mul(y0, ko, X, Y, a, -0.7, 1.3, npdt(dtype))) <= tol, (name, d)
        assert relerr((G @ torch.from_numpy(A3).cuda()).cpu().numpy(), oracle.mul(None, ko, X, Y, A3, dtype=npdt(dtype))) <= tol, (name, d)
        assert relerr(G.to_dense().cpu().numpy(), oracle.matrix(ko, X, Y, npdt(dtype))) <= tol, (name, d)
        K = cg.gramian(cg.GradientKernel(k), Xd, Yd)
        bd = torch.from_numpy(yg0.copy()).cuda()
        cg.mul_(bd, K, torch.from_numpy(ag).cuda(), 0.6, -0.4)
        ref = oracle.grad_mul(yg0, ko, X, Y, ag, 0.6, -0.4, npdt(dtype))
        assert relerr(bd.cpu().numpy(), ref) <= gtol, (name, d, relerr(bd.cpu().numpy(), ref))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_plain_sums_run_term_by_term(cg, oracle, dtype):
    """A Sum of single-profile kernels (src/algebra.jl:5-14) is evaluated as one MVM per term on that term's own path
    (option composite_termwise = 1, the default) — same results as the composite interpreter (= 0) and the oracle, for the
    dense, gradient and value-gradient Gramians, with alpha / beta (beta applies once, NaNs in y are ignored for beta = 0)."""
    o = oracle
    tol = TOL[dtype]; gtol = 1e-5 if dtype == torch.float32 else 1e-12
    rng = np.random.default_rng(0xC0F + 91)
    sums = [("iso", 1.5 * cg.Lengthscale(cg.MaternP(2), 0.7) + 0.5 * cg.Lengthscale(cg.EQ(), 2.0) + 0.25 * cg.RQ(1.5) ** 2,
             o.Composite(((o.Kernel(o.MATERNP, p=2, lengthscale=0.7, scale=1.5),), (o.Kernel(o.EQ, lengthscale=2.0, scale=0.5),),
                          (o.Kernel(o.RQ, param=1.5, power=2, scale=0.25),)), o.ISOTROPIC, 1.0)),
            # constant-only terms (G = c 1 1'): c sum(a) on every row, no derivative entries
            ("iso_plus_constants", 2.0 * (0.3 + cg.EQ() + 0.5 * cg.Cauchy() + 0.2),
             o.Composite(((o.Kernel(o.CONSTANT, scale=0.3),), (o.Kernel(o.EQ),), (o.Kernel(o.CAUCHY, scale=0.5),),
                          (o.Kernel(o.CONSTANT, scale=0.2),)), o.ISOTROPIC, 2.0)),
            kernel_cases.composite_cases(cg)[2]]
    for d in (3, 70):
        n, m = 150, 90
        X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(npdt(dtype)); Y = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(npdt(dtype))
        a = rng.standard_normal(m).astype(npdt(dtype)); y0 = rng.standard_normal(n).astype(npdt(dtype))
        ag = rng.standard_normal(m * d).astype(npdt(dtype)); yg0 = rng.standard_normal(n * d).astype(npdt(dtype))
        av = rng.standard_normal(m * (d + 1)).astype(npdt(dtype)); A3 = rng.standard_normal((m, 3)).astype(npdt(dtype))
        Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
        for name, k, ko in sums:
            want = o.mul(y0, ko, X, Y, a, -0.7, 1.3, npdt(dtype))
            wantg = o.grad_mul(yg0, ko, X, Y, ag, 0.6, -0.4, npdt(dtype))
            wantv = o.valgrad_mul(None, ko, X, Y, av, 1.0, 0.0, npdt(dtype))
            try:
                for tw in (1, 0):
                    cg.set_option("composite_termwise", tw)
                    yd = torch.from_numpy(y0.copy()).cuda()
                    cg.mul_(yd, cg.gramian(k, Xd, Yd), torch.from_numpy(a).cuda(), -0.7, 1.3)
                    assert relerr(yd.cpu().numpy(), want) <= tol, (name, d, tw)
                    got3 = (cg.gramian(k, Xd, Yd) @ torch.from_numpy(A3).cuda()).cpu().numpy()
                    assert relerr(got3, o.mul(None, ko, X, Y, A3, dtype=npdt(dtype))) <= tol, (name, d, tw)
                    bd = torch.from_numpy(yg0.copy()).cuda()
                    cg.mul_(bd, cg.gramian(cg.GradientKernel(k), Xd, Yd), torch.from_numpy(ag).cuda(), 0.6, -0.4)
                    assert relerr(bd.cpu().numpy(), wantg) <= gtol, (name, d, tw, relerr(bd.cpu().numpy(), wantg))
                    bv = torch.full((n * (d + 1),), float("nan"), dtype=dtype, device="cuda")
                    cg.mul_(bv, cg.gramian(cg.ValueGradientKernel(k), Xd, Yd), torch.from_numpy(av).cuda(), 1.0, 0.0)
                    assert relerr(bv.cpu().numpy(), wantv) <= gtol, (name, d, tw, relerr(bv.cpu().numpy(), wantv))
            finally:
                cg.set_option("composite_termwise", 1)


@pytest.mark.parametrize("d", [2, 5, 12])
def test_product_kernels_on_the_matrix_cores(cg, oracle, d):
    """Products / powers of smooth profiles (src/algebra.jl:28-63) as ONE composite on the generic matrix-core kernels (general,
    LDS-shared for d > 8, symmetric), against the oracle and the direct-difference interpreter; a product with a profile that is
    not smooth in s (Exp) stays on direct differences."""
    o = oracle
    rng = np.random.default_rng(7000 + d)
    prods = [(cg.EQ() * cg.Lengthscale(cg.Cauchy(), 1.5), o.Composite(((o.Kernel(o.EQ), o.Kernel(o.CAUCHY, lengthscale=1.5)),), o.ISOTROPIC, 1.0)),
             (1.7 * cg.Lengthscale(cg.MaternP(2), 1.4) * cg.RQ(1.5) ** 2,
              o.Composite(((o.Kernel(o.MATERNP, p=2, lengthscale=1.4), o.Kernel(o.RQ, param=1.5, power=2)),), o.ISOTROPIC, 1.7)),
             (cg.Dot() ** 2 * cg.ExponentialDot(), o.Composite(((o.Kernel(o.DOT, power=2), o.Kernel(o.EXPDOT)),), o.DOTPRODUCT, 1.0))]
    try:
        for n, m in ((200, 333), (777, 777)):
            X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
            Y = X if n == m else (rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)
            a = rng.standard_normal(m).astype(np.float32); A3 = rng.standard_normal((m, 3)).astype(np.float32)
            Xd = torch.from_numpy(X).cuda(); Yd = Xd if n == m else torch.from_numpy(Y).cuda()
            for k, ko in prods:
                G = cg.gramian(k, Xd) if n == m else cg.gramian(k, Xd, Yd)
                ref = o.mul(None, ko, X, Y, a, dtype=np.float32); ref3 = o.mul(None, ko, X, Y, A3, dtype=np.float32)
                got = {}
                for variant, sym, lds in ((1, 0, -1), (2, 0, 0), (2, 0, 1), (2, 1, -1)):
                    cg.set_option("dense_variant", variant); cg.set_option("mfma_sym", sym); cg.set_option("mfma_lds", lds)
                    got[(variant, sym, lds)] = (G @ torch.from_numpy(a).cuda()).cpu().numpy()
                    assert cg.get_info("last_dense_path") == (1 if variant == 1 else 2)
                    assert cg.get_info("last_mfma_sym") == (1 if (sym == 1 and n == m) else 0)
                    assert relerr(got[(variant, sym, lds)], ref) <= 1e-5, (d, n, variant, sym, lds, relerr(got[(variant, sym, lds)], ref))
                    assert relerr((G @ torch.from_numpy(A3).cuda()).cpu().numpy(), ref3) <= 1e-5
                assert np.array_equal(got[(2, 0, 0)], got[(2, 0, 1)])
        cg.set_option("dense_variant", 0); cg.set_option("mfma_sym", -1); cg.set_option("mfma_lds", -1)
        (cg.gramian(cg.EQ() * cg.Exp(), Xd) @ torch.from_numpy(a).cuda()); assert cg.get_info("last_dense_path") == 1
    finally:
        cg.set_option("dense_variant", 0); cg.set_option("mfma_sym", -1); cg.set_option("mfma_lds", -1)


def test_composite_golden_and_toeplitz(cg, oracle):
    g = np.load(f"{GOLD}/composite.npz")
    for d in (1, 3, 8):
        for (n, m) in ((4, 6), (65, 33)):
            tag = f"d{d}_n{n}_m{m}"
            X, Y, a, y0, ag, yg0 = (g[f"{tag}_{s}"] for s in ("X", "Y", "a", "y0", "ag", "yg0"))
            alpha, beta = g[f"{tag}_ab"]
            Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
            for name, k, ko in kernel_cases.composite_cases(cg):
                yd = torch.from_numpy(y0.copy()).cuda()
                cg.mul_(yd, cg.gramian(k, Xd, Yd), torch.from_numpy(a).cuda(), alpha, beta)
                assert relerr(yd.cpu().numpy(), g[f"{tag}_{name}_b"]) <= 1e-12, (name, tag)
                bd = torch.from_numpy(yg0.copy()).cuda()
                cg.mul_(bd, cg.gramian(cg.GradientKernel(k), Xd, Yd), torch.from_numpy(ag).cuda(), alpha, beta)
                assert relerr(bd.cpu().numpy(), g[f"{tag}_{name}_bg"]) <= 1e-12, (name, tag)
    # a composite isotropic kernel on a range is still a SymmetricToeplitz (src/gramian.jl:167-176)
    name, k, ko = kernel_cases.composite_cases(cg)[0]
    n = 2000
    T = cg.gramian(k, cg.srange(-1, 1, n))
    assert isinstance(T, cg.SymmetricToeplitz)
    a = np.random.default_rng(5).standard_normal(n)
    x = oracle.srange_points(oracle.srange(-1, 1, n))
    assert relerr((T @ torch.from_numpy(a).cuda()).cpu().numpy(), oracle.mul(None, ko, x, x, a)) <= 1e-10


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("d", [1, 3, 8, 32])
def test_value_gradient_kernel(cg, oracle, dtype, d):
    """ValueGradientKernel Gramian (src/gradient.jl:400-474), blocks of d+1 — test/gradient.jl:87-125: mul! with α, β
    against the explicit block matrix, symmetric and rectangular, single-chunk and split-J launches."""
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    for (n, m) in ((2, 2), (33, 33), (300, 700)):
        rng = np.random.default_rng(1000 * d + n)
        X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(npdt(dtype))
        Y = X if n == m and n < 100 else (rng.standard_normal((m, d)) / np.sqrt(d)).astype(npdt(dtype))
        a = rng.standard_normal(m * (d + 1)).astype(npdt(dtype)); b0 = rng.standard_normal(n * (d + 1)).astype(npdt(dtype))
        alpha, beta = rng.standard_normal(2)
        Xd = torch.from_numpy(X).cuda(); Yd = Xd if Y is X else torch.from_numpy(Y).cuda()
        for name, k, ko in kernel_cases.valgrad_cases(cg):
            K = cg.gramian(cg.ValueGradientKernel(k), Xd, None if Y is X else Yd)
            assert isinstance(K, cg.BlockGramian) and tuple(K.shape) == (n * (d + 1), m * (d + 1))
            ref = oracle.valgrad_mul(b0, ko, X, Y, a, alpha, beta, npdt(dtype))
            bd = torch.from_numpy(b0.copy()).cuda()
            cg.mul_(bd, K, torch.from_numpy(a).cuda(), alpha, beta)
            assert relerr(bd.cpu().numpy(), ref) <= tol, (name, d, n, relerr(bd.cpu().numpy(), ref))
            if n == 2 and d <= 3:
                MK = oracle.valgrad_matrix(ko, X, Y, npdt(dtype))
                assert relerr((K @ torch.from_numpy(a).cuda()).cpu().numpy(), MK @ a.astype(np.float64)) <= 10 * tol
                assert relerr(K.to_dense().cpu().numpy(), MK) <= 10 * tol
            # beta == 0 ignores NaN in the output
            yn = torch.full((n * (d + 1),), float("nan"), dtype=dtype, device="cuda")
            cg.mul_(yn, K, torch.from_numpy(a).cuda(), 1.0, 0.0)
            assert relerr(yn.cpu().numpy(), oracle.valgrad_mul(None, ko, X, Y, a, 1.0, 0.0, npdt(dtype))) <= tol, (name, d, n)


def test_value_gradient_golden_and_limits(cg, oracle):
    g = np.load(f"{GOLD}/composite.npz")
    byname = {c[0]: c for c in kernel_cases.valgrad_cases(cg)}
    alias = {"EQ": "EQ", "RQ1": "RQ(1.0)", "MaternP2": "MaternP(2)", "Dot3": "Dot()^3", "ExpDot": "ExponentialDot", "EQ_l07": "Lengthscale(EQ,0.7)"}
    for d in (1, 3, 8):
        for (n, m) in ((4, 6), (65, 33)):
            tag = f"d{d}_n{n}_m{m}"
            X, Y, av, yv0 = (g[f"{tag}_{s}"] for s in ("X", "Y", "av", "yv0"))
            alpha, beta = g[f"{tag}_ab"]
            Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
            for gname in list(g["valgrad_names"]) + list(g["names"]):
                k = byname[alias.get(str(gname), str(gname))][1]
                bd = torch.from_numpy(yv0.copy()).cuda()
                cg.mul_(bd, cg.gramian(cg.ValueGradientKernel(k), Xd, Yd), torch.from_numpy(av).cuda(), alpha, beta)
                assert relerr(bd.cpu().numpy(), g[f"{tag}_{gname}_bv"]) <= 1e-12, (gname, tag)
    # beyond the lane-per-row limit (d > 48 fp64 / 64 fp32) the panel kernels of grad_wide.hpp carry the value row as well
    rng = np.random.default_rng(31)
    for dt, tol in ((np.float64, 1e-12), (np.float32, 1e-5)):
        for d in (70, 130):
            n, m = 90, 150
            X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(dt); Y = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(dt)
            a = rng.standard_normal(m * (d + 1)).astype(dt); b0 = rng.standard_normal(n * (d + 1)).astype(dt)
            for name in ("EQ", "MaternP(2)", "Dot()^3", "iso_sum_of_products"):
                _, k, ko = byname[name]
                K = cg.gramian(cg.ValueGradientKernel(k), torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
                bd = torch.from_numpy(b0.copy()).cuda()
                cg.mul_(bd, K, torch.from_numpy(a).cuda(), 0.8, -0.3)
                ref = oracle.valgrad_mul(b0, ko, X, Y, a, 0.8, -0.3, dt)
                assert relerr(bd.cpu().numpy(), ref) <= tol, (name, d, dt, relerr(bd.cpu().numpy(), ref))


# ---- the matrix-core EQ path (dense_mfma.hip) ---------------------------------------------------------------------------
@pytest.mark.parametrize("d", [1, 2, 3, 5, 8, 12, 16, 24, 32])
def test_eq_matrix_core_path_matches_direct_differences(cg, oracle, d):
    """fp32 EQ: the bf16x3-split MFMA path against the fp64 oracle and against the direct-difference kernel, ragged
    shapes (tile padding on both sides), one and two row tiles per wave, alpha/beta, lengthscale and scale."""
    rng = np.random.default_rng(77 + d)
    k = 1.7 * cg.Lengthscale(cg.EQ(), 0.9)
    ko = oracle.Kernel(oracle.EQ, lengthscale=0.9, scale=1.7)
    try:
        for (n, m) in ((1, 1), (33, 31), (257, 1000), (1500, 700), (300, 3000)):
            X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32); Y = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)
            a = rng.standard_normal(m).astype(np.float32); y0 = rng.standard_normal(n).astype(np.float32)
            G = cg.gramian(k, torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
            ref = oracle.mul(y0, ko, X, Y, a, -0.7, 1.3, np.float32)
            outs = {}
            # both splits of the coordinates (round 4): the sweep below under the fp16 two-way split (option mfma_f16 = 1: one MFMA per four
            # coordinates), then, last, under the bf16 three-way split (0); the comparisons behind the sweep see the bf16 results
            for f16 in (1, 0):
              cg.set_option("mfma_f16", f16)
              # (dense_variant, rows_per_lane, mfma_lds, jsplit): mfma_lds = 1 makes four waves share the column tiles through
            # LDS (d <= 8 with two row tiles per wave — four with rows_per_lane = 4 at d = 5 ... 8 —, d > 8 with one), jsplit = 3 gives ragged column chunks (stage counts
            # not a multiple of the 4 waves / odd tile counts)
              for variant, rpl, lds, js in ((1, 0, -1, 0), (2, 1, 0, 0), (2, 1, 1, 0), (2, 1, 1, 3), (2, 2, 0, 0), (2, 2, 1, 0), (2, 2, 1, 3), (2, 4, 1, 0), (2, 4, 1, 3), (0, 0, -1, 0)):
                cg.set_option("dense_variant", variant); cg.set_option("rows_per_lane", rpl); cg.set_option("mfma_lds", lds); cg.set_option("jsplit", js)
                yd = torch.from_numpy(y0.copy()).cuda()
                cg.mul_(yd, G, torch.from_numpy(a).cuda(), -0.7, 1.3)
                assert cg.get_info("last_dense_path") == (1 if variant == 1 else 2), (variant, d)
                assert variant == 1 or cg.get_info("last_mfma_f16") == f16, (variant, d, f16)
                outs[(("h",) if f16 else ()) + (variant, rpl, lds, js)] = yd.cpu().numpy()
                assert relerr(yd.cpu().numpy(), ref) <= 1e-5, (f16, variant, rpl, lds, js, d, n, m, relerr(yd.cpu().numpy(), ref))
            assert np.array_equal(outs[("h", 2, 2, 0, 0)], outs[("h", 2, 2, 1, 0)]) and np.array_equal(outs[("h", 2, 1, 0, 0)], outs[("h", 2, 1, 1, 0)])
            assert np.array_equal(outs[("h", 2, 4, 1, 3)], outs[("h", 2, 2, 1, 3)])
            assert relerr(outs[("h", 2, 2, 0, 0)], outs[(2, 2, 0, 0)]) <= 2e-6      # the two splits agree far inside the tolerance
            assert relerr(outs[(2, 2, 0, 0)], outs[(1, 0, -1, 0)]) <= 5e-6
            assert np.array_equal(outs[(2, 2, 0, 0)], outs[(2, 2, 1, 0)])      # same tiles, same order: bit-identical
            assert np.array_equal(outs[(2, 1, 0, 0)], outs[(2, 1, 1, 0)])      # (d > 8: one tile per stage, slices split over the waves)
            assert np.array_equal(outs[(2, 4, 1, 3)], outs[(2, 2, 1, 3)])      # four row tiles per wave (d = 5 ... 8; elsewhere the option means two): same column chunks, same sums
            # beta == 0 ignores NaN in y
            cg.set_option("dense_variant", 2); cg.set_option("rows_per_lane", 0); cg.set_option("mfma_lds", -1); cg.set_option("jsplit", 0)
            yn = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
            cg.mul_(yn, G, torch.from_numpy(a).cuda(), 1.0, 0.0)
            assert relerr(yn.cpu().numpy(), oracle.mul(None, ko, X, Y, a, 1.0, 0.0, np.float32)) <= 1e-5
    finally:
        cg.set_option("dense_variant", 0); cg.set_option("rows_per_lane", 0); cg.set_option("mfma_lds", -1); cg.set_option("jsplit", 0); cg.set_option("mfma_f16", -1)


@pytest.mark.parametrize("d", [3, 7, 20])
def test_eq_matrix_core_lds_sharing_at_size(cg, oracle, d):
    """Shapes at which the library itself turns on the LDS-shared column tiles (long column chunks, >= 64 row tiles): ragged
    n and m, row blocks of 256 / 128 rows that end past n; checked on the first, last and some middle rows against the fp64
    oracle, and bit-for-bit against the one-wave-per-workgroup kernel."""
    rng = np.random.default_rng(400 + d)
    n, m = (20011, 131072 - 37) if d != 3 else (65600 + 21, 33000 + 5)      # d = 3: >= 1024 row tiles -> eight waves per workgroup
    X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32); Y = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)
    a = rng.standard_normal(m).astype(np.float32)
    G = cg.gramian(cg.Lengthscale(cg.EQ(), 0.8), torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
    ad = torch.from_numpy(a).cuda()
    try:
        # d = 20 under the bf16 three-way split: K2 = 12, the one-tile-per-stage LDS form (under the fp16 split d = 20 is K2 = 6 and the planner
        # keeps its shorter column chunks off LDS at this size); d = 3, 7 run the default, the fp16 two-way split
        cg.set_option("mfma_f16", 0 if d == 20 else -1)
        cg.set_option("rows_per_lane", 2)       # d = 7: the automatic choice is four row tiles per wave (other column chunks: checked below)
        cg.set_option("mfma_lds", 0); b0 = (G @ ad).cpu().numpy()
        assert cg.get_info("last_mfma_lds") == 0
        cg.set_option("mfma_lds", -1); b1 = (G @ ad).cpu().numpy()
        assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_lds") == 1
        cg.set_option("mfma_lds", 1); b2 = (G @ ad).cpu().numpy()
        cg.set_option("rows_per_lane", 0); cg.set_option("mfma_lds", -1); b3 = (G @ ad).cpu().numpy()
        assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_lds") == 1
    finally:
        cg.set_option("mfma_lds", -1); cg.set_option("rows_per_lane", 0); cg.set_option("mfma_f16", -1)
    rows = np.r_[0:40, n // 2:n // 2 + 40, n - 40:n]
    want = oracle.mul(None, oracle.Kernel(oracle.EQ, lengthscale=0.8), X[rows], Y, a, dtype=np.float32)
    assert relerr(b1[rows], want) <= 1e-5 and relerr(b3[rows], want) <= 1e-5
    assert np.isfinite(b3).all() and relerr(b3, b1) <= 2e-6
    assert np.array_equal(b0, b1) and np.array_equal(b1, b2)
    assert np.isfinite(b1).all()


@pytest.mark.parametrize("d", [1, 2, 3, 5, 8, 9, 12, 16, 24, 32])
def test_eq_symmetric_matrix_core_kernel(cg, oracle, d):
    """gramian(EQ, x) on the matrix cores with the upper triangle evaluated once (row sums + column sums of the same tiles,
    mfma_sym = 1; d > 8: the 128-row-panel kernel with one tile per stage): ragged n (row tiles, panels and stages all end
    past n), ragged column chunks (jsplit),
    alpha / beta, NaN in y with beta = 0, against the fp64 oracle and against the full kernel; a Gramian of two different
    point sets never takes it."""
    rng = np.random.default_rng(900 + d)
    k = 1.3 * cg.Lengthscale(cg.EQ(), 0.9)
    ko = oracle.Kernel(oracle.EQ, lengthscale=0.9, scale=1.3)
    try:
        for n in (1, 31, 33, 257, 1000, 2309):
            X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
            a = rng.standard_normal(n).astype(np.float32); y0 = rng.standard_normal(n).astype(np.float32)
            Xd = torch.from_numpy(X).cuda(); ad = torch.from_numpy(a).cuda()
            G = cg.gramian(k, Xd)
            ref = oracle.mul(y0, ko, X, X, a, -0.7, 1.3, np.float32)
            cg.set_option("mfma_sym", 0)
            yf = torch.from_numpy(y0.copy()).cuda(); cg.mul_(yf, G, ad, -0.7, 1.3)
            assert cg.get_info("last_mfma_sym") == 0
            for js, f16 in ((0, 1), (1, 1), (3, 1), (0, 0), (1, 0), (3, 0)):      # fp16 two-way / bf16 three-way split of the coordinates (round 4)
                cg.set_option("mfma_sym", 1); cg.set_option("jsplit", js); cg.set_option("mfma_f16", f16)
                ys = torch.from_numpy(y0.copy()).cuda(); cg.mul_(ys, G, ad, -0.7, 1.3)
                assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_sym") == 1 and cg.get_info("last_mfma_f16") == f16
                assert relerr(ys.cpu().numpy(), ref) <= 1e-5, (d, n, js, f16, relerr(ys.cpu().numpy(), ref))
                assert relerr(ys.cpu().numpy(), yf.cpu().numpy()) <= 5e-6
            cg.set_option("jsplit", 0); cg.set_option("mfma_f16", -1)
            yn = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
            cg.mul_(yn, G, ad, 1.0, 0.0)
            assert relerr(yn.cpu().numpy(), oracle.mul(None, ko, X, X, a, 1.0, 0.0, np.float32)) <= 1e-5
            # two point sets (even equal values in different buffers): the general kernel
            G2 = cg.gramian(k, Xd, Xd.clone()); (G2 @ ad)
            assert cg.get_info("last_mfma_sym") == 0
    finally:
        cg.set_option("mfma_sym", -1); cg.set_option("jsplit", 0); cg.set_option("mfma_f16", -1)


def test_eq_symmetric_partial_products_sum_to_the_mvm(cg, oracle):
    """covgram_mvm_sym_partial (the multi-GPU form): rank r of P takes the upper-triangle tiles of the panels p = r (mod P) and
    their mirror images; emulated on one GPU, the partials of all ranks must sum to G a — also when there are fewer panels
    than ranks — and a kernel / point set the symmetric kernel does not serve must say so."""
    rng = np.random.default_rng(31)
    try:
        cg.set_option("mfma_sym", 1)
        for n, d in ((1500, 3), (5, 2), (777, 8)):
            X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32); a = rng.standard_normal(n).astype(np.float32)
            Xd = torch.from_numpy(X).cuda(); ad = torch.from_numpy(a).cuda()
            G = cg.gramian(2.0 * cg.EQ(), Xd)
            assert G.sym_partial_supported()
            want = oracle.mul(None, oracle.Kernel(oracle.EQ, scale=2.0), X, X, a, dtype=np.float32)
            for world in (1, 2, 3, 8):
                tot = torch.zeros(n, dtype=torch.float32, device="cuda")
                for r in range(world):
                    part = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
                    G.sym_partial_(part, ad, r, world)
                    tot += part
                assert relerr(tot.cpu().numpy(), want) <= 1e-5, (n, d, world, relerr(tot.cpu().numpy(), want))
        assert cg.gramian(cg.MaternP(2), Xd).sym_partial_supported()             # every smooth matrix-core profile
        assert cg.gramian(cg.Exp(), Xd).sym_partial_supported()                    # not differentiable in s at 0: the DIRECT-difference symmetric kernel (round 4, tests/test_gpu_sym32.py)
        assert cg.gramian(cg.EQ(), Xd.double()).sym_partial_supported()          # fp64: 64-row blocks of the direct-difference kernel (tests/test_gpu_sym_partial64.py)
        assert not cg.gramian(cg.EQ(), Xd, Xd.clone()).sym_partial_supported()
        cg.set_option("mfma_sym", -1)
        assert not cg.gramian(cg.EQ(), Xd).sym_partial_supported()          # below the size from which it pays
    finally:
        cg.set_option("mfma_sym", -1)


@pytest.mark.parametrize("d", [1, 3, 6, 7, 8, 12, 20, 31])
def test_generic_symmetric_matrix_core_kernels(cg, oracle, d):
    """The other matrix-core profiles on gramian(k, x) with the upper triangle evaluated once (mfma_sym = 1): RQ, Cauchy, IMQ,
    MaternP(1..3), Dot^p, ExponentialDot, EQ^2 — against the fp64 oracle and against the general matrix-core kernel, ragged n,
    alpha / beta; the multi-GPU partial form summed over emulated ranks."""
    o = oracle
    rng = np.random.default_rng(1200 + d)
    kernels = [(cg.Lengthscale(cg.RQ(1.5), 1.3), o.Kernel(o.RQ, param=1.5, lengthscale=1.3)),
               (2.0 * cg.Cauchy(), o.Kernel(o.CAUCHY, scale=2.0)),
               (cg.InverseMultiQuadratic(1.2), o.Kernel(o.IMQ, param=1.2)),
               (cg.MaternP(1), o.Kernel(o.MATERNP, p=1)), (cg.Lengthscale(cg.MaternP(2), 1.5), o.Kernel(o.MATERNP, p=2, lengthscale=1.5)),
               (cg.MaternP(3), o.Kernel(o.MATERNP, p=3)),
               (cg.Dot() ** 3, o.Kernel(o.DOT, power=3)), (cg.ExponentialDot(), o.Kernel(o.EXPDOT)), (cg.EQ() ** 2, o.Kernel(o.EQ, power=2))]
    try:
        for n in (33, 700, 1601):
            X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
            a = rng.standard_normal(n).astype(np.float32); y0 = rng.standard_normal(n).astype(np.float32)
            Xd = torch.from_numpy(X).cuda(); ad = torch.from_numpy(a).cuda()
            for k, ko in kernels:
                G = cg.gramian(k, Xd)
                ref = o.mul(y0, ko, X, X, a, 0.8, -0.6, np.float32)
                cg.set_option("mfma_sym", 0); cg.set_option("dense_variant", 2)
                yf = torch.from_numpy(y0.copy()).cuda(); cg.mul_(yf, G, ad, 0.8, -0.6)
                assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_sym") == 0
                cg.set_option("mfma_sym", 1)
                ys = torch.from_numpy(y0.copy()).cuda(); cg.mul_(ys, G, ad, 0.8, -0.6)
                assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_sym") == 1, type(k).__name__
                assert relerr(ys.cpu().numpy(), ref) <= 1e-5, (type(k).__name__, d, n, relerr(ys.cpu().numpy(), ref))
                assert relerr(ys.cpu().numpy(), yf.cpu().numpy()) <= 5e-6
                if n == 700:
                    assert G.sym_partial_supported()
                    tot = torch.zeros(n, dtype=torch.float32, device="cuda"); part = torch.empty_like(tot)
                    for r in range(3):
                        G.sym_partial_(part, ad, r, 3); tot += part
                    assert relerr(tot.cpu().numpy(), o.mul(None, ko, X, X, a, dtype=np.float32)) <= 1e-5
    finally:
        cg.set_option("mfma_sym", -1); cg.set_option("dense_variant", 0)


def test_eq_symmetric_kernel_at_size(cg, oracle):
    """The sizes at which the library picks the symmetric kernel by itself: sampled rows against the oracle."""
    rng = np.random.default_rng(77)
    n, d = 50021, 3
    X = rng.standard_normal((n, d)).astype(np.float32); a = rng.standard_normal(n).astype(np.float32)
    G = cg.gramian(cg.EQ(), torch.from_numpy(X).cuda())
    b = (G @ torch.from_numpy(a).cuda()).cpu().numpy()
    assert cg.get_info("last_mfma_sym") == 1
    rows = np.r_[0:50, n // 2:n // 2 + 50, n - 50:n]
    assert relerr(b[rows], oracle.mul(None, oracle.Kernel(oracle.EQ), X[rows], X, a, dtype=np.float32)) <= 1e-5
    assert np.isfinite(b).all()
    # the thresholds by profile cost (csrc/common.hpp MFMA_SYM_MIN_N_*): EQ d <= 4 from 18000, wider EQ from 15000, MaternP / RQ from 12500 (d <= 4) / 10000 (wider points, round 5)
    for kern, ko, nn, dd, want in ((cg.EQ(), oracle.Kernel(oracle.EQ), 17000, 3, 0), (cg.EQ(), oracle.Kernel(oracle.EQ), 19000, 3, 1),
                                   (cg.EQ(), oracle.Kernel(oracle.EQ), 16000, 8, 1), (cg.MaternP(2), oracle.Kernel(oracle.MATERNP, p=2), 9000, 12, 0),
                                   (cg.MaternP(2), oracle.Kernel(oracle.MATERNP, p=2), 11000, 12, 1),
                                   # (MaternP at d <= 4: gramian(k, x) from the threshold on the symmetric matrix-core kernel — its order is decided once per tile since late
                                   # round 4 —, below it and for two point sets packed on the lane-per-row kernels)
                                   (cg.MaternP(2), oracle.Kernel(oracle.MATERNP, p=2), 13000, 3, 1), (cg.MaternP(2), oracle.Kernel(oracle.MATERNP, p=2), 12000, 3, 0),
                                   (cg.RQ(1.5), oracle.Kernel(oracle.RQ, param=1.5), 13000, 3, 1)):
        Xs = rng.standard_normal((nn, dd)).astype(np.float32); as_ = rng.standard_normal(nn).astype(np.float32)
        bb = (cg.gramian(kern, torch.from_numpy(Xs).cuda()) @ torch.from_numpy(as_).cuda()).cpu().numpy()
        assert cg.get_info("last_mfma_sym") == want, (type(kern).__name__, nn, dd)
        rr = np.r_[0:40, nn - 40:nn]
        assert relerr(bb[rr], oracle.mul(None, ko, Xs[rr], Xs, as_, dtype=np.float32)) <= 1e-5


def test_eq_matrix_core_gate(cg, oracle):
    """The expanded exponent is only used while max(max|x~|, max|y~|)^2 <= 126, x~ = (x - c) / l relative to the column set's own centre c
    (a sample mean: here, with n <= 1024, the mean of all points): wide or short-lengthscale data falls back to direct differences (and stays accurate), a translation
    changes nothing; fp64 and the profiles that are not smooth in s never take it."""
    rng = np.random.default_rng(5)
    n, d = 600, 3
    X0 = rng.standard_normal((n, d)).astype(np.float32)
    a = rng.standard_normal(n).astype(np.float32)
    ad = torch.from_numpy(a).cuda()
    # centred unit-scale data: eligible
    G = cg.gramian(cg.EQ(), torch.from_numpy(X0).cuda()); b = (G @ ad).cpu().numpy()
    assert cg.get_info("last_dense_path") == 2 and relerr(b, oracle.mul(None, oracle.Kernel(oracle.EQ), X0, X0, a, dtype=np.float32)) <= 1e-5
    # right below the gate (P = max|x~| max|y~| = 125): still far inside the fp32 tolerance
    s = np.sqrt(125.0 / (1.4426950408889634 * float(((X0.astype(np.float64) - X0.astype(np.float64).mean(0).astype(np.float32)) ** 2).sum(1).max())))
    Xg = (X0 * s).astype(np.float32)
    G = cg.gramian(cg.EQ(), torch.from_numpy(Xg).cuda()); b = (G @ ad).cpu().numpy()
    assert cg.get_info("last_dense_path") == 2 and relerr(b, oracle.mul(None, oracle.Kernel(oracle.EQ), Xg, Xg, a, dtype=np.float32)) <= 2e-6
    # the largest exponent the path can meet is x~_i . x~_i = P on the diagonal: exp2(125.9) is finite in fp32 (no inf / NaN)
    s2 = np.sqrt(125.9 / (1.4426950408889634 * float(((X0.astype(np.float64) - X0.astype(np.float64).mean(0).astype(np.float32)) ** 2).sum(1).max())))
    Xe = (X0 * s2).astype(np.float32)
    for sym in (0, 1):
        cg.set_option("mfma_sym", sym)
        G = cg.gramian(cg.EQ(), torch.from_numpy(Xe).cuda()); b = (G @ ad).cpu().numpy()
        if cg.get_info("last_dense_path") == 2:      # (float rounding of the bound may already send it to the exact kernel)
            assert np.isfinite(b).all() and relerr(b, oracle.mul(None, oracle.Kernel(oracle.EQ), Xe, Xe, a, dtype=np.float32)) <= 2e-6
    cg.set_option("mfma_sym", -1)
    # right above it: the exact kernel
    Xa = (X0 * (s * 1.02)).astype(np.float32)
    G = cg.gramian(cg.EQ(), torch.from_numpy(Xa).cuda()); b = (G @ ad).cpu().numpy()
    assert cg.get_info("last_dense_path") == 1 and relerr(b, oracle.mul(None, oracle.Kernel(oracle.EQ), Xa, Xa, a, dtype=np.float32)) <= 1e-5
    # the same cloud shifted far from the origin (|x|^2 ~ 3e4): both sides are centred first, so the gate and the accuracy are unchanged
    Xs = (X0 + 100.0).astype(np.float32)
    G = cg.gramian(cg.EQ(), torch.from_numpy(Xs).cuda()); b = (G @ ad).cpu().numpy()
    assert cg.get_info("last_dense_path") == 2 and relerr(b, oracle.mul(None, oracle.Kernel(oracle.EQ), Xs, Xs, a, dtype=np.float32)) <= 1e-5
    # two clouds far from EACH OTHER: |x - c_Y| is large whatever the centre -> exact kernel (all entries underflow to 0 here)
    Ys = (X0 - 100.0).astype(np.float32)
    G = cg.gramian(cg.EQ(), torch.from_numpy(Xs).cuda(), torch.from_numpy(Ys).cuda()); b = (G @ ad).cpu().numpy()
    assert cg.get_info("last_dense_path") == 1 and np.all(b == 0)
    # short lengthscale: |x/l| large
    G = cg.gramian(cg.Lengthscale(cg.EQ(), 0.05), torch.from_numpy(X0).cuda()); b = (G @ ad).cpu().numpy()
    assert cg.get_info("last_dense_path") == 1
    assert relerr(b, oracle.mul(None, oracle.Kernel(oracle.EQ, lengthscale=0.05), X0, X0, a, dtype=np.float32)) <= 1e-5
    # fp64 and the profiles that are not differentiable in s at 0 always use the lane-per-row kernel; the smooth fp32
    # profiles, dot-product kernels and several right-hand sides take the generic matrix-core kernel (dense_mfma.hpp)
    (cg.gramian(cg.EQ(), torch.from_numpy(X0.astype(np.float64)).cuda()) @ ad.double()); assert cg.get_info("last_dense_path") == 1
    (cg.gramian(cg.Exp(), torch.from_numpy(X0).cuda()) @ ad); assert cg.get_info("last_dense_path") == 1
    (cg.gramian(cg.MaternP(0), torch.from_numpy(X0).cuda()) @ ad); assert cg.get_info("last_dense_path") == 1
    (cg.gramian(cg.GammaExp(1.5), torch.from_numpy(X0).cuda()) @ ad); assert cg.get_info("last_dense_path") == 1
    (cg.gramian(cg.InverseMultiQuadratic(0.01), torch.from_numpy(X0).cuda()) @ ad); assert cg.get_info("last_dense_path") == 1   # 1/c^2 sensitivity
    (cg.gramian(cg.RQ(1.0), torch.from_numpy(X0).cuda()) @ ad); assert cg.get_info("last_dense_path") == 2
    (cg.gramian(cg.EQ(), torch.from_numpy(X0).cuda()) @ torch.randn(n, 3, device="cuda")); assert cg.get_info("last_dense_path") == 2


def rowwise_err(b, ref, absref, L=None):
    """max_i |b_i - ref_i| / sum_j |k_ij a_j|: the row-wise bound — a norm-wise check cannot see a
    row whose entries are all tiny (a test point far from the training cloud).
    With L_i = -ln max_j k_ij given, row i's error is divided by max(1, L_i / 10): exp(-r^2/2) has relative condition number
    r^2/2 = L, so ANY fp32 evaluation of r^2 — the reference's direct differences (src/util.jl:40-47) included — carries a
    relative error ~ L * few * 2^-24 in such an entry (L = 35: ~1e-5); rows with a neighbour within r^2/2 <= 10 keep 1e-5."""
    b = np.asarray(b, dtype=np.float64)
    e = np.abs(b - ref) / absref
    if L is not None:
        e = e / np.maximum(1.0, L / 10.0)
    return float(np.max(e))


def eq_row_logs(X, Y):
    """L_i = min_j |x_i - y_j|^2 / 2 = -ln max_j k_ij for the EQ kernel (fp64)."""
    X = X.astype(np.float64); Y = Y.astype(np.float64)
    d2 = (X * X).sum(1)[:, None] + (Y * Y).sum(1)[None, :] - 2.0 * X @ Y.T
    return np.maximum(d2.min(1), 0.0) / 2.0


SQRT_LOG2E = 1.2011224087864498      # x~ = SQRT_LOG2E * x / l: the scaled units of the matrix-core gate


@pytest.mark.parametrize("d", [2, 3, 8])
def test_eq_matrix_core_band_far_rows_and_columns(cg, oracle, d):
    """round 1's gate bounded only the PRODUCT max|x~| max|y~| <= 126 while the kernel kept
    e_i = exp2(-|x~_i|^2/2) and a_j e_j as separate fp32 factors, so a test cluster ~16 scaled units from a compact training
    cluster passed the gate and lost every row to an underflowing e_i.  Now the integer parts of the half-norms ride through the
    MFMA (every exponential is <= 4), their fractions are factors in (1/2, 1] (dense_mfma.hpp: norm_split), and the gate bounds
    EACH radius.  Reference behaviour: direct differences never lose these rows
    (src/util.jl:40-47, src/stationary.jl:42).  Checked norm-wise (<= 1e-5), ROW-wise (|b_i - ref_i| <= 1e-5 max(1, L_i / 10) sum_j |k_ij a_j|,
    L_i = -ln max_j k_ij: see rowwise_err)
    and for finiteness, on the library's own choice (dense_variant 0) and with the matrix cores forced where the gate admits them."""
    rng = np.random.default_rng(4200 + d)
    ko = oracle.Kernel(oracle.EQ)
    k = cg.EQ()

    def unit(v):
        return v / np.linalg.norm(v, axis=-1, keepdims=True)

    def ball(n, radius):          # points in a d-ball of the given radius (scaled units), some ON the sphere
        u = unit(rng.standard_normal((n, d))) * radius * rng.random((n, 1)) ** (1.0 / d)
        u[:4] = unit(rng.standard_normal((4, d))) * radius
        return u

    e1 = np.zeros(d); e1[0] = 1.0
    f16_seen = False
    cases = []
    # (name, X~, Y~, expected path under dense_variant 0: 1 direct differences, 2 matrix cores)
    for off, ry in ((14.0, 7.0), (16.0, 7.0), (18.0, 5.0), (17.5, 6.9)):
        # X: a cluster of radius 1 at `off` scaled units from the Y ball's centre — round 1: product <= 126 passed, rows = 0
        cases.append((f"far X cluster {off}/{ry}", ball(300, 1.0) + off * e1, ball(700, ry), 1))
    # one far outlier row against a tight Y, inside the gate (|x~| = 10.5 -> 110 <= 126) and outside it (15)
    Xo = ball(257, 3.0); Xo[5] = 10.5 * unit(rng.standard_normal(d)); cases.append(("outlier row 10.5", Xo, ball(600, 3.0), 2))
    Xo = ball(257, 3.0); Xo[5] = 15.0 * unit(rng.standard_normal(d)); cases.append(("outlier row 15", Xo, ball(600, 3.0), 1))
    # the mirror: far columns (their weights a_j e_j used to underflow), tight rows
    Yo = ball(600, 3.0); Yo[7] = 10.5 * unit(rng.standard_normal(d)); Yo[300] = -10.5 * unit(rng.standard_normal(d))
    cases.append(("outlier columns 10.5", ball(257, 3.0), Yo, 2))
    # both sets wide and aligned: x~ . y~ reaches ~120 (round 1: exp2(120) * tiny weights)
    cases.append(("wide aligned", ball(400, 9.8), ball(500, 9.8), 2))
    # a compact X well inside a wide Y and vice versa
    cases.append(("compact in wide", ball(300, 2.0) + 5.0 * e1, ball(800, 9.8), 2))
    # wide and aligned again, near the edge of the fp16 split's gate (7.5^2 = 56; the bound on the radii about the sample centres ~67 <= 72); last and from its own stream, so that the cases above keep
    # the data they always had
    st = rng.bit_generator.state; rng = np.random.default_rng(4300 + d)
    cases.append(("wide aligned 7.5", ball(400, 7.5), ball(500, 7.5), 2))
    rng = np.random.default_rng(0); rng.bit_generator.state = st
    try:
        for name, Xs, Ys, path in cases:
            X = (Xs / SQRT_LOG2E + 3.0).astype(np.float32); Y = (Ys / SQRT_LOG2E + 3.0).astype(np.float32)   # a common offset changes nothing
            a = rng.standard_normal(len(Y)).astype(np.float32)
            ref = oracle.mul(None, ko, X, Y, a, dtype=np.float32)
            absref = oracle.mul(None, ko, X, Y, np.abs(a), dtype=np.float32)
            assert np.all(absref > 0)
            L = eq_row_logs(X, Y)
            G = cg.gramian(k, torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()); ad = torch.from_numpy(a).cuda()
            cg.set_option("dense_variant", 0)
            b = (G @ ad).cpu().numpy()
            assert cg.get_info("last_dense_path") == path, (name, d, cg.get_info("last_dense_path"))
            assert np.isfinite(b).all(), (name, d)
            assert relerr(b, ref) <= 1e-5, (name, d, relerr(b, ref))
            assert rowwise_err(b, ref, absref, L) <= 1e-5, (name, d, rowwise_err(b, ref, absref, L), L.max())
            print(f"band d={d} {name}: path {path} norm-wise {relerr(b, ref):.2e} row-wise {rowwise_err(b, ref, absref):.2e} scaled {rowwise_err(b, ref, absref, L):.2e} Lmax {L.max():.1f}")
            if path == 2:      # where the gate admits the matrix cores: every variant of that kernel
                for rpl, lds, f16 in ((1, 0, 0), (2, 0, 0), (2, 1, 0), (1, 1, 0), (4, 1, 0), (2, 0, 1), (2, 1, 1), (1, 1, 1), (4, 1, 1)):
                    cg.set_option("dense_variant", 2); cg.set_option("rows_per_lane", rpl); cg.set_option("mfma_lds", lds); cg.set_option("mfma_f16", f16)
                    b2 = (G @ ad).cpu().numpy()
                    assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_f16") in (0, f16)     # fp16 split: only inside ITS gate (72)
                    f16_seen = f16_seen or cg.get_info("last_mfma_f16") == 1
                    assert np.isfinite(b2).all() and relerr(b2, ref) <= 1e-5 and rowwise_err(b2, ref, absref, L) <= 1e-5, \
                        (name, d, rpl, lds, relerr(b2, ref), rowwise_err(b2, ref, absref, L))
                cg.set_option("rows_per_lane", 0); cg.set_option("mfma_lds", -1); cg.set_option("mfma_f16", -1)
            else:              # forced past the gate the kernel must still be finite and lose no row (it is merely less accurate)
                cg.set_option("dense_variant", 2)
                b2 = (G @ ad).cpu().numpy()
                assert cg.get_info("last_dense_path") == 2 and np.isfinite(b2).all()
                assert np.all(b2[np.abs(ref) > 1e-30 * np.abs(ref).max()] != 0), (name, d)
                print(f"   forced past the gate: row-wise {rowwise_err(b2, ref, absref):.2e}")
                assert rowwise_err(b2, ref, absref) <= 2e-4, (name, d, rowwise_err(b2, ref, absref))
            # the product is linear in a over the whole fp32 range (the weights stay within a factor 2 of a_j; round 1's a_j e_j
            # went denormal below |a| ~ 2^-62): scaling a by a power of two scales b (up to the flush of terms < 2^-126)
            cg.set_option("dense_variant", 0)
            for sc in (2.0 ** -60, 2.0 ** 60):
                bs = (G @ (ad * sc)).cpu().numpy().astype(np.float64) / sc
                ok = (absref * sc > 1e-30) & (absref * sc < 1e30)      # rows whose scaled result fp32 can hold at all
                assert ok.sum() >= len(ok) // 2 or "far X" in name
                assert np.isfinite(bs).all() and (not ok.any() or rowwise_err(bs[ok], b.astype(np.float64)[ok], absref[ok]) <= 1e-6), (name, d, sc)
    finally:
        cg.set_option("dense_variant", 0); cg.set_option("rows_per_lane", 0); cg.set_option("mfma_lds", -1); cg.set_option("mfma_f16", -1)
    assert f16_seen


@pytest.mark.parametrize("d", [3, 8, 12])
def test_symmetric_matrix_core_kernels_far_outliers_rowwise(cg, oracle, d):
    """gramian(k, x) with a few points far from the cloud (|x~ - c| up to 10.5 scaled units, inside the gate): their rows AND
    their columns must survive on the symmetric and on the general matrix-core kernel — norm-wise, row-wise, finite; EQ and RQ."""
    rng = np.random.default_rng(4300 + d)
    n = 1500
    Xs = rng.standard_normal((n, d)); Xs *= 2.5 / np.sqrt(d)
    for i, r in ((3, 10.5), (700, 10.0), (1499, 9.0)):
        v = rng.standard_normal(d); Xs[i] = r * v / np.linalg.norm(v)
    Xs[701] = Xs[700] * 1.01                                   # a neighbour for one of the outliers
    X = (Xs / SQRT_LOG2E - 7.0).astype(np.float32)
    a = rng.standard_normal(n).astype(np.float32)
    Xd = torch.from_numpy(X).cuda(); ad = torch.from_numpy(a).cuda()
    try:
        for k, ko in ((cg.EQ(), oracle.Kernel(oracle.EQ)), (cg.RQ(2.0), oracle.Kernel(oracle.RQ, param=2.0))):
            ref = oracle.mul(None, ko, X, X, a, dtype=np.float32)
            absref = oracle.mul(None, ko, X, X, np.abs(a), dtype=np.float32)
            G = cg.gramian(k, Xd)
            for sym in (0, 1):
                cg.set_option("mfma_sym", sym)
                b = (G @ ad).cpu().numpy()
                assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_sym") == sym
                assert np.isfinite(b).all()
                assert relerr(b, ref) <= 1e-5 and rowwise_err(b, ref, absref) <= 1e-5, (type(k).__name__, d, sym, relerr(b, ref), rowwise_err(b, ref, absref))
    finally:
        cg.set_option("mfma_sym", -1)


@pytest.mark.parametrize("d", [1, 3, 4, 8, 16, 31])
def test_generic_matrix_core_path(cg, oracle, d):
    """fp32: RQ, Cauchy, IMQ, MaternP(1..3), Dot^p, ExponentialDot and multi-RHS EQ on the matrix cores against the fp64 oracle and
    against the direct-difference kernel (dense_variant = 1), ragged shapes, 1 / 3 / 9 right-hand sides, alpha / beta."""
    rng = np.random.default_rng(900 + d)
    wanted = {"EQ", "RQ(1.0)", "RQ(0.37)", "Cauchy", "IMQ(0.8)", "MaternP(1)", "MaternP(2)", "MaternP(3)", "2.5*Lengthscale(MaternP(2),1.3)",
              "Dot()^3", "Dot()", "ExponentialDot", "EQ^2", "Lengthscale(EQ,0.7)"}
    cases = [c for c in kernel_cases.cases(cg) if c[0] in wanted]
    try:
        for (n, m) in ((33, 31), (300, 1000)):
            X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32); Y = (rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)
            Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
            for p in (1, 3, 9):
                A = rng.standard_normal((m, p)).astype(np.float32); Y0 = rng.standard_normal((n, p)).astype(np.float32)
                if p == 1: A, Y0 = A[:, 0], Y0[:, 0]
                for name, k, ko in cases:
                    if p == 1 and name in ("EQ", "Lengthscale(EQ,0.7)"):
                        continue                                   # single-RHS EQ has its own kernel and test
                    G = cg.gramian(k, Xd, Yd)
                    ref = oracle.mul(Y0, ko, X, Y, A, -0.7, 1.3, np.float32)
                    got = {}
                    for variant in (1, 2):
                        cg.set_option("dense_variant", variant)
                        yd = torch.from_numpy(Y0.copy()).cuda()
                        cg.mul_(yd, G, torch.from_numpy(A).cuda(), -0.7, 1.3)
                        assert cg.get_info("last_dense_path") == variant, (name, variant)
                        got[variant] = yd.cpu().numpy()
                        assert relerr(got[variant], ref) <= 1e-5, (name, d, n, p, variant, relerr(got[variant], ref))
                    assert relerr(got[2], got[1]) <= 5e-6, (name, d, n, p)
                    # four waves sharing every column tile through LDS (what long column chunks get): bit-identical
                    for js in (0, 3):
                        cg.set_option("mfma_lds", 1); cg.set_option("jsplit", js)
                        yl = torch.from_numpy(Y0.copy()).cuda()
                        cg.mul_(yl, G, torch.from_numpy(A).cuda(), -0.7, 1.3)
                        assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_lds") == 1
                        assert relerr(yl.cpu().numpy(), ref) <= 1e-5, (name, d, n, p, js)
                        if js == 0:
                            assert np.array_equal(yl.cpu().numpy(), got[2]), (name, d, n, p)
                    cg.set_option("mfma_lds", -1); cg.set_option("jsplit", 0)
    finally:
        cg.set_option("dense_variant", 0); cg.set_option("mfma_lds", -1); cg.set_option("jsplit", 0)


# ---- (f)-3 / (f)-4: factorizations and Toeplitz solves on top of the hot path -------------------------------------------
def test_cholesky_factorize_and_lazy_pivoted_cholesky(cg, oracle):
    """cholesky / factorize (src/gramian.jl:192-213): dense tile + rocSOLVER, and the lazy pivoted variant that only ever
    evaluates the diagonal and one Gramian column per step, against the oracle's dense pivoted Cholesky."""
    rng = np.random.default_rng(21)
    X = rng.standard_normal((300, 2))
    Xd = torch.from_numpy(X).cuda()
    G = cg.gramian(cg.EQ(), Xd)
    M = oracle.matrix(oracle.Kernel(oracle.EQ), X)
    P = cg.cholesky(G, pivoted=True, tol=1e-6)
    Lo, pivo, ranko = oracle.pivoted_cholesky(M, tol=1e-6)
    assert P.rank == ranko and P.piv.cpu().numpy()[:ranko].tolist() == pivo[:ranko].tolist()
    assert np.abs(P.L.cpu().numpy() - Lo).max() <= 1e-8
    assert np.abs(P.to_dense().cpu().numpy() - M).max() <= 300 * 1e-6
    F = cg.factorize(G)                                          # n <= 2^14: pivoted Cholesky with tol = 1e-6
    assert isinstance(F, cg.PivotedCholesky) and F.rank == ranko
    assert cg.factorize(G, max_cholesky_size=100) is G           # too large to instantiate: stays lazy
    # exact low rank is detected: Dot kernel in d = 3
    Gd = cg.gramian(cg.Dot(), torch.from_numpy(rng.standard_normal((200, 3))).cuda())
    assert cg.cholesky(Gd, pivoted=True, tol=1e-10).rank == 3
    assert np.allclose(cg.diagonal(Gd).cpu().numpy(), np.diag(oracle.matrix(oracle.Kernel(oracle.DOT), Gd.x.cpu().numpy())), rtol=1e-13)
    # plain Cholesky of a well-conditioned Gramian, and a solve through it
    Xs = rng.standard_normal((200, 3))
    Ge = cg.gramian(cg.Exp(), torch.from_numpy(Xs).cuda())
    C = cg.cholesky(Ge)
    Me = oracle.matrix(oracle.Kernel(oracle.EXP), Xs)
    assert np.abs(C.L.cpu().numpy() - np.linalg.cholesky(Me)).max() <= 1e-10
    b = rng.standard_normal(200)
    assert relerr(C.solve(torch.from_numpy(b).cuda()).cpu().numpy(), np.linalg.solve(Me, b)) <= 1e-9
    with pytest.raises(ValueError):                              # PosDefException with check = true (rank-deficient Dot Gramian)
        cg.cholesky(Gd)


@pytest.mark.parametrize("n", [257, 4096])
def test_toeplitz_solve_pcg_vs_levinson(cg, oracle, n):
    """T \\ b for SPD Toeplitz Gramians: PCG over the FFT MVM with a circulant preconditioner against the oracle's
    restatement of the reference's Levinson recursion (src/toeplitz.jl:77-111), diagonal != 1 included."""
    rng = np.random.default_rng(n)
    b = rng.standard_normal(n)
    bd = torch.from_numpy(b).cuda()
    for k, ko in ((cg.Exp(), oracle.Kernel(oracle.EXP)), (2.5 * cg.Lengthscale(cg.Exp(), 0.3), oracle.Kernel(oracle.EXP, lengthscale=0.3, scale=2.5))):
        T = cg.gramian(k, cg.srange(-1, 1, n))
        assert isinstance(T, cg.SymmetricToeplitz)
        x, info = cg.toeplitz_solve(T, bd, reltol=1e-13)
        vc, _ = oracle.toeplitz_vectors(ko, oracle.srange(-1, 1, n))
        ref = oracle.levinson_toeplitz(vc, b)
        assert info["converged"] and info["iterations"] <= 25, info     # the circulant preconditioner clusters the spectrum
        assert relerr(x.cpu().numpy(), ref) <= 1e-8, (n, relerr(x.cpu().numpy(), ref), info)
        r = (T @ x).cpu().numpy() - b                                  # residual through the hot path
        assert np.linalg.norm(r) <= 1e-10 * np.linalg.norm(b)
    # a smoother kernel plus a nugget (the GP-regression system K + sigma^2 I): built from its first column directly
    ko = oracle.Kernel(oracle.MATERNP, p=1, lengthscale=0.3)
    vc, _ = oracle.toeplitz_vectors(ko, oracle.srange(-1, 1, n))
    vc = vc.copy(); vc[0] += 0.05
    T = cg.SymmetricToeplitz(torch.from_numpy(vc).cuda())
    x, info = cg.toeplitz_solve(T, bd, reltol=1e-12)
    assert info["converged"] and relerr(x.cpu().numpy(), oracle.levinson_toeplitz(vc, b)) <= 1e-8, info


@pytest.mark.parametrize("n", [1, 2, 3, 32, 257, 2049])
def test_toeplitz_direct_solvers_durbin_levinson_trench(cg, oracle, n):
    """src/toeplitz.jl:12-111 on the device (csrc/toeplitz_direct.hip) against the oracle's restatement and against dense
    algebra, as test/toeplitz.jl:7-42 does: durbin(r) = inv(K) (-r), levinson(r, b) = inv(K) b, trench(r) = inv(K); EQ / Exponential
    Gramians on a grid (positive definite), unit and non-unit diagonals, fp64 and fp32."""
    rng = np.random.default_rng(800 + n)
    xs = np.linspace(-1.0, 1.0, n + 1)
    for fam, tolf in ((lambda t: np.exp(-np.abs(t)), 1.0), (lambda t: np.exp(-0.5 * (3.0 * t) ** 2) + 1e-3 * (t == 0), 100.0)):
        vc = fam(xs - xs[0])                                         # first column of the (n+1) x (n+1) Toeplitz Gramian, vc[0] = 1 (+ jitter)
        r = vc[1:] / vc[0]
        K = oracle.toeplitz_dense(vc / vc[0]) if hasattr(oracle, "toeplitz_dense") else np.array([[vc[abs(i - j)] / vc[0] for j in range(n + 1)] for i in range(n + 1)])
        b = rng.standard_normal(n + 1)
        rd = torch.from_numpy(r).cuda()
        # Durbin: K_n \ (-r) with K_n the leading n x n block
        y = cg.durbin(rd).cpu().numpy()
        assert relerr(y, oracle.durbin(r)) <= 1e-9 * tolf, ("durbin", n, relerr(y, oracle.durbin(r)))
        assert relerr(K[:n, :n] @ y, -r) <= 1e-9 * tolf
        # Levinson, vector form and operator form (non-unit diagonal: T = 2.5 K)
        x = cg.levinson(rd, torch.from_numpy(b).cuda()).cpu().numpy()
        assert relerr(x, oracle.levinson(r, b)) <= 1e-9 * tolf, ("levinson", n, relerr(x, oracle.levinson(r, b)))
        assert relerr(K @ x, b) <= 1e-8 * tolf
        T = cg.SymmetricToeplitz(torch.from_numpy(2.5 * vc).cuda())
        xT = cg.levinson(T, torch.from_numpy(b).cuda()).cpu().numpy()
        assert relerr(xT, oracle.levinson_toeplitz(2.5 * vc, b)) <= 1e-9 * tolf
        # Trench: the full symmetric inverse
        if n <= 257:
            Bi = cg.trench(rd).cpu().numpy()
            assert relerr(Bi, oracle.trench(r)) <= 1e-9 * tolf, ("trench", n, relerr(Bi, oracle.trench(r)))
            assert np.array_equal(Bi, Bi.T)
            assert relerr(Bi @ K, np.eye(n + 1)) <= 1e-7 * tolf
            BT = cg.trench(T).cpu().numpy()
            assert relerr(BT @ (2.5 * vc[0] * K), np.eye(n + 1)) <= 1e-7 * tolf
    # fp32: the same chains in single precision (Exponential: well conditioned)
    vc32 = np.exp(-np.abs(xs - xs[0])).astype(np.float32)
    x32 = cg.levinson(torch.from_numpy(vc32[1:]).cuda(), torch.from_numpy(b.astype(np.float32)).cuda()).cpu().numpy()
    # (the grid Gramian's condition number grows like n^2: single precision keeps ~ 7 - 2 log10(n) digits)
    assert relerr(x32, oracle.levinson(vc32[1:].astype(np.float64), b.astype(np.float32))) <= max(1e-5, 1e-6 * (n + 1) ** 2)
    with pytest.raises(cg._ffi.DimensionMismatch):
        cg.levinson(rd, torch.zeros(n + 3, dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("m", [63, 64, 65, 511, 512, 513, 1025, 4097, 16383, 16384, 16385])
def test_toeplitz_direct_solvers_on_chip_boundaries(cg, oracle, m):
    """Durbin / Levinson at the sizes where the on-chip kernel (csrc/toeplitz_direct.hip: levinson_reg_kernel, m <= 16384, 16 / 32
    entries per thread, 64-lane waves) changes regime — one entry either side of a wave, of the whole workgroup, and of the size
    cap where the global-memory kernel takes over — against the oracle's restatement of src/toeplitz.jl:14-27 / :77-98 and
    through the residual of the FFT MVM.  Exponential kernel on a grid: condition number ~ m^2, so the fp64 chains keep
    ~ 16 - 2 log10(m) digits."""
    rng = np.random.default_rng(4100 + m)
    xs = np.linspace(-1.0, 1.0, m)
    vc = np.exp(-np.abs(xs - xs[0]))
    r, b = vc[1:].copy(), rng.standard_normal(m)
    tol = max(1e-10, 1e-15 * m * m)
    rd, bd = torch.from_numpy(r).cuda(), torch.from_numpy(b).cuda()
    x = cg.levinson(rd, bd)
    if m <= 4097:                                                   # (the oracle's Levinson is a Python double loop: a minute at m = 16384)
        assert relerr(x.cpu().numpy(), oracle.levinson(r, b)) <= tol, ("levinson", m)
    T = cg.SymmetricToeplitz(torch.from_numpy(vc).cuda())
    assert relerr((T @ x).cpu().numpy(), b) <= tol
    y = cg.durbin(rd).cpu().numpy()                                 # length m - 1: the next size down of the same kernel
    assert relerr(y, oracle.durbin(r)) <= tol, ("durbin", m)
    # fp32: the same chains on a diagonal of 1.5 (well conditioned at every size: the unit-diagonal grid system loses ~ 4e-10 m^2
    # to the rounding of its INPUTS alone, tools/levinson_fp32_accuracy.py) against the fp64 chain; measured 1.4e-7 sqrt(m)
    # (Levinson) and 6e-7 sqrt(m) (Durbin)
    rj = rd / 1.5
    xj32, xj64 = cg.levinson(rj.float(), bd.float()), cg.levinson(rj, bd)
    assert relerr(xj32.double().cpu().numpy(), xj64.cpu().numpy()) <= 1e-6 * np.sqrt(m), ("levinson fp32", m)
    yj32, yj64 = cg.durbin(rj.float()), cg.durbin(rj)
    assert relerr(yj32.double().cpu().numpy(), yj64.cpu().numpy()) <= 3e-6 * np.sqrt(m), ("durbin fp32", m)
    if m <= 4097:
        assert relerr(yj64.cpu().numpy(), oracle.durbin(r / 1.5)) <= 1e-11


@pytest.mark.parametrize("dtype,n", [(torch.float64, 65536), (torch.float64, 250000), (torch.float64, 1000000), (torch.float64, 3000001),
                                     (torch.float32, 250000), (torch.float32, 4000000),
                                     (torch.float64, 500000), (torch.float32, 100000),      # N = 2^20, 2^18: column length 512 (x 1024, x 256)
                                     (torch.float64, 2000000), (torch.float32, 6000000)])   # N = 2^22: 512 x 4096; N = 2^24: 2048 x 4096
def test_toeplitz_fused_row_fft_kernel(cg, oracle, dtype, n):
    """Row length M' in {64, 256, 1024, 4096} with the column length (512, 1024 or 2048) that gives it: the row FFT, spectral step and inverse row FFT run as ONE kernel
    (rowfft_fused_kernel; M' = 4096: rowfft16_fused_kernel, radix-16 stages with global I/O in the outer stages) — against the numpy circulant-embedding oracle, explicit dense rows, and the rocFFT-batch path
    (option toeplitz_fused = 0), symmetric and non-symmetric, alpha / beta."""
    tol = 1e-5 if dtype == torch.float32 else 1e-10
    rng = np.random.default_rng(n)
    x = cg.srange(-1, 1, n, dtype)
    a = rng.standard_normal(n).astype(npdt(dtype)); y0 = rng.standard_normal(n).astype(npdt(dtype))
    ad = torch.from_numpy(a).cuda()
    ko = oracle.Kernel(oracle.EXP)
    G = cg.gramian(cg.Exp(), x)
    vc, _ = oracle.toeplitz_vectors(ko, oracle.srange(-1, 1, n))
    ref = oracle.toeplitz_mul(None, vc, None, a.astype(np.float64))
    try:
        out = {}
        for fused in (1, 0, 2):      # 1: the default (radix-16 stages at M' = 4096), 0: rocFFT batches, 2: radix-4 stages everywhere
            cg.set_option("toeplitz_fused", fused)
            yd = torch.from_numpy(y0.copy()).cuda()
            cg.mul_(yd, G, ad, 0.3, -1.1)
            out[fused] = yd.cpu().numpy()
            assert relerr(out[fused], 0.3 * ref - 1.1 * y0) <= tol, (fused, n, relerr(out[fused], 0.3 * ref - 1.1 * y0))
        assert relerr(out[1], out[0]) <= tol and relerr(out[2], out[1]) <= tol
        cg.set_option("toeplitz_fused", 1)
        # the column FFT: radix-16 register butterflies (colfft16_kernel, the default, used above) against the radix-4 LDS kernel
        cg.set_option("toeplitz_colfft", 4)
        yd = torch.from_numpy(y0.copy()).cuda()
        cg.mul_(yd, G, ad, 0.3, -1.1)
        cg.set_option("toeplitz_colfft", 16)
        assert relerr(yd.cpu().numpy(), 0.3 * ref - 1.1 * y0) <= tol and relerr(yd.cpu().numpy(), out[1]) <= tol
        # the symmetric matrix keeps the row kernel's spectrum copy as reals; a handle created with the complex copy agrees
        cg.set_option("toeplitz_real_spectrum", 0)
        Gc = cg.gramian(cg.Exp(), x)
        cg.set_option("toeplitz_real_spectrum", 1)
        yd = torch.from_numpy(y0.copy()).cuda()
        cg.mul_(yd, Gc, ad, 0.3, -1.1)
        assert relerr(yd.cpu().numpy(), 0.3 * ref - 1.1 * y0) <= tol and relerr(yd.cpu().numpy(), out[1]) <= tol
        b = (G @ ad).cpu().numpy()
        rows = rng.choice(n, 8, replace=False)
        dense_rows = np.array([np.dot(vc[np.abs(i - np.arange(n))], a.astype(np.float64)) for i in rows])
        assert relerr(b[rows], dense_rows) <= tol
        Tn = cg.gramian(cg.Exp(), x, x + 0.123)
        rg = oracle.srange(-1, 1, n)
        vc2, vr2 = oracle.toeplitz_vectors(ko, rg, (rg[0] + 0.123, rg[1], rg[2]))
        assert relerr((Tn @ ad).cpu().numpy(), oracle.toeplitz_mul(None, vc2, vr2, a.astype(np.float64))) <= tol
    finally:
        cg.set_option("toeplitz_fused", 1)
        cg.set_option("toeplitz_colfft", 16)
        cg.set_option("toeplitz_real_spectrum", 1)


def test_matern_real_nu_golden(cg, oracle):
    """Matern(nu) with the device Bessel function (Temme series / Steed continued fraction) against the mpmath-checked fixtures:
    dense, gradient and value-gradient MVMs; Matern(p + 1/2) equals MaternP(p); Matrix(G) diagonal is exactly 1."""
    g = np.load(f"{GOLD}/composite.npz")
    for d in (1, 3, 8):
        for (n, m) in ((4, 6), (65, 33)):
            tag = f"d{d}_n{n}_m{m}"
            X, Y, a, y0, ag, yg0, av, yv0 = (g[f"{tag}_{s}"] for s in ("X", "Y", "a", "y0", "ag", "yg0", "av", "yv0"))
            alpha, beta = g[f"{tag}_ab"]
            Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
            for name in g["matern_names"]:
                fam, p, power, param, ls, sc = g[f"matern_{name}_fields"]
                k = sc * cg.Lengthscale(cg.Matern(param), ls) if ls != 1.0 else sc * cg.Matern(param)
                for K, vec, y_init, key in ((cg.gramian(k, Xd, Yd), a, y0, "b"), (cg.gramian(cg.GradientKernel(k), Xd, Yd), ag, yg0, "bg"),
                                            (cg.gramian(cg.ValueGradientKernel(k), Xd, Yd), av, yv0, "bv")):
                    yd = torch.from_numpy(y_init.copy()).cuda()
                    cg.mul_(yd, K, torch.from_numpy(vec).cuda(), alpha, beta)
                    assert relerr(yd.cpu().numpy(), g[f"{tag}_{name}_{key}"]) <= 1e-12, (name, tag, key, relerr(yd.cpu().numpy(), g[f"{tag}_{name}_{key}"]))
    rng = np.random.default_rng(8)
    X = torch.from_numpy(rng.standard_normal((200, 3))).cuda(); a = torch.from_numpy(rng.standard_normal(200)).cuda()
    for p in (0, 1, 2, 3):
        assert relerr((cg.gramian(cg.Matern(p + 0.5), X) @ a).cpu().numpy(), (cg.gramian(cg.MaternP(p), X) @ a).cpu().numpy()) <= 1e-12
    M = cg.gramian(cg.Matern(0.8), X).to_dense()
    assert torch.all(torch.diagonal(M) == 1.0) and bool(torch.isfinite(M).all())
    with pytest.raises(cg.DomainError):
        cg.Matern(-1.0)


def test_transformed_kernels(cg, oracle):
    """ARD / ScaledInputKernel / Warped / Periodic / VerticalRescaling / Cosine / Polynomial (src/transformation.jl,
    src/stationary.jl:197-211, src/mercer.jl:12-14): the points (or the output) are transformed once on the host side and the
    hot path does the rest — against the definitions restated in the oracle; test/stationary.jl:132-175, test/transformation.jl:12-45."""
    rng = np.random.default_rng(17)
    n, m, d = 70, 45, 3
    X = rng.standard_normal((n, d)); Y = rng.standard_normal((m, d))
    a = rng.standard_normal(m); ag = rng.standard_normal(m * d)
    Xd, Yd, ad, agd = (torch.from_numpy(v).cuda() for v in (X, Y, a, ag))
    ko = oracle.Kernel(oracle.MATERNP, p=2)
    # ARD == scaled inputs; ScaledInputKernel with a rectangular U; Warped with a callable
    l = np.array([0.5, 1.3, 2.0])
    G = cg.gramian(cg.ARD(cg.MaternP(2), l), Xd, Yd)
    assert relerr((G @ ad).cpu().numpy(), oracle.matrix(ko, oracle.scaled_input_points(1 / l, X), oracle.scaled_input_points(1 / l, Y)) @ a) <= 1e-12
    assert isinstance(cg.ARD(cg.EQ(), 2.0), cg.Lengthscale)                                       # scalar l: plain Lengthscale
    U = rng.standard_normal((5, d))
    G = cg.gramian(cg.ScaledInputKernel(cg.RQ(1.0), U), Xd, Yd)
    assert relerr((G @ ad).cpu().numpy(), oracle.matrix(oracle.Kernel(oracle.RQ, param=1.0), oracle.scaled_input_points(U, X), oracle.scaled_input_points(U, Y)) @ a) <= 1e-12
    G = cg.gramian(cg.Warped(cg.EQ(), lambda p: torch.tanh(p)), Xd, Yd)
    assert relerr((G @ ad).cpu().numpy(), oracle.matrix(oracle.Kernel(oracle.EQ), np.tanh(X), np.tanh(Y)) @ a) <= 1e-12
    # gradient Gramians through the linear maps (chain rule U' B U)
    K = cg.gramian(cg.GradientKernel(cg.ARD(cg.MaternP(2), l)), Xd, Yd)
    assert relerr((K @ agd).cpu().numpy(), oracle.linear_map_grad_matrix(ko, 1 / l, X, Y) @ ag) <= 1e-12
    K = cg.gramian(cg.GradientKernel(cg.ScaledInputKernel(cg.EQ(), U)), Xd, Yd)
    assert relerr((K @ agd).cpu().numpy(), oracle.linear_map_grad_matrix(oracle.Kernel(oracle.EQ), U, X, Y) @ ag) <= 1e-12
    # Periodic on 1-D inputs, incl. a regular grid (stays a plain Gramian on the embedded points)
    x1 = rng.standard_normal(60); y1 = rng.standard_normal(33); a1 = rng.standard_normal(33)
    P = cg.gramian(cg.Periodic(cg.EQ()), torch.from_numpy(x1).cuda(), torch.from_numpy(y1).cuda())
    assert relerr((P @ torch.from_numpy(a1).cuda()).cpu().numpy(), oracle.periodic_matrix(oracle.Kernel(oracle.EQ), x1, y1) @ a1) <= 1e-12
    assert abs(cg.Periodic(cg.EQ())(0.3, 1.3) - 1.0) < 1e-12                                       # 1-periodic
    # VerticalRescaling = Diagonal * G * Diagonal
    f = lambda p: 1.0 / (1.0 + (p * p).sum(dim=1) if torch.is_tensor(p) else 1.0 / (1.0 + (np.asarray(p) ** 2).sum(axis=1)))
    fnp = lambda p: 1.0 / (1.0 + (p ** 2).sum(axis=1))
    V = cg.gramian(cg.VerticalRescaling(cg.MaternP(2), lambda p: 1.0 / (1.0 + (p * p).sum(1))), Xd, Yd)
    assert isinstance(V, cg.ScaledOperator)
    ref = fnp(X)[:, None] * oracle.matrix(ko, X, Y) * fnp(Y)[None, :]
    assert relerr((V @ ad).cpu().numpy(), ref @ a) <= 1e-12 and relerr(V.to_dense().cpu().numpy(), ref) <= 1e-12
    # Cosine: rank-2 Gramian, and its gradient Gramian (rank-2) x (c c')
    c = rng.standard_normal(d) * 0.3
    C = cg.gramian(cg.Cosine(c), Xd, Yd)
    assert isinstance(C, cg.LazyMatrixProduct) and C.U.shape[1] == 2
    assert relerr((C @ ad).cpu().numpy(), oracle.cosine_matrix(c, X, Y) @ a) <= 1e-12
    KC = cg.gramian(cg.GradientKernel(cg.Cosine(c)), Xd, Yd)
    assert relerr((KC @ agd).cpu().numpy(), oracle.cosine_grad_matrix(c, X, Y) @ ag) <= 1e-12
    assert isinstance(cg.input_trait(cg.Cosine(c)), cg.StationaryLinearFunctionalInput)
    # Polynomial(d, sigma) = (Dot() + sigma)^d as one composite
    Pk = cg.gramian(cg.Polynomial(4, 0.5), Xd, Yd)
    assert relerr((Pk @ ad).cpu().numpy(), ((X @ Y.T + 0.5) ** 4) @ a) <= 1e-12


def test_sharded_gramian_single_rank_dense_and_gradient(cg, oracle):
    """ShardedGramian without a process group (world = 1): same numbers as the local operators; block Gramians included."""
    rng = np.random.default_rng(41)
    X = rng.standard_normal((90, 4)); a = rng.standard_normal(90); ag = rng.standard_normal(90 * 4); av = rng.standard_normal(90 * 5)
    Xd = torch.from_numpy(X).cuda()
    S = cg.ShardedGramian(cg.MaternP(2), Xd)
    assert relerr((S @ torch.from_numpy(a).cuda()).cpu().numpy(), oracle.mul(None, oracle.Kernel(oracle.MATERNP, p=2), X, X, a)) <= 1e-12
    Sg = cg.ShardedGramian(cg.GradientKernel(cg.EQ()), Xd)
    assert Sg.block == 4 and Sg.shape == (360, 360)
    assert relerr((Sg @ torch.from_numpy(ag).cuda()).cpu().numpy(), oracle.grad_mul(None, oracle.Kernel(oracle.EQ), X, X, ag)) <= 1e-12
    Sv = cg.ShardedGramian(cg.ValueGradientKernel(cg.EQ()), Xd)
    assert Sv.block == 5
    assert relerr((Sv @ torch.from_numpy(av).cuda()).cpu().numpy(), oracle.valgrad_mul(None, oracle.Kernel(oracle.EQ), X, X, av)) <= 1e-12


def test_cg_on_sharded_gramian(cg, oracle):
    """(G + σ²I) \\ b by CG with the row-sharded operator as the MVM (world = 1 here; the only collective of an iteration is
    the all-gather inside the MVM)."""
    rng = np.random.default_rng(43)
    X = rng.standard_normal((300, 2)); b = rng.standard_normal(300)
    S = cg.ShardedGramian(cg.MaternP(1), torch.from_numpy(X).cuda())
    A = cg.LazyMatrixSum(S, 0.1 * torch.ones(300, dtype=torch.float64, device="cuda"))
    x, info = cg.cg(A, torch.from_numpy(b).cuda(), reltol=1e-10)
    M = oracle.matrix(oracle.Kernel(oracle.MATERNP, p=1), X) + 0.1 * np.eye(300)
    assert info["converged"] and relerr(x.cpu().numpy(), np.linalg.solve(M, b)) <= 1e-8


def test_neural_network_kernel(cg, oracle):
    """NN(σ) (src/mercer.jl:73-85): AsinDot on normalised augmented points; gradient Gramian by the chain rule through the
    normalisation == the reference's Woodbury block for σ = 0 (src/gradient.jl:187-210); test/gradient.jl:21 lists NN()."""
    rng = np.random.default_rng(23)
    n, m, d = 60, 41, 4
    X = rng.standard_normal((n, d)); Y = rng.standard_normal((m, d)); a = rng.standard_normal(m); ag = rng.standard_normal(m * d)
    Xd, Yd, ad, agd = (torch.from_numpy(v).cuda() for v in (X, Y, a, ag))
    for sigma in (0.0, 0.7):
        G = cg.gramian(cg.NN(sigma), Xd, Yd)
        M = oracle.nn_matrix(sigma, X, Y)
        assert relerr((G @ ad).cpu().numpy(), M @ a) <= 1e-12 and relerr(G.to_dense().cpu().numpy(), M) <= 1e-12
        assert abs(cg.NN(sigma)(X[0], Y[0]) - M[0, 0]) <= 1e-14
    Gs = cg.gramian(cg.NN(), Xd)                                                      # symmetric, diagonal < 1
    assert relerr((Gs @ torch.from_numpy(rng.standard_normal(n)).cuda()).shape[0], n) == 0
    K = cg.gramian(cg.GradientKernel(cg.NN()), Xd, Yd)
    assert relerr((K @ agd).cpu().numpy(), oracle.nn_grad_matrix(X, Y) @ ag) <= 1e-12
    Ks = cg.gramian(cg.GradientKernel(cg.NN()), Xd)
    Ms = oracle.nn_grad_matrix(X)
    assert np.abs(Ms - Ms.T).max() < 1e-13 and relerr(Ks.to_dense().cpu().numpy(), Ms) <= 1e-12
    Xf = X.astype(np.float32)
    Gf = cg.gramian(cg.NN(), torch.from_numpy(Xf).cuda())
    assert relerr((Gf @ torch.from_numpy(a[:1].repeat(n).astype(np.float32)).cuda()).cpu().numpy(), oracle.nn_matrix(0.0, Xf) @ a[:1].repeat(n)) <= 1e-5


def test_gramian_follows_in_place_point_updates(cg, oracle):
    """A lazy Gramian holds its points by reference (src/gramian.jl:10-21): after an in-place update of x the next mul! must
    see the new points — also on the matrix-core path, whose norm gate and packed fragments are cached per handle."""
    rng = np.random.default_rng(3)
    X = rng.standard_normal((500, 3)).astype(np.float32); a = rng.standard_normal(500).astype(np.float32)
    Xd = torch.from_numpy(X.copy()).cuda(); ad = torch.from_numpy(a).cuda()
    G = cg.gramian(cg.EQ(), Xd)
    b0 = (G @ ad).cpu().numpy()
    assert cg.get_info("last_dense_path") == 2 and relerr(b0, oracle.mul(None, oracle.Kernel(oracle.EQ), X, X, a, dtype=np.float32)) <= 1e-5
    Xd.mul_(0.5).add_(0.25)                                                # in place
    X2 = (X * 0.5 + 0.25).astype(np.float32)
    assert relerr((G @ ad).cpu().numpy(), oracle.mul(None, oracle.Kernel(oracle.EQ), X2, X2, a, dtype=np.float32)) <= 1e-5
    Xd.add_(200.0)                                                         # far from the origin: kernels and gate work relative
    X3 = (X2 + 200.0).astype(np.float32)                                   # to the set's own centre, so nothing changes
    assert relerr((G @ ad).cpu().numpy(), oracle.mul(None, oracle.Kernel(oracle.EQ), X3, X3, a, dtype=np.float32)) <= 1e-5
    assert cg.get_info("last_dense_path") == 2


@pytest.mark.parametrize("shift", [0.0, 1.0e3, -3.0e4])
def test_isotropic_paths_are_translation_invariant(cg, oracle, shift):
    """r = |x - y| does not depend on where the cloud sits (src/util.jl:40-47 subtracts first): every isotropic device path
    centres both sides on a common point before its pre-scale, so fp32 parity holds far from the origin, on the direct,
    matrix-core, wide, gradient and value-gradient kernels alike."""
    rng = np.random.default_rng(11)
    n, m = 300, 260
    for d in (2, 5, 70):
        X = (rng.standard_normal((n, d)) + shift).astype(np.float32)
        Y = (rng.standard_normal((m, d)) + shift).astype(np.float32)
        a = rng.standard_normal(m).astype(np.float32)
        Xd, Yd, ad = (torch.from_numpy(t).cuda() for t in (X, Y, a))
        l = 1.0 if d <= 5 else 6.0
        for kern, ok in ((cg.Lengthscale(cg.EQ(), l), oracle.Kernel(oracle.EQ, lengthscale=l)),
                         (cg.Lengthscale(cg.MaternP(2), 1.7 * l), oracle.Kernel(oracle.MATERNP, p=2, lengthscale=1.7 * l)),
                         (cg.Lengthscale(cg.RQ(1.5), l), oracle.Kernel(oracle.RQ, param=1.5, lengthscale=l))):
            want = oracle.mul(None, ok, X, Y, a, dtype=np.float32)
            for variant in (0, 1):
                cg.set_option("dense_variant", variant)
                try:
                    got = (cg.gramian(kern, Xd, Yd) @ ad).cpu().numpy()
                finally:
                    cg.set_option("dense_variant", 0)
                assert relerr(got, want) <= 1e-5, (shift, d, variant, type(kern).__name__, relerr(got, want))
    # gradient blocks (lane-per-row and panel kernels) and value-gradient blocks
    for d in (3, 70):
        X = (rng.standard_normal((120, d)) + shift).astype(np.float32)
        A = rng.standard_normal((120, d)).astype(np.float32)
        l = 1.0 if d == 3 else 6.0
        kern, ok = cg.Lengthscale(cg.EQ(), l), oracle.Kernel(oracle.EQ, lengthscale=l)
        Xd, Ad = torch.from_numpy(X).cuda(), torch.from_numpy(A).cuda()
        got = (cg.gramian(cg.GradientKernel(kern), Xd) @ Ad.reshape(-1)).cpu().numpy()
        want = oracle.grad_mul(None, ok, X, X, A.reshape(-1), dtype=np.float32)
        assert relerr(got, want) <= 1e-5, (shift, d, relerr(got, want))
        Av = rng.standard_normal((120, d + 1)).astype(np.float32)
        got = (cg.gramian(cg.ValueGradientKernel(kern), Xd) @ torch.from_numpy(Av).cuda().reshape(-1)).cpu().numpy()
        want = oracle.valgrad_mul(None, ok, X, X, Av.reshape(-1), dtype=np.float32)
        assert relerr(got, want) <= 1e-5, (shift, d, relerr(got, want))


def test_fp64_exponential_entrywise(cg, oracle):
    """The library's own fp64 exponential (csrc/profiles.hpp exp2_scaled_nonpos: EQ values everywhere, Matern / exponential /
    gamma-exponential derivative profiles) ENTRY BY ENTRY against numpy over the whole range of its argument: Matrix(G) of EQ on
    collinear points with squared distances from 0 to 3000 (exp(-s/2) down to the denormals and 0), and the gradient MVM of the
    exponential profile on one pair per distance.  Relative error <= 2 ulp where the value is normal, absolute below."""
    m = 6000
    s = np.concatenate([np.linspace(0.0, 40.0, 2000), np.linspace(40.0, 1400.0, 2000), np.linspace(1400.0, 3000.0, 2000)])
    y = np.zeros((m, 2)); y[:, 0] = np.sqrt(s)
    x = np.zeros((1, 2))
    G = cg.gramian(cg.EQ(), torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    got = G.to_dense().cpu().numpy()[0]
    s_exact = y[:, 0] ** 2                                      # what the kernel sees (direct differences of these coordinates)
    ref = np.exp(-0.5 * s_exact)
    normal = ref > 1e-300
    assert np.max(np.abs(got[normal] / ref[normal] - 1.0)) <= 4.5e-16, np.max(np.abs(got[normal] / ref[normal] - 1.0))
    assert np.all(np.abs(got[~normal] - ref[~normal]) <= 1e-300) and got[-1] == 0.0
    assert got[0] == 1.0 and not np.any(np.isnan(got))
    # NaN in, NaN out; an infinite distance gives exactly 0
    yb = y[:4].copy(); yb[1, 0] = np.nan; yb[2, 0] = np.inf
    gb = cg.gramian(cg.EQ(), torch.from_numpy(x).cuda(), torch.from_numpy(yb).cuda()).to_dense().cpu().numpy()[0]
    assert gb[0] == 1.0 and np.isnan(gb[1]) and gb[2] == 0.0
    # exp(-r) of the exponential profile through the gradient kernel (d = 2, one row against every column, a = e_1 per column)
    r = np.sqrt(s_exact[1:])                                    # r = 0 is singular for this profile (src/stationary.jl:56-60)
    K = cg.gramian(cg.GradientKernel(cg.Exp()), torch.from_numpy(x).cuda(), torch.from_numpy(y[1:]).cuda())
    a = np.zeros((m - 1, 2)); a[:, 1] = 1.0                     # picks column 2 of each block: r r^T has no (1,2) entry here
    out = (K @ torch.from_numpy(a.reshape(-1)).cuda()).cpu().numpy()
    ko = oracle.Kernel(oracle.EXP)
    ref_g = oracle.grad_mul(np.zeros(2), ko, x, y[1:], a.reshape(-1), 1.0, 0.0, np.float64)
    assert relerr(out, ref_g) <= 1e-14, relerr(out, ref_g)


def test_many_right_hand_sides_on_the_matrix_cores(cg, oracle):
    """mul!(B, G, A) with p >= 12 columns (src/gramian.jl:89-99) runs the accumulation as an fp32 GEMM on the matrix cores
    (dense_mfma_mrhs_kernel: the evaluated tile, transposed, is the B operand of v_mfma_f32_32x32x2_f32): ragged n / m / p, several
    profiles, alpha / beta, against the oracle column by column and against the four-at-a-time VALU form (option mfma_mrhs = 0)."""
    rng = np.random.default_rng(31)
    try:
        for (kern, ko, n, m, d, p) in ((cg.EQ(), oracle.Kernel(oracle.EQ), 1000, 1537, 3, 13), (cg.MaternP(2), oracle.Kernel(oracle.MATERNP, p=2), 777, 2050, 5, 40),
                                       (cg.RQ(1.5), oracle.Kernel(oracle.RQ, param=1.5), 2049, 999, 8, 70), (cg.Dot() ** 2, oracle.Kernel(oracle.DOT, power=2), 515, 1025, 4, 33),
                                       (cg.Lengthscale(cg.EQ(), 0.8) * cg.Cauchy(), None, 640, 700, 3, 12), (cg.EQ(), oracle.Kernel(oracle.EQ), 300, 4100, 20, 130),
                                       (cg.EQ(), oracle.Kernel(oracle.EQ), 5, 7, 3, 13), (cg.EQ(), oracle.Kernel(oracle.EQ), 1, 1, 1, 12), (cg.EQ(), oracle.Kernel(oracle.EQ), 33, 1000, 8, 75),
                                       (cg.EQ(), oracle.Kernel(oracle.EQ), 64, 64, 3, 65)):
            X = (rng.standard_normal((n, d)) * 0.7).astype(np.float32); Y = (rng.standard_normal((m, d)) * 0.7).astype(np.float32)
            A = rng.standard_normal((m, p)).astype(np.float32); B0 = rng.standard_normal((n, p)).astype(np.float32)
            G = cg.gramian(kern, torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
            outs = {}
            for opt in (0, -1):
                cg.set_option("mfma_mrhs", opt)
                Bd = torch.from_numpy(B0.copy()).cuda()
                cg.mul_(Bd, G, torch.from_numpy(A).cuda(), 0.6, -1.2)
                outs[opt] = Bd.cpu().numpy()
            assert cg.get_info("last_dense_path") == 2
            assert relerr(outs[-1], outs[0]) <= 1e-5, (type(kern).__name__, p, relerr(outs[-1], outs[0]))
            if ko is not None:
                ref = 0.6 * np.stack([oracle.mul(None, ko, X, Y, A[:, c], dtype=np.float32) for c in range(p)], 1) - 1.2 * B0
                assert relerr(outs[-1], ref) <= 1e-5, (type(kern).__name__, p, relerr(outs[-1], ref))
    finally:
        cg.set_option("mfma_mrhs", -1)


def test_fp64_rational_quadratic_entrywise(cg, oracle):
    """The library's own fp64 power for the rational-quadratic profile (csrc/profiles.hpp rq_pow: log2 by frexp + atanh series, then
    the library's exp2) ENTRY BY ENTRY against mpmath-free numpy over the range of its argument and several alpha: Matrix(G) on collinear
    points, relative error <= (4 + alpha log2 u) ulp, u = 1 exactly 1, NaN / inf; and the gradient MVM against the oracle."""
    m = 4000
    s = np.concatenate([np.linspace(0.0, 1e-6, 500), np.linspace(1e-6, 50.0, 2000), np.geomspace(50.0, 1e12, 1500)])
    y = np.zeros((m, 1)); y[:, 0] = np.sqrt(s)
    x = np.zeros((1, 1))
    s_exact = y[:, 0] ** 2
    for alpha in (0.37, 1.0, 2.5, 30.0):
        got = cg.gramian(cg.RQ(alpha), torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()).to_dense().cpu().numpy()[0]
        u = 1.0 + s_exact / (2.0 * alpha)
        ref = np.exp(-alpha * np.log(u))                          # numpy's log / exp are correctly rounded to < 1 ulp each
        ok = ref > 1e-300
        err = np.max(np.abs(got[ok] / ref[ok] - 1.0))
        # u^(-alpha) = exp2(-x), x = alpha log2 u: one rounding of x costs ln(2) |x| eps in the result (|x| <= ~1000 over the normal range)
        assert err <= 2.3e-16 * (4 + alpha * np.max(np.log2(u[ok]))), (alpha, err, alpha * np.max(np.log2(u[ok])))
        assert got[0] == 1.0 and not np.any(np.isnan(got))
    yb = np.array([[0.0], [np.nan], [np.inf]])
    gb = cg.gramian(cg.RQ(1.5), torch.from_numpy(x).cuda(), torch.from_numpy(yb).cuda()).to_dense().cpu().numpy()[0]
    assert gb[0] == 1.0 and np.isnan(gb[1]) and gb[2] == 0.0
    rng = np.random.default_rng(8)
    X = rng.standard_normal((300, 5)); Y = rng.standard_normal((211, 5)) * 3.0; a = rng.standard_normal(211 * 5)
    K = cg.gramian(cg.GradientKernel(cg.RQ(0.8)), torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
    out = (K @ torch.from_numpy(a).cuda()).cpu().numpy()
    assert relerr(out, oracle.grad_mul(None, oracle.Kernel(oracle.RQ, param=0.8), X, Y, a, 1.0, 0.0, np.float64)) <= 1e-13


def test_fp64_gamma_exponential_entrywise(cg, oracle):
    """The library's own fp64 power for the gamma-exponential profile (csrc/profiles.hpp pow_pos: s^(gamma/2) by log2_ge1 + the
    library's exp2, either sign of log2 s) ENTRY BY ENTRY against numpy over tiny, moderate and large s and several gamma: Matrix(G) on
    collinear points; k(0) = 1 exactly, NaN propagates, inf gives 0; and the gradient MVM against the oracle
    (src/stationary.jl:96-111: exp(-r^gamma / 2))."""
    s = np.concatenate([[0.0], np.geomspace(1e-300, 1e-6, 800), np.linspace(1e-6, 40.0, 2000), np.geomspace(40.0, 1e6, 1200)])
    m = len(s)
    y = np.zeros((m, 1)); y[:, 0] = np.sqrt(s)
    x = np.zeros((1, 1))
    s_exact = y[:, 0] ** 2
    for gamma in (0.3, 1.0, 1.3, 1.999, 2.0):
        got = cg.gramian(cg.GammaExp(gamma), torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()).to_dense().cpu().numpy()[0]
        with np.errstate(divide="ignore", over="ignore"):
            t = np.where(s_exact > 0, np.exp(0.5 * gamma * np.log(np.where(s_exact > 0, s_exact, 1.0))), 0.0)
        ref = np.exp(-0.5 * t)
        ok = ref > 1e-300
        lg = np.abs(0.5 * gamma * np.log2(np.where(s_exact > 0, s_exact, 1.0)))
        # t = exp2(x), x = (gamma/2) log2 s rounded once: relative error ln(2) |x| eps in t, times t/2 in exp(-t/2)
        bound = 2.3e-16 * (4 + (4 + lg) * np.maximum(0.5 * t, 1.0))
        err = np.abs(got[ok] / ref[ok] - 1.0)
        assert np.all(err <= bound[ok]), (gamma, float(np.max(err / bound[ok])))
        assert got[0] == 1.0 and not np.any(np.isnan(got))
    yb = np.array([[0.0], [np.nan], [np.inf]])
    gb = cg.gramian(cg.GammaExp(1.5), torch.from_numpy(x).cuda(), torch.from_numpy(yb).cuda()).to_dense().cpu().numpy()[0]
    assert gb[0] == 1.0 and np.isnan(gb[1]) and gb[2] == 0.0
    rng = np.random.default_rng(9)
    X = rng.standard_normal((300, 5)); Y = rng.standard_normal((211, 5)) * 2.0; a = rng.standard_normal(211 * 5)
    K = cg.gramian(cg.GradientKernel(cg.GammaExp(1.4)), torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
    out = (K @ torch.from_numpy(a).cuda()).cpu().numpy()
    assert relerr(out, oracle.grad_mul(None, oracle.Kernel(oracle.GAMMAEXP, param=1.4), X, Y, a, 1.0, 0.0, np.float64)) <= 1e-13
    av = rng.standard_normal(211)
    G = cg.gramian(cg.GammaExp(1.4), torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
    assert relerr((G @ torch.from_numpy(av).cuda()).cpu().numpy(), oracle.mul(None, oracle.Kernel(oracle.GAMMAEXP, param=1.4), X, Y, av, dtype=np.float64)) <= 1e-13


def test_fp64_sqrt_rcp_rsqrt_profiles_entrywise(cg, oracle):
    """The library's own fp64 square root / reciprocal / reciprocal square root (csrc/profiles.hpp cg_sqrt, cg_rcp, cg_rsqrt: hardware
    seed + Goldschmidt / cubic Newton refinement, no range scaling) through the profiles that use them, ENTRY BY ENTRY against numpy:
    Exponential exp(-sqrt s) (src/stationary.jl:60), MaternP (:117-158), Cauchy 1/(1+s) (:224), InverseMultiQuadratic 1/sqrt(s + c^2)
    (:235) over denormal, tiny, moderate, huge and special s; then the gradient MVMs of the same profiles against the oracle."""
    r = np.concatenate([[0.0], np.geomspace(1e-160, 1e-6, 700), np.linspace(1e-6, 30.0, 2500), np.geomspace(30.0, 1e150, 800)])
    y = r.reshape(-1, 1).copy()
    x = np.zeros((1, 1))
    s = y[:, 0] ** 2
    eps = 2.3e-16

    def dense_row(k, pts=y):
        return cg.gramian(k, torch.from_numpy(x).cuda(), torch.from_numpy(pts).cuda()).to_dense().cpu().numpy()[0]

    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        got = dense_row(cg.Exp())
        ref = np.exp(-np.sqrt(s)); ok = ref > 1e-300
        # one rounding of r = sqrt(s) costs r eps in exp(-r)
        assert np.all(np.abs(got[ok] / ref[ok] - 1.0) <= eps * (4 + 2 * np.sqrt(s[ok]))), float(np.max(np.abs(got[ok] / ref[ok] - 1.0)))
        assert got[0] == 1.0 and np.all(got[~ok] <= 1e-300) and not np.any(np.isnan(got))
        for p in (1, 2, 3):
            got = dense_row(cg.MaternP(p))
            ref = oracle.matrix(oracle.Kernel(oracle.MATERNP, p=p), x, y)[0]; ok = ref > 1e-300
            rr = np.sqrt((2 * p + 1) * s[ok])
            assert np.all(np.abs(got[ok] / ref[ok] - 1.0) <= eps * (8 + 2 * rr)), (p, float(np.max(np.abs(got[ok] / ref[ok] - 1.0))))
            assert got[0] == 1.0
        got = dense_row(cg.Cauchy())
        ref = 1.0 / (1.0 + s); ok = ref > 1e-300
        assert np.all(np.abs(got[ok] / ref[ok] - 1.0) <= 3 * eps), float(np.max(np.abs(got[ok] / ref[ok] - 1.0)))
        assert got[0] == 1.0 and not np.any(np.isnan(got))
        for c in (1.0, 0.37, 1e-3):
            got = dense_row(cg.InverseMultiQuadratic(c))
            ref = 1.0 / np.sqrt(s + c * c); ok = ref > 1e-300
            assert np.all(np.abs(got[ok] / ref[ok] - 1.0) <= 4 * eps), (c, float(np.max(np.abs(got[ok] / ref[ok] - 1.0))))
    yb = np.array([[0.0], [np.nan], [np.inf], [5e-324], [1e-162]])
    for k, at_inf in ((cg.Exp(), 0.0), (cg.Cauchy(), 0.0), (cg.InverseMultiQuadratic(1.0), 0.0)):
        gb = dense_row(k, yb)
        assert gb[0] == 1.0 and np.isnan(gb[1]) and gb[2] == at_inf and gb[3] == 1.0 and gb[4] == 1.0, (type(k).__name__, gb)
    gm = dense_row(cg.MaternP(2), yb)
    assert gm[0] == 1.0 and np.isnan(gm[1]) and gm[3] == 1.0 and gm[4] == 1.0
    rng = np.random.default_rng(10)
    X = rng.standard_normal((260, 4)); Y = rng.standard_normal((190, 4)) * 1.5; a = rng.standard_normal(190 * 4)
    for kg, ko in ((cg.Exp(), oracle.Kernel(oracle.EXP)), (cg.MaternP(2), oracle.Kernel(oracle.MATERNP, p=2)), (cg.Cauchy(), oracle.Kernel(oracle.CAUCHY)),
                   (cg.InverseMultiQuadratic(0.8), oracle.Kernel(oracle.IMQ, param=0.8)), (cg.RQ(1.2), oracle.Kernel(oracle.RQ, param=1.2))):
        K = cg.gramian(cg.GradientKernel(kg), torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
        out = (K @ torch.from_numpy(a).cuda()).cpu().numpy()
        assert relerr(out, oracle.grad_mul(None, ko, X, Y, a, 1.0, 0.0, np.float64)) <= 1e-13, type(kg).__name__


def test_fp64_symmetric_direct_kernel(cg, oracle):
    """gramian(k, x) * a in fp64 on the direct-difference path evaluates the upper triangle once (csrc/dense_mvm.hpp dense_sym_kernel; the
    reference loops over all n*n entries, src/gramian.jl:78-87): against the fp64 oracle and against the all-entries kernel, for sizes that
    are and are not multiples of the 64-row blocks / column chunks, every single-profile family, alpha / beta (beta = 0 must not read y:
    NaN-filled), a Lengthscale, and forced on at a size below the automatic threshold."""
    rng = np.random.default_rng(11)
    fams = [(cg.EQ(), oracle.Kernel(oracle.EQ)), (cg.MaternP(2), oracle.Kernel(oracle.MATERNP, p=2)), (cg.Exp(), oracle.Kernel(oracle.EXP)),
            (cg.RQ(1.3), oracle.Kernel(oracle.RQ, param=1.3)), (cg.Cauchy(), oracle.Kernel(oracle.CAUCHY)),
            (cg.GammaExp(1.5), oracle.Kernel(oracle.GAMMAEXP, param=1.5)), (cg.InverseMultiQuadratic(0.9), oracle.Kernel(oracle.IMQ, param=0.9)),
            (cg.MaternP(5), oracle.Kernel(oracle.MATERNP, p=5)), (cg.ExponentialDot(), oracle.Kernel(oracle.EXPDOT)),
            (cg.Lengthscale(cg.MaternP(1), 0.6), oracle.Kernel(oracle.MATERNP, p=1, lengthscale=0.6))]
    try:
        for n, d in ((1, 2), (63, 3), (64, 3), (65, 1), (777, 3), (1000, 8), (2049, 5)):
            X = rng.standard_normal((n, d)) * (0.4 if d > 3 else 1.0); a = rng.standard_normal(n)
            Xd = torch.from_numpy(X).cuda(); ad = torch.from_numpy(a).cuda()
            for kg, ko in fams if n in (777, 1000) else fams[:2]:
                G = cg.gramian(kg, Xd)
                cg.set_option("dense_sym", 1)
                ys = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
                G.mul_(ys, ad)
                assert cg.get_info("last_dense_sym") == 1, (n, d, type(kg).__name__)
                cg.set_option("dense_sym", 0)
                yg = (G @ ad); assert cg.get_info("last_dense_sym") == 0
                ref = oracle.mul(None, ko, X, X, a, dtype=np.float64)
                assert relerr(ys.cpu().numpy(), ref) <= 1e-13, (n, d, type(kg).__name__, relerr(ys.cpu().numpy(), ref))
                assert relerr(ys.cpu().numpy(), yg.cpu().numpy()) <= 1e-13
        # alpha / beta, and the automatic rule (n >= 8192, same point set, one right-hand side, fp64)
        n, d = 8300, 3
        X = rng.standard_normal((n, d)); a = rng.standard_normal(n); y0 = rng.standard_normal(n)
        Xd = torch.from_numpy(X).cuda(); ad = torch.from_numpy(a).cuda()
        cg.set_option("dense_sym", -1)
        G = cg.gramian(cg.MaternP(2), Xd)
        yd = torch.from_numpy(y0).cuda()
        G.mul_(yd, ad, -0.7, 1.3)
        assert cg.get_info("last_dense_sym") == 1
        ref = oracle.mul(y0.copy(), oracle.Kernel(oracle.MATERNP, p=2), X, X, a, -0.7, 1.3, np.float64)
        assert relerr(yd.cpu().numpy(), ref) <= 1e-13
        G2 = cg.gramian(cg.MaternP(2), Xd, torch.from_numpy(X.copy()).cuda())      # equal values, different handles: not the symmetric path
        G2 @ ad; assert cg.get_info("last_dense_sym") == 0
        (G @ torch.from_numpy(rng.standard_normal((n, 2))).cuda()); assert cg.get_info("last_dense_sym") == 0      # matrix right-hand side
        Gf = cg.gramian(cg.Exp(), Xd.float()); Gf @ ad.float(); assert cg.get_info("last_dense_sym") == 0             # fp32
        # symmetric result is symmetric in the bilinear form: a' (G b) == b' (G a)
        b = torch.from_numpy(rng.standard_normal(n)).cuda()
        assert abs(float(ad @ (G @ b)) - float(b @ (G @ ad))) <= 1e-10 * n
        # the Krylov caller on top (src/gramian.jl:229-238): (G + sigma^2 I) \ b eagerly and as a replayed HIP graph, every MVM on the
        # symmetric kernel; the two agree and the residual through the ALL-entries kernel confirms the solution
        S = G + 0.5 * torch.ones(n, device="cuda", dtype=torch.float64)
        xe, ie = cg.cg(S, b, reltol=1e-9)
        xg, ig = cg.cg(S, b, reltol=1e-9, graph=True, check_every=4)
        assert ie["converged"] and ig["converged"] and ig["graph"] and cg.get_info("last_dense_sym") == 1
        assert relerr(xg.cpu().numpy(), xe.cpu().numpy()) <= 1e-7
        cg.set_option("dense_sym", 0)
        res = (G @ xe) + 0.5 * xe - b
        assert cg.get_info("last_dense_sym") == 0 and float(res.norm() / b.norm()) <= 1e-8
    finally:
        cg.set_option("dense_sym", -1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n,m,d", [(70001, 50003, 3), (33333, 66667, 8), (40000, 40001, 32), (9, 5000, 16), (5000, 7, 64), (20011, 30011, 5)])
def test_dot_gramian_long_ragged_point_sets(cg, oracle, dtype, n, m, d):
    """The same product on point sets long enough for several row slabs, trips of four (eight) rows in flight and the last workgroup's slab sum,
    with ragged ends; against numpy in fp64."""
    rng = np.random.default_rng(n + m + d)
    dt = npdt(dtype)
    X = rng.standard_normal((n, d)).astype(dt); Y = rng.standard_normal((m, d)).astype(dt); a = rng.standard_normal(m).astype(dt); y0 = rng.standard_normal(n).astype(dt)
    G = cg.gramian(cg.Dot(), torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
    ref = X.astype(np.float64) @ (Y.astype(np.float64).T @ a.astype(np.float64))
    yd = torch.full((n,), float("nan"), dtype=dtype, device="cuda")
    cg.mul_(yd, G, torch.from_numpy(a).cuda())
    assert cg.get_info("last_dense_path") == 4
    assert relerr(yd.cpu().numpy(), ref) <= (2e-6 if dtype == torch.float32 else 1e-13)
    yd = torch.from_numpy(y0.copy()).cuda()
    cg.mul_(yd, G, torch.from_numpy(a).cuda(), -0.7, 1.3)
    assert relerr(yd.cpu().numpy(), -0.7 * ref + 1.3 * y0.astype(np.float64)) <= (2e-6 if dtype == torch.float32 else 1e-13)
