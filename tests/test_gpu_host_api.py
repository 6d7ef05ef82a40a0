"""GPU tier: the C ABI exactly as a `ccall` shim would use it — HOST pointers (numpy arrays), library-side staging — for
every entry point of include/covgram.h, plus the edge cases the reference's loops accept (empty inputs, many RHS)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    den = np.linalg.norm(ref)
    return np.linalg.norm(b - ref) / (den if den > 0 else 1.0)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def ctx(cg):
    lib = cg._ffi.lib()
    h = cg._ffi._P()
    cg._ffi.check(lib.covgram_ctx_create(C.byref(h), 0, None))     # NULL stream = the default stream
    yield h
    assert lib.covgram_ctx_destroy(h) == 0


def make_points(cg, ctx, X):
    lib = cg._ffi.lib()
    h = cg._ffi._P()
    X = np.ascontiguousarray(X)
    cg._ffi.check(lib.covgram_points_create(ctx, C.byref(h), P(X), X.shape[0], X.shape[1],
                                            cg._ffi.F64 if X.dtype == np.float64 else cg._ffi.F32, cg._ffi.HOST))
    return h


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_host_pointer_mvm_matrix_gradient(cg, oracle, ctx, dt):
    lib, f = cg._ffi.lib(), cg._ffi
    tol = 1e-5 if dt == np.float32 else 1e-12
    rng = np.random.default_rng(3)
    n, m, d = 301, 123, 3
    X = rng.standard_normal((n, d)).astype(dt); Y = rng.standard_normal((m, d)).astype(dt)
    hx, hy = make_points(cg, ctx, X), make_points(cg, ctx, Y)
    nn, dd, tt = C.c_int64(), C.c_int32(), C.c_int32()
    assert lib.covgram_points_info(hx, C.byref(nn), C.byref(dd), C.byref(tt)) == 0 and (nn.value, dd.value) == (n, d)
    spec = cg.device_spec(cg.Lengthscale(cg.MaternP(2), 0.8) * 1.7)
    ko = oracle.Kernel(oracle.MATERNP, p=2, lengthscale=0.8, scale=1.7)
    # vector, 7-column and (fp32: fp32 matrix cores, dense_mfma_mrhs_kernel) 13- / 40- / 70-column right-hand sides, leading dimensions
    # larger than the extents
    for nrhs in (1, 7, 13, 40, 70):
        lda, ldy = m + 5, n + 3
        A = np.zeros((nrhs, lda), dtype=dt); A[:, :m] = rng.standard_normal((nrhs, m))          # column-major m x nrhs, lda
        Yo = np.zeros((nrhs, ldy), dtype=dt); Yo[:, :n] = rng.standard_normal((nrhs, n))
        Y0 = Yo.copy()
        f.check(lib.covgram_mvm(ctx, C.byref(spec), hx, hy, P(A), lda, P(Yo), ldy, nrhs, -0.7, 1.3, f.HOST))
        ref = oracle.mul(Y0[:, :n].T, ko, X, Y, A[:, :m].T, -0.7, 1.3, dt)
        assert relerr(Yo[:, :n].T, ref) <= tol
        assert np.array_equal(Yo[:, n:], Y0[:, n:])                                              # padding rows untouched
    # Matrix(G)
    M = np.zeros((m, n + 2), dtype=dt)
    f.check(lib.covgram_matrix(ctx, C.byref(spec), hx, hy, P(M), n + 2, f.HOST))
    assert relerr(M[:, :n].T, oracle.matrix(ko, X, Y, dt)) <= tol
    # row-shard slice == the corresponding rows
    hs = f._P()
    f.check(lib.covgram_points_slice(hx, 100, 50, C.byref(hs)))
    a = rng.standard_normal(m).astype(dt); ys = np.zeros(50, dtype=dt)
    f.check(lib.covgram_mvm(ctx, C.byref(spec), hs, hy, P(a), m, P(ys), 50, 1, 1.0, 0.0, f.HOST))
    assert relerr(ys, oracle.mul(None, ko, X[100:150], Y, a, dtype=dt)) <= tol
    assert lib.covgram_points_slice(hx, 290, 50, C.byref(f._P())) == f.EINVAL
    # gradient MVM through host pointers
    ag = rng.standard_normal(m * d).astype(dt); yg = rng.standard_normal(n * d).astype(dt); yg0 = yg.copy()
    f.check(lib.covgram_grad_mvm(ctx, C.byref(spec), hx, hy, P(ag), ag.size, P(yg), yg.size, 1, 0.4, -0.9, f.HOST))
    assert relerr(yg, oracle.grad_mul(yg0, ko, X, Y, ag, 0.4, -0.9, dt)) <= (1e-5 if dt == np.float32 else 1e-12)
    # value-gradient blocks and a composite kernel through host pointers (the composite goes in through its head)
    av = rng.standard_normal(m * (d + 1)).astype(dt); yv = rng.standard_normal(n * (d + 1)).astype(dt); yv0 = yv.copy()
    f.check(lib.covgram_valgrad_mvm(ctx, C.byref(spec), hx, hy, P(av), av.size, P(yv), yv.size, 1, 0.4, -0.9, f.HOST))
    assert relerr(yv, oracle.valgrad_mul(yv0, ko, X, Y, av, 0.4, -0.9, dt)) <= (1e-5 if dt == np.float32 else 1e-12)
    comp = cg.device_spec(1.5 * cg.Lengthscale(cg.MaternP(2), 0.8) + 0.5 * cg.EQ())
    kc = oracle.Composite(((oracle.Kernel(oracle.MATERNP, p=2, lengthscale=0.8, scale=1.5),), (oracle.Kernel(oracle.EQ, scale=0.5),)))
    yc = np.zeros(n, dtype=dt)
    f.check(lib.covgram_mvm(ctx, f.kref(comp), hx, hy, P(a), m, P(yc), n, 1, 1.0, 0.0, f.HOST))
    assert relerr(yc, oracle.mul(None, kc, X, Y, a, dtype=dt)) <= tol
    f.check(lib.covgram_grad_mvm(ctx, f.kref(comp), hx, hy, P(ag), ag.size, P(yg), yg.size, 1, 1.0, 0.0, f.HOST))
    assert relerr(yg, oracle.grad_mul(None, kc, X, Y, ag, dtype=dt)) <= (1e-5 if dt == np.float32 else 1e-12)
    Mc = np.zeros((m, n), dtype=dt)
    f.check(lib.covgram_matrix(ctx, f.kref(comp), hx, hy, P(Mc), n, f.HOST))
    assert relerr(Mc.T, oracle.matrix(kc, X, Y, dt)) <= tol
    bad = cg.device_spec(cg.EQ() + cg.RQ(1.0)); bad.factors[1].trait = f.DOTPRODUCT
    assert lib.covgram_mvm(ctx, f.kref(bad), hx, hy, P(a), m, P(yc), n, 1, 1.0, 0.0, f.HOST) == f.EINVAL
    v = C.c_int64(-1)
    assert lib.covgram_ctx_get_info(ctx, b"last_dense_path", C.byref(v)) == 0 and v.value in (1, 2)
    assert lib.covgram_ctx_get_info(ctx, b"num_cus", C.byref(v)) == 0 and v.value >= 64
    assert lib.covgram_ctx_get_info(ctx, b"nonsense", C.byref(v)) == f.EINVAL
    # dimension mismatch -> status, not a crash
    hz = make_points(cg, ctx, rng.standard_normal((5, 2)).astype(dt))
    assert lib.covgram_mvm(ctx, C.byref(spec), hx, hz, P(a), m, P(ys), n, 1, 1.0, 0.0, f.HOST) == f.EINVAL
    assert b"same length" in lib.covgram_last_error()
    for h in (hs, hx, hy, hz):
        assert lib.covgram_points_destroy(h) == 0


def test_host_pointer_toeplitz_kron_lowrank(cg, oracle, ctx):
    lib, f = cg._ffi.lib(), cg._ffi
    rng = np.random.default_rng(4)
    # Toeplitz, non-symmetric, both the rocFFT path (small) and the four-step path (N >= 16384)
    for n, m in ((300, 200), (9000, 9000)):
        vc = rng.standard_normal(n); vr = rng.standard_normal(m); vr[0] = vc[0]
        a = rng.standard_normal(m); y = rng.standard_normal(n); y0 = y.copy()
        h = f._P()
        f.check(lib.covgram_toeplitz_create(ctx, C.byref(h), P(vc), P(vr), n, m, f.F64, f.HOST, 0))
        f.check(lib.covgram_toeplitz_mvm(h, P(a), P(y), 0.5, 2.0, f.HOST))
        assert relerr(y, oracle.toeplitz_mul(y0, vc, vr, a, 0.5, 2.0)) <= 1e-10
        assert lib.covgram_toeplitz_destroy(h) == 0
    # circulant of odd length
    vc = rng.standard_normal(33); a = rng.standard_normal(33); y = np.zeros(33)
    h = f._P()
    f.check(lib.covgram_toeplitz_create(ctx, C.byref(h), P(vc), None, 33, 33, f.F64, f.HOST, 1))
    f.check(lib.covgram_toeplitz_mvm(h, P(a), P(y), 1.0, 0.0, f.HOST))
    assert relerr(y, oracle.toeplitz_dense(vc, circulant=True) @ a) <= 1e-12
    assert lib.covgram_toeplitz_destroy(h) == 0
    # Kronecker with host factors (column-major with padding in the leading dimension)
    shapes = [(3, 5), (4, 2), (2, 6)]
    Fs = [rng.standard_normal(s) for s in shapes]
    lds = [s[0] + 1 for s in shapes]
    bufs = []
    for Fm, ld in zip(Fs, lds):
        b = np.zeros((Fm.shape[1], ld)); b[:, :Fm.shape[0]] = Fm.T
        bufs.append(b)
    ptrs = (f._P * 3)(*[P(b) for b in bufs])
    rows = (C.c_int64 * 3)(*[s[0] for s in shapes]); cols = (C.c_int64 * 3)(*[s[1] for s in shapes]); ldarr = (C.c_int64 * 3)(*lds)
    av = rng.standard_normal(5 * 2 * 6); yv = rng.standard_normal(3 * 4 * 2); yv0 = yv.copy()
    f.check(lib.covgram_kron_mvm(ctx, ptrs, rows, cols, ldarr, 3, f.F64, P(av), av.size, P(yv), yv.size, 1, 1.5, -0.5, f.HOST))
    assert relerr(yv, 1.5 * np.kron(np.kron(Fs[0], Fs[1]), Fs[2]) @ av - 0.5 * yv0) <= 1e-12
    # low rank
    n, m, r = 500, 300, 5
    U = rng.standard_normal((n, r)); V = rng.standard_normal((m, r))
    Uc = np.asfortranarray(U); Vc = np.asfortranarray(V)
    a = rng.standard_normal(m); y = rng.standard_normal(n); y0 = y.copy()
    f.check(lib.covgram_lowrank_mvm(ctx, P(Uc), n, P(Vc), m, n, m, r, f.F64, P(a), m, P(y), n, 1, 2.0, 0.25, f.HOST))
    assert relerr(y, oracle.lowrank_mul(y0, U, V, a, 2.0, 0.25)) <= 1e-12


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_host_pointer_direct_toeplitz_solvers(cg, oracle, ctx, dt):
    """covgram_toeplitz_durbin / _levinson / _trench with HOST pointers — the form julia/CovGram.jl's `\\`, durbin!, levinson!, trench!
    (src/toeplitz.jl:12-111) use — on the on-chip kernel (n <= 16384), on the global-memory kernel above it, at n = 1, and above the
    size cap (COVGRAM_EUNSUPPORTED, nothing launched)."""
    lib, f = cg._ffi.lib(), cg._ffi
    code = f.F64 if dt == np.float64 else f.F32
    tol = 1e-9 if dt == np.float64 else 2e-3
    rng = np.random.default_rng(77)
    for m in (1, 2, 300, 16384, 16500):
        vc = (1.5 * (np.arange(m) == 0) + np.exp(-np.abs(np.linspace(0.0, 3.0, m)))).astype(np.float64)   # diagonal 2.5: well conditioned
        r = (vc[1:] / vc[0]).astype(dt)
        b = rng.standard_normal(m).astype(dt)
        x = np.full(m, np.nan, dtype=dt)
        f.check(lib.covgram_toeplitz_levinson(ctx, P(r) if m > 1 else None, P(b), m, P(x), code, f.HOST))
        r64, b64 = r.astype(np.float64), b.astype(np.float64)
        if m <= 300:
            K = np.array([[1.0 if i == j else r64[abs(i - j) - 1] for j in range(m)] for i in range(m)])
            assert relerr(x, np.linalg.solve(K, b64)) <= tol, ("levinson", m)
        else:                                                       # residual through the FFT MVM of the same matrix
            T = cg.SymmetricToeplitz(torch.from_numpy(np.concatenate([[1.0], r64])).cuda())
            assert relerr((T @ torch.from_numpy(x.astype(np.float64)).cuda()).cpu().numpy(), b64) <= tol, ("levinson", m)
        if m > 1:
            y = np.full(m - 1, np.nan, dtype=dt)
            f.check(lib.covgram_toeplitz_durbin(ctx, P(r), m - 1, P(y), code, f.HOST))
            assert relerr(y, oracle.durbin(r64)) <= tol, ("durbin", m)
        if m <= 300:
            B = np.full((m, m + 1), np.nan, dtype=dt)               # column-major with ldb = m + 1: as a (m, m + 1) C array of columns
            Bc = np.zeros((m, m + 1), dtype=dt)
            f.check(lib.covgram_toeplitz_trench(ctx, P(r) if m > 1 else None, m, P(Bc), m + 1, code, f.HOST))
            Binv = Bc[:, :m].T                                       # entry (i, j) at i + j * ldb
            assert relerr(Binv @ K, np.eye(m)) <= tol * 10, ("trench", m)
    big = int(lib.covgram_version() and 65536) + 1
    z = np.zeros(8, dtype=np.float64)
    assert lib.covgram_toeplitz_levinson(ctx, P(z), P(z), big, P(z), f.F64, f.HOST) == f.EUNSUPPORTED
    assert lib.covgram_toeplitz_durbin(ctx, P(z), big, P(z), f.F64, f.HOST) == f.EUNSUPPORTED
    assert lib.covgram_toeplitz_levinson(ctx, P(z), P(z), 0, P(z), f.F64, f.HOST) == f.EINVAL


def test_empty_and_degenerate_inputs(cg, oracle):
    """Edge cases the reference's loops accept: no columns (y <- beta y), no rows, one point, many RHS, d at the boundaries."""
    dev = "cuda"
    x = torch.randn(17, 3, device=dev, dtype=torch.float64)
    empty = torch.zeros(0, 3, device=dev, dtype=torch.float64)
    G = cg.gramian(cg.EQ(), x, empty)                       # 17 x 0
    y = torch.randn(17, device=dev, dtype=torch.float64); y0 = y.clone()
    cg.mul_(y, G, torch.zeros(0, device=dev, dtype=torch.float64), 1.0, 0.5)
    assert torch.allclose(y, 0.5 * y0)
    yn = torch.full((17,), float("nan"), device=dev, dtype=torch.float64)
    cg.mul_(yn, G, torch.zeros(0, device=dev, dtype=torch.float64), 1.0, 0.0)
    assert torch.all(yn == 0)                               # beta == 0: zero-filled, NaN ignored (gramian.jl:80)
    G0 = cg.gramian(cg.EQ(), empty, x)                      # 0 x 17
    assert (G0 @ torch.randn(17, device=dev, dtype=torch.float64)).shape == (0,)
    Kg = cg.gramian(cg.GradientKernel(cg.EQ()), x, empty)
    yg = torch.randn(51, device=dev, dtype=torch.float64); yg0 = yg.clone()
    cg.mul_(yg, Kg, torch.zeros(0, device=dev, dtype=torch.float64), 1.0, -2.0)
    assert torch.allclose(yg, -2.0 * yg0)
    # d exactly at the lane-owned limits and one past them (dense 64/65; gradient fp64 48/49)
    rng = np.random.default_rng(12)
    for d in (64, 65):
        X = rng.standard_normal((130, d)) / np.sqrt(d); a = rng.standard_normal(130)
        Gd = cg.gramian(cg.EQ(), torch.from_numpy(X).cuda())
        assert relerr((Gd @ torch.from_numpy(a).cuda()).cpu().numpy(), oracle.mul(None, oracle.Kernel(oracle.EQ), X, X, a)) <= 1e-12
    for d in (48, 49):
        X = rng.standard_normal((40, d)) / np.sqrt(d); a = rng.standard_normal(40 * d)
        Kd = cg.gramian(cg.GradientKernel(cg.RQ(1.0)), torch.from_numpy(X).cuda())
        assert relerr((Kd @ torch.from_numpy(a).cuda()).cpu().numpy(), oracle.grad_mul(None, oracle.Kernel(oracle.RQ, param=1.0), X, X, a)) <= 1e-12
    # 9 right-hand sides (three kernel passes of <= 4 columns), non-contiguous inputs
    X = rng.standard_normal((200, 2)); A = rng.standard_normal((200, 9))
    Gm = cg.gramian(cg.Cauchy(), torch.from_numpy(X).cuda())
    At = torch.from_numpy(np.asfortranarray(A)).cuda()      # non-C-contiguous tensor
    assert relerr((Gm @ At).cpu().numpy(), oracle.mul(None, oracle.Kernel(oracle.CAUCHY), X, X, A)) <= 1e-12
    one = torch.randn(1, 3, device=dev, dtype=torch.float32)
    assert abs(float((cg.gramian(cg.EQ(), one) @ torch.ones(1, device=dev))[0]) - 1.0) < 1e-6


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_cg_step_is_the_cg_recurrence(cg, ctx, dt):
    """covgram_cg_step against the recurrences of IterativeSolvers.cg! written out in numpy (src/gramian.jl:229-238 calls it):
    alpha = rho / (p . Ap); x += alpha p; r -= alpha Ap; rho' = r . r; p = r + (rho' / rho) p — two consecutive steps (the
    second divides by the rho the first committed), ragged n, and n = 0."""
    lib, f = cg._ffi.lib(), cg._ffi
    DP = lambda t: f._P(t.data_ptr())                      # device pointers (this entry point takes no host memory)
    tdt = torch.float32 if dt == np.float32 else torch.float64
    tol = 2e-6 if dt == np.float32 else 1e-14
    rng = np.random.default_rng(17)
    for n in (1, 777, 70001):
        x, r, p = (rng.standard_normal(n).astype(dt) for _ in range(3))
        xd, rd, pd = (torch.from_numpy(v.copy()).cuda() for v in (x, r, p))
        scal = torch.zeros(2 + 512, dtype=tdt, device="cuda")
        scal[1] = torch.dot(rd, rd)
        x64, r64, p64 = (v.astype(np.float64) for v in (x, r, p))
        rho = float(r64 @ r64)
        for step in range(2):
            Ap = rng.standard_normal(n).astype(dt)
            Apd = torch.from_numpy(Ap).cuda()
            torch.cuda.synchronize()
            f.check(lib.covgram_cg_step(ctx, n, f.F32 if dt == np.float32 else f.F64, DP(xd), DP(rd), DP(pd), DP(Apd), DP(scal)))
            torch.cuda.synchronize()
            Ap64 = Ap.astype(np.float64)
            alpha = rho / float(p64 @ Ap64)
            x64 = x64 + alpha * p64; r64 = r64 - alpha * Ap64
            rho_new = float(r64 @ r64)
            p64 = r64 + (rho_new / rho) * p64
            assert abs(float(scal[0]) - rho) <= tol * abs(rho) * 10 and abs(float(scal[1]) - rho_new) <= tol * abs(rho_new) * 10
            rho = rho_new
            for got, want in ((xd, x64), (rd, r64), (pd, p64)):
                assert relerr(got.cpu().numpy(), want) <= tol * 50, (n, step, relerr(got.cpu().numpy(), want))
    f.check(lib.covgram_cg_step(ctx, 0, f.F64, None, None, None, None, None))
    assert lib.covgram_cg_step(ctx, 5, 7, None, None, None, None, None) == f.EINVAL


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_cg_step_shifted_adds_the_diagonal_term_and_leaves_the_norm(cg, ctx, dt):
    """covgram_cg_step_shifted: the step for A = G + Diagonal(d) (src/gramian.jl:55-60; mul! of the lazy sum,
    src/lazy_linear_algebra.jl:126-133) given Ap = G p of the Gramian alone — Ap comes back completed (Ap + d .* p), the
    recurrences use it, and |r| sits in scal[2 + 512]; diag = NULL is the plain step plus the norm."""
    lib, f = cg._ffi.lib(), cg._ffi
    DP = lambda t: f._P(t.data_ptr())
    tdt = torch.float32 if dt == np.float32 else torch.float64
    tol = 2e-6 if dt == np.float32 else 1e-14
    rng = np.random.default_rng(23)
    for n in (1, 1000, 70001):
        for with_diag in (True, False):
            x, r, p, d = (rng.standard_normal(n).astype(dt) for _ in range(4))
            d = np.abs(d) + dt(0.1)
            xd, rd, pd, dd = (torch.from_numpy(v.copy()).cuda() for v in (x, r, p, d))
            scal = torch.zeros(2 + 512 + 1, dtype=tdt, device="cuda")
            scal[1] = torch.dot(rd, rd)
            x64, r64, p64, d64 = (v.astype(np.float64) for v in (x, r, p, d))
            rho = float(r64 @ r64)
            for step in range(2):
                Gp = rng.standard_normal(n).astype(dt)
                Apd = torch.from_numpy(Gp.copy()).cuda()
                f.check(lib.covgram_cg_step_shifted(ctx, n, f.F32 if dt == np.float32 else f.F64, DP(xd), DP(rd), DP(pd), DP(Apd), DP(scal),
                                                    DP(dd) if with_diag else None))
                torch.cuda.synchronize()
                Ap64 = Gp.astype(np.float64) + (d64 * p64 if with_diag else 0.0)
                assert relerr(Apd.cpu().numpy(), Ap64) <= tol * 10
                alpha = rho / float(p64 @ Ap64)
                x64 = x64 + alpha * p64; r64 = r64 - alpha * Ap64
                rho_new = float(r64 @ r64)
                p64 = r64 + (rho_new / rho) * p64
                assert abs(float(scal[1]) - rho_new) <= tol * abs(rho_new) * 50
                assert abs(float(scal[2 + 512]) - np.sqrt(rho_new)) <= tol * np.sqrt(rho_new) * 50
                rho = rho_new
                for got, want in ((xd, x64), (rd, r64), (pd, p64)):
                    assert relerr(got.cpu().numpy(), want) <= tol * 200, (n, step, relerr(got.cpu().numpy(), want))
    f.check(lib.covgram_cg_step_shifted(ctx, 0, f.F64, None, None, None, None, None, None))
