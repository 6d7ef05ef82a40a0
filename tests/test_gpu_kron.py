"""Kronecker MVM kernels (csrc/kron.hip) against the oracle's mode products (oracle/covgram_oracle.py::kron_mul, the identity
KroneckerProducts 1.1.1 implements for the operators src/algebra.jl:91-95 and src/separable.jl:33-42 build).

Every kernel and every template branch: the fused last-two-modes pass (kron_pair_kernel: 16 NB1 slab columns, NB1 in {4, 8}; output chunks
of 16 NB2 columns incl. more than one chunk; ragged strips; small slabs go mode by mode), the single-mode kernels with trailing modes
(kron_mode_kernel) and without (kron_modet_kernel: q = 1, c_q > 128, too few slabs), aligned (16-byte vector) and unaligned tensors, fp32 and
fp64, alpha / beta, matrix right-hand sides, padded leading dimensions through the raw ABI, and the rocBLAS route for factors >= 1024."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {np.float32: 2e-6, np.float64: 1e-13}


def relerr(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    den = np.linalg.norm(ref)
    return np.linalg.norm(b - ref) / (den if den > 0 else 1.0)


def run_case(cg, oracle, shapes, dt, rng, alpha=1.0, beta=0.0, nrhs=0):
    """shapes: [(rows, cols)] per factor; nrhs = 0: vector.  Asymmetric random factors (a transposed or row/column-swapped kernel fails)."""
    tdt = torch.float32 if dt == np.float32 else torch.float64
    Fs = [rng.standard_normal(s).astype(dt) for s in shapes]
    nin = int(np.prod([s[1] for s in shapes])); nout = int(np.prod([s[0] for s in shapes]))
    ash = (nin,) if nrhs == 0 else (nin, nrhs)
    ysh = (nout,) if nrhs == 0 else (nout, nrhs)
    a = rng.standard_normal(ash).astype(dt); y0 = rng.standard_normal(ysh).astype(dt)
    Kp = cg.kronecker(*[torch.from_numpy(Fm).cuda() for Fm in Fs])
    yd = torch.from_numpy(y0.copy()).cuda()
    if beta == 0.0:
        yd.fill_(float("nan"))               # beta == 0: previous contents are not read (src/gramian.jl:80)
    cg.mul_(yd, Kp, torch.from_numpy(a).cuda(), alpha, beta)
    cols = [a] if nrhs == 0 else [a[:, c] for c in range(nrhs)]
    ycols = [y0] if nrhs == 0 else [y0[:, c] for c in range(nrhs)]
    ref = np.stack([oracle.kron_mul(yc, Fs, ac, alpha, beta) for ac, yc in zip(cols, ycols)], axis=-1)
    if nrhs == 0:
        ref = ref[:, 0]
    got = yd.cpu().numpy()
    assert np.isfinite(got).all(), (shapes, dt.__name__)
    # fp32: sums of up to prod(cols) products in fp32 fma chains against the fp64 oracle
    e = relerr(got, ref)
    tol = TOL[dt] * max(1.0, np.sqrt(max(s[1] for s in shapes) / 16.0))
    assert e <= tol, (shapes, dt.__name__, alpha, beta, nrhs, e)
    return e


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_kron_pair_kernel_shapes(cg, oracle, dt):
    rng = np.random.default_rng(11)
    cases = [
        [(128, 128)] * 3,                      # README.md:205-210: one pair pass + one mode pass
        [(64, 64)] * 3, [(32, 32)] * 3, [(16, 16)] * 4,
        [(20, 24), (36, 40), (52, 56)],        # NB1 = 4, ragged strips
        [(7, 5), (130, 100), (96, 112)],       # M1 = 130: 9 strips, NB1 = 8 with 7 live blocks
        [(9, 11), (33, 17), (200, 31)],        # unaligned slab rows (K2 = 31), N2 = 200: two output chunks
        [(3, 4), (48, 128), (300, 128)],       # N2 = 300: three chunks
        [(40, 40), (100, 90)],                 # q = 2 with a leading batch of one: the separate-modes route
        [(500, 3), (17, 19), (23, 29)],        # the pair shrinks nothing, pre = 500 * ...
        [(2, 300), (120, 16), (16, 120)],      # the pair first (it shrinks the tensor)
    ]
    for shapes in cases:
        run_case(cg, oracle, shapes, dt, rng)
        run_case(cg, oracle, shapes, dt, rng, alpha=-0.7, beta=1.3)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_kron_single_mode_kernels(cg, oracle, dt):
    rng = np.random.default_rng(12)
    cases = [
        [(37, 41)], [(200, 513)], [(1, 1)], [(16, 4)],                 # q = 1: the k-contiguous kernel alone
        [(30, 200), (50, 150)],                                         # c_q = 150 > 128: mode + modet
        [(129, 129), (129, 129)], [(5, 7), (3, 260)], [(64, 64), (256, 256)],
        [(6, 5), (4, 3), (2, 7), (3, 2), (5, 4)],                       # q = 5, tiny and odd everywhere
        [(1, 9), (9, 1), (4, 4)],
        [(256, 256), (256, 256)],
    ]
    for shapes in cases:
        run_case(cg, oracle, shapes, dt, rng)
        run_case(cg, oracle, shapes, dt, rng, alpha=0.5, beta=-2.0)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_kron_matrix_right_hand_sides(cg, oracle, dt):
    """Y = (F1 x ... x Fq) A: the columns are one more leading tensor index (src/gramian.jl:89-99 shape of mul!)."""
    rng = np.random.default_rng(13)
    for shapes, p in (([(8, 8)] * 3, 3), ([(20, 24), (36, 40), (52, 56)], 2), ([(37, 41)], 5), ([(30, 200), (50, 150)], 4)):
        run_case(cg, oracle, shapes, dt, rng, nrhs=p)
        run_case(cg, oracle, shapes, dt, rng, alpha=1.5, beta=0.25, nrhs=p)


def test_kron_raw_abi_padded_leading_dimensions(cg, oracle):
    """Device and host pointers, padded lds / lda / ldy, through the C ABI itself."""
    rng = np.random.default_rng(14)
    f = cg._ffi; lib = f.lib()
    ctx = cg.get_ctx(torch.device("cuda", 0)).bind_stream()
    shapes = [(5, 6), (33, 20), (40, 70)]
    Fs = [rng.standard_normal(s) for s in shapes]
    nin = int(np.prod([s[1] for s in shapes])); nout = int(np.prod([s[0] for s in shapes]))
    p, lda, ldy = 3, nin + 5, nout + 2
    A = rng.standard_normal((p, lda)); Y = rng.standard_normal((p, ldy)); Y0 = Y.copy()
    lds = [s[0] + 3 for s in shapes]
    bufs = []
    for Fm, ld in zip(Fs, lds):
        b = np.full((Fm.shape[1], ld), np.nan); b[:, :Fm.shape[0]] = Fm.T
        bufs.append(b)
    P = lambda arr: arr.ctypes.data_as(C.c_void_p)
    rows = (C.c_int64 * 3)(*[s[0] for s in shapes]); cols = (C.c_int64 * 3)(*[s[1] for s in shapes]); ldarr = (C.c_int64 * 3)(*lds)
    ref = np.stack([oracle.kron_mul(Y0[c, :nout], Fs, A[c, :nin], 0.5, 2.0) for c in range(p)])
    # host pointers
    ptrs = (f._P * 3)(*[P(b) for b in bufs])
    f.check(lib.covgram_kron_mvm(ctx, ptrs, rows, cols, ldarr, 3, f.F64, P(A), lda, P(Y), ldy, p, 0.5, 2.0, f.HOST))
    assert relerr(Y[:, :nout], ref) <= 1e-13 and np.array_equal(Y[:, nout:], Y0[:, nout:])
    # device pointers
    dbufs = [torch.from_numpy(b).cuda() for b in bufs]
    Ad = torch.from_numpy(A).cuda(); Yd = torch.from_numpy(Y0.copy()).cuda()
    ptrs = (f._P * 3)(*[f._P(b.data_ptr()) for b in dbufs])
    f.check(lib.covgram_kron_mvm(ctx, ptrs, rows, cols, ldarr, 3, f.F64, f._P(Ad.data_ptr()), lda, f._P(Yd.data_ptr()), ldy, p, 0.5, 2.0, f.DEVICE))
    Yh = Yd.cpu().numpy()
    assert relerr(Yh[:, :nout], ref) <= 1e-13 and np.array_equal(Yh[:, nout:], Y0[:, nout:])
    # argument errors are reported, not executed
    assert lib.covgram_kron_mvm(ctx, ptrs, rows, cols, ldarr, 3, f.F64, f._P(Ad.data_ptr()), nin - 1, f._P(Yd.data_ptr()), ldy, p, 1.0, 0.0, f.DEVICE) == f.EINVAL
    assert lib.covgram_kron_mvm(ctx, ptrs, rows, cols, ldarr, 0, f.F64, f._P(Ad.data_ptr()), lda, f._P(Yd.data_ptr()), ldy, p, 1.0, 0.0, f.DEVICE) == f.EINVAL


def test_kron_large_factor_takes_the_library_gemm(cg, oracle):
    """A factor side >= 1024 is a compute-bound dense GEMM: rocBLAS for that mode, the hand-written kernels for the others."""
    rng = np.random.default_rng(15)
    for dt in (np.float64, np.float32):
        run_case(cg, oracle, [(1024, 1030), (24, 40)], dt, rng, alpha=2.0, beta=-1.0)
        run_case(cg, oracle, [(12, 10), (1100, 1024)], dt, rng)


def test_kron_paths_by_shape(cg, oracle):
    """Which kernels a shape runs (info key "last_kron_path", round 4 — csrc/kron.hip sent shapes its kernels refuse to rocBLAS silently):
    bit 1 = fused last-two-modes pass, 2 = single-mode kernel, 4 = last-mode kernel, 8 = rocBLAS, 16 = two small trailing factors multiplied
    out.  The README case and everything GP-sized stays on the hand-written kernels; only compute-bound modes reach the library."""
    rng = np.random.default_rng(16)
    expect = [([(128, 128)] * 3, 1 | 2), ([(64, 64)] * 3, 2 | 4),          # 64 slabs are too few for the fused pass: mode by mode ([(32, 32)] * 4, None), ([(16, 16)] * 5, None), ([(40, 40), (100, 90)], None),
              ([(300, 200)], 4), ([(1024, 1030), (24, 40)], None), ([(256, 256)] * 2, None)]
    for shapes, want in expect:
        run_case(cg, oracle, shapes, np.float64, rng)
        path = cg.get_info("last_kron_path")
        assert path != 0
        big = any(max(sh) >= 1024 for sh in shapes)
        assert bool(path & 8) == big, (shapes, path)            # rocBLAS exactly for the factor sides >= 1024 among these shapes
        if want is not None:
            assert path == want, (shapes, path, want)


def test_kron_lazy_grid_full_size_properties(cg, oracle):
    """README.md:205-210's case (128^3 grid, three 128 x 128 Gramians) at full size: linearity and the transpose identity
    <u, K v> = <K^T u, v> (size-independent), plus 64 entries of K a against explicit rows of the Kronecker matrix."""
    rng = np.random.default_rng(16)
    ax = torch.linspace(0, 1, 128, dtype=torch.float64, device="cuda")
    G = cg.gramian(cg.separable("*", cg.Exp(), cg.EQ(), cg.MaternP(2)), cg.LazyGrid(ax, ax, ax))
    N = 128 ** 3
    u = torch.from_numpy(rng.standard_normal(N)).cuda(); v = torch.from_numpy(rng.standard_normal(N)).cuda()
    Gu, Gv = G @ u, G @ v
    assert relerr((G @ (2.0 * u - 3.0 * v)).cpu().numpy(), (2.0 * Gu - 3.0 * Gv).cpu().numpy()) <= 1e-13
    lhs = float(torch.dot(u, Gv)); rhs = float(torch.dot(Gu, v))      # the factors are symmetric Gramians
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), 1.0)
    xs = np.linspace(0, 1, 128)
    F = [oracle.matrix(oracle.Kernel(fam, **kw), xs, xs) for fam, kw in ((oracle.EXP, {}), (oracle.EQ, {}), (oracle.MATERNP, {"p": 2}))]
    vh = v.cpu().numpy().reshape(128, 128, 128)
    got = Gv.cpu().numpy()
    for idx in rng.integers(0, N, 64):
        i, j, k = idx // 16384, (idx // 128) % 128, idx % 128
        ref = np.einsum("a,b,c,abc->", F[0][i], F[1][j], F[2][k], vh)
        assert abs(got[idx] - ref) <= 1e-12 * max(abs(ref), 1.0)
