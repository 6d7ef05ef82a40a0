"""Round 4: the dense kernels sum their own split-J partials — the LAST workgroup of a row block to arrive (ticket counter behind a
device-scope fence, csrc/pack.hpp: last_arrival) adds the block's slab rows in the separate reduce kernel's fixed order and applies
alpha / beta.  The results must be BIT-identical to the two-launch form (option "inkernel_reduce" = 0) whatever the arrival order, launch
after launch (the tickets return to zero by themselves), for the matrix-core EQ kernel (LDS-shared and one-wave instances) and the
lane-per-row kernel (fp32 / fp64, one and several right-hand sides, alpha / beta, ragged sizes)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(cg, fn):
    cg.set_option("inkernel_reduce", 0)
    y0 = fn()
    assert cg.get_info("last_inkernel_reduce") == 0
    cg.set_option("inkernel_reduce", 1)
    ys = [fn() for _ in range(3)]                       # repeated launches: the tickets must have gone back to zero
    used = cg.get_info("last_inkernel_reduce")
    return y0, ys, used


@pytest.mark.parametrize("n,m,d", [(4096, 4096, 3), (30001, 30001, 3), (16384, 131072, 3), (5000, 70000, 8), (1234, 40000, 20)])
def test_matrix_core_eq_kernel_sums_its_own_slab(cg, oracle, n, m, d):
    rng = np.random.default_rng(n + m + d)
    Y = torch.from_numpy((rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)).cuda()
    X = Y[:n]
    a = torch.from_numpy(rng.standard_normal(m).astype(np.float32)).cuda()
    try:
        cg.set_option("mfma_sym", 0)
        G = cg.gramian(cg.EQ(), X, Y)
        for alpha, beta in ((1.0, 0.0), (-0.7, 1.3)):
            ystart = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
            def fn():
                y = ystart.clone(); G.mul_(y, a, alpha, beta); return y
            y0, ys, used = _both(cg, fn)
            assert cg.get_info("last_dense_path") == 2
            assert used == 1, "the column split should be > 1 at these sizes"
            for y in ys:
                assert torch.equal(y, y0)
        rows = np.sort(rng.choice(n, min(n, 256), replace=False))
        import c_oracle
        ref = c_oracle.mvm(oracle.Kernel(oracle.EQ), X[rows].double().cpu().numpy(), Y.double().cpu().numpy(), a.double().cpu().numpy())
        y = torch.empty(n, dtype=torch.float32, device="cuda"); G.mul_(y, a)
        assert np.linalg.norm(y.cpu().numpy()[rows] - ref) / np.linalg.norm(ref) <= 1e-5
    finally:
        cg.set_option("mfma_sym", -1); cg.set_option("inkernel_reduce", -1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n,m,d,p", [(4096, 4096, 3, 1), (1000, 9000, 5, 1), (777, 5000, 2, 3), (4100, 4100, 40, 1), (300, 20000, 8, 6)])
def test_lane_per_row_kernel_sums_its_own_slab(cg, oracle, dtype, n, m, d, p):
    o = oracle
    rng = np.random.default_rng(n + m + d + p)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    Xh = rng.standard_normal((n, d)).astype(npdt); Yh = rng.standard_normal((m, d)).astype(npdt)
    Ah = rng.standard_normal((m, p) if p > 1 else m).astype(npdt)
    X = torch.from_numpy(Xh).cuda(); Y = torch.from_numpy(Yh).cuda(); A = torch.from_numpy(Ah).cuda()
    try:
        cg.set_option("dense_variant", 1); cg.set_option("dense_sym", 0)
        for k, ko in ((cg.MaternP(2), o.Kernel(o.MATERNP, p=2)), (cg.Exp(), o.Kernel(o.EXP))):
            G = cg.gramian(k, X, Y)
            for alpha, beta in ((1.0, 0.0), (2.5, -0.5)):
                ystart = torch.from_numpy(rng.standard_normal((n, p) if p > 1 else n).astype(npdt)).cuda()
                def fn():
                    y = ystart.clone(); G.mul_(y, A, alpha, beta); return y
                y0, ys, used = _both(cg, fn)
                assert cg.get_info("last_dense_path") == 1
                assert used == (1 if cg.get_info("last_jsplit") > 1 else 0)
                for y in ys:
                    assert torch.equal(y, y0)
            y = torch.empty((n, p) if p > 1 else (n,), dtype=dtype, device="cuda"); G.mul_(y, A)
            ref = o.mul(None, ko, Xh.astype(np.float64), Yh.astype(np.float64), Ah.astype(np.float64))
            assert np.linalg.norm(y.double().cpu().numpy() - ref) / np.linalg.norm(ref) <= (1e-5 if dtype == torch.float32 else 1e-12)
    finally:
        cg.set_option("dense_variant", 0); cg.set_option("dense_sym", -1); cg.set_option("inkernel_reduce", -1)


def test_column_weights_formed_in_the_kernel_are_bit_identical(cg):
    """Option mfma_fuse_w: the general matrix-core EQ kernel forms w_j = a_j exp2(f_j) where it fetches a tile's weights (the default) or reads
    them from the pack launch's buffer — the same fp32 product either way, so the results are bit-identical; ragged sizes, both splits, the in-kernel reduce on top."""
    import numpy as np
    rng = np.random.default_rng(5)
    try:
        for n, m, d in ((1000, 777, 3), (4099, 5000, 8), (300, 20001, 2), (2500, 2500, 12)):
            X = torch.from_numpy((rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)).cuda()
            Y = torch.from_numpy((rng.standard_normal((m, d)) / np.sqrt(d)).astype(np.float32)).cuda()
            a = torch.from_numpy(rng.standard_normal(m).astype(np.float32)).cuda()
            G = cg.gramian(cg.EQ(), X, Y)
            outs = []
            for f16 in (1, 0):
                for fw in (0, 1, -1):
                    cg.set_option("mfma_f16", f16); cg.set_option("mfma_fuse_w", fw); cg.set_option("dense_variant", 2)
                    outs.append((f16, (G @ a).clone()))
                    assert cg.get_info("last_dense_path") == 2
            for f16 in (1, 0):
                group = [o for f, o in outs if f == f16]
                assert all(torch.equal(group[0], o) for o in group[1:]), (n, m, d, f16)
    finally:
        cg.set_option("mfma_f16", -1); cg.set_option("mfma_fuse_w", -1); cg.set_option("dense_variant", 0)
