"""fp32 gramian(k, x) * a on the direct-difference SYMMETRIC kernel (csrc/dense_sym32.hpp, round 4): the profiles the matrix-core path
refuses — Exponential (src/stationary.jl:60), gamma-exponential (:71), MaternP(0) — and clouds that fail its radius gate evaluate the
upper triangle once instead of all n^2 entries (the reference's mul! follows the data's element type for every kernel,
src/gramian.jl:27-33, 78-87).  Against the fp64 oracle norm-wise AND row-wise at BASELINE.json's 1e-5; the single-launch form, the
partial (multi-GPU) form, ragged sizes, every rows-per-lane instance (d <= 8: 4, d <= 32: 2, beyond: 1)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return np.linalg.norm(b - ref) / np.linalg.norm(ref)


def rowwise(b, ref, absref):
    return float(np.max(np.abs(np.asarray(b, dtype=np.float64) - ref) / absref))


def _cases(cg, o):
    return [("Exp", cg.Exp(), o.Kernel(o.EXP)), ("GammaExp1.5", cg.GammaExp(1.5), o.Kernel(o.GAMMAEXP, param=1.5)),
            ("MaternP0_l", cg.Lengthscale(cg.MaternP(0), 0.7), o.Kernel(o.MATERNP, p=0, lengthscale=0.7)),
            ("EQ_narrow", cg.Lengthscale(cg.EQ(), 0.05), o.Kernel(o.EQ, lengthscale=0.05)),      # fails the matrix-core radius gate
            ("RQ_narrow", cg.Lengthscale(cg.RQ(1.5), 0.02), o.Kernel(o.RQ, param=1.5, lengthscale=0.02)),
            ("Dot3", cg.Dot() ** 3, None)]


@pytest.mark.parametrize("n,d", [(1000, 3), (4099, 1), (2600, 8), (1537, 12), (900, 40), (8, 2), (513, 5)])
def test_fp32_symmetric_direct_kernel_matches_the_oracle(cg, oracle, n, d):
    o = oracle
    rng = np.random.default_rng(400 + n + d)
    Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    Xd, ad = Xh.astype(np.float64), ah.astype(np.float64)
    try:
        for name, k, ko in _cases(cg, o):
            if ko is None:
                continue
            G = cg.gramian(k, X)
            cg.set_option("dense_sym", 1)                       # below the automatic size too
            y = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
            G.mul_(y, a)
            used = cg.get_info("last_dense_sym") == 1 and cg.get_info("last_dense_path") == 1
            cg.set_option("dense_sym", 0)
            y0 = torch.empty_like(y); G.mul_(y0, a)
            assert cg.get_info("last_dense_sym") == 0
            if not used:       # a cloud inside the matrix-core gate (EQ_narrow at small d can be): nothing to compare
                assert cg.get_info("last_dense_path") == 2, name
                continue
            ref = o.mul(None, ko, Xd, Xd, ad)
            absref = np.abs(o.matrix(ko, Xd, Xd)) @ np.abs(ad)
            b = y.cpu().numpy()
            assert np.isfinite(b).all()
            assert relerr(b, ref) <= 1e-5 and rowwise(b, ref, absref) <= 1e-5, (name, n, d, relerr(b, ref), rowwise(b, ref, absref))
            # ... and it is the same arithmetic as the all-entries kernel up to the order of the sums
            assert rowwise(y0.cpu().numpy(), ref, absref) <= 1e-5
            # alpha / beta through the same path (beta = 0 above clobbered the NaN fill)
            cg.set_option("dense_sym", 1)
            y2 = torch.from_numpy(ah[::-1].copy()).cuda()
            G.mul_(y2, a, -0.7, 1.3)
            assert relerr(y2.cpu().numpy(), -0.7 * ref + 1.3 * ad[::-1]) <= 1e-5, name
    finally:
        cg.set_option("dense_sym", -1)


def test_fp32_symmetric_direct_kernel_is_the_default_from_24576(cg, oracle):
    """Automatic choice: Exponential at n = 25000 takes it (n = 20000 does not: break-even is at n ~ 22000), a 4-column right-hand side, two point sets, a Power wrapper and a composite do
    not; 512 oracle rows."""
    o = oracle
    n, d = 25000, 3
    rng = np.random.default_rng(77)
    Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    cg.gramian(cg.Exp(), X[:20000].contiguous()) @ a[:20000].contiguous()
    assert cg.get_info("last_dense_sym") == 0
    G = cg.gramian(cg.Exp(), X)
    b = (G @ a).cpu().numpy()
    assert cg.get_info("last_dense_sym") == 1 and cg.get_info("last_dense_path") == 1
    rows = np.sort(rng.choice(n, 512, replace=False))
    import c_oracle
    ref = c_oracle.mvm(o.Kernel(o.EXP), Xh[rows].astype(np.float64), Xh.astype(np.float64), ah.astype(np.float64))
    assert relerr(b[rows], ref) <= 1e-5
    A4 = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32)).cuda()
    G @ A4
    assert cg.get_info("last_dense_sym") == 0
    cg.gramian(cg.Exp(), X, X.clone()) @ a
    assert cg.get_info("last_dense_sym") == 0
    cg.gramian(cg.Exp() ** 2, X) @ a
    assert cg.get_info("last_dense_sym") == 0
    cg.gramian(cg.Exp() * cg.EQ(), X) @ a
    assert cg.get_info("last_dense_sym") == 0


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_fp32_direct_partials_add_up(cg, oracle, world):
    """covgram_mvm_sym_partial in fp32 for what the matrix cores refuse: rank r of P evaluates the 64 R-row blocks r, r + P, ... and the
    partials add up to G a (ONE all-reduce in covgram.dist); supported() answers for the world size it is asked about."""
    o = oracle
    rng = np.random.default_rng(190 + world)
    for n, d, k, ko in ((3000, 3, cg.Exp(), o.Kernel(o.EXP)), (4133, 5, cg.GammaExp(1.2), o.Kernel(o.GAMMAEXP, param=1.2)),
                        (1500, 12, cg.Lengthscale(cg.MaternP(0), 2.0), o.Kernel(o.MATERNP, p=0, lengthscale=2.0))):
        Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
        X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
        G = cg.gramian(k, X)
        assert G.sym_partial_supported(world)
        total = torch.zeros(n, dtype=torch.float64, device="cuda")
        part = torch.empty(n, dtype=torch.float32, device="cuda")
        for r in range(world):
            part.fill_(float("nan"))
            G.sym_partial_(part, a, r, world)
            assert cg.get_info("last_dense_sym") == 1 and torch.isfinite(part).all()
            total += part.double()
        ref = o.mul(None, ko, Xh.astype(np.float64), Xh.astype(np.float64), ah.astype(np.float64))
        assert relerr(total.cpu().numpy(), ref) <= 1e-5, (n, d, world)
    assert not cg.gramian(cg.Exp() ** 2, X).sym_partial_supported(world)          # Power wrapper: refused, not computed some other way


def test_sym_supported_agrees_with_sym_partial_at_the_slab_cap(cg):
    """supported() ignored the world-dependent column-sum slab cap that the partial call enforces, so for fp64 point sets
    above n ~ 131072 sqrt(world) every ShardedGramian.matmul raised EUNSUPPORTED after supported() had said yes.  Now ONE predicate serves
    both: around the 2 GiB cap (fp64: ceil(n / 64 / world) * n * 8 bytes) the two agree for every world size — nothing is launched where
    supported() says no, and the call is served where it says yes."""
    n, d = 150016, 2                       # fp64: world 1 -> 2344 blocks x 150016 x 8 B = 2.8 GB (no); world 2 -> 1.4 GB (yes)
    X = torch.randn(n, d, dtype=torch.float64, device="cuda"); a = torch.randn(n, dtype=torch.float64, device="cuda")
    G = cg.gramian(cg.Cauchy(), X)
    part = torch.empty(n, dtype=torch.float64, device="cuda")
    answers = {}
    for world in (1, 2, 4):
        ok = G.sym_partial_supported(world)
        answers[world] = ok
        if ok:
            G.sym_partial_(part, a, world - 1, world)           # the last rank: the fewest blocks, quick
            assert cg.get_info("last_dense_sym") == 1 and torch.isfinite(part).all()
        else:
            with pytest.raises(Exception) as ei:
                G.sym_partial_(part, a, world - 1, world)
            assert "symmetric" in str(ei.value)
    assert answers == {1: False, 2: True, 4: True}, answers
    # ShardedGramian asks with ITS world size: at world 1 (forced collective off) it never takes the partial form, and the plain MVM of the
    # same Gramian still works above the cap (row path)
    S = cg.ShardedGramian(cg.Cauchy(), X[:20000].contiguous())
    assert S.sym_partial is None
