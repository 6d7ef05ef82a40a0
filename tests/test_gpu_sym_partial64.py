"""fp64 form of covgram_mvm_sym_partial (the reference's default element type): rank r of P evaluates the 64-row blocks r, r + P, ... of the
upper triangle of gramian(k, x) on the direct-difference symmetric kernel and returns the partial product of those entries and their mirror
images; the partials of all ranks add up to G a (one all-reduce in covgram.dist).  Rows of src/gramian.jl:81 are independent, and so are
the unordered pairs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return np.linalg.norm(b - ref) / np.linalg.norm(ref)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_fp64_partials_add_up(cg, oracle, world):
    o = oracle
    rng = np.random.default_rng(90 + world)
    for n, d, k, ko in ((1000, 3, cg.MaternP(2), o.Kernel(o.MATERNP, p=2)), (4133, 5, cg.Lengthscale(cg.EQ(), 1.3), o.Kernel(o.EQ, lengthscale=1.3)),
                        (2500, 8, cg.RQ(2.0), o.Kernel(o.RQ, param=2.0)), (777, 2, cg.Dot() + 0.0 if False else cg.Cauchy(), o.Kernel(o.CAUCHY))):
        Xh = rng.standard_normal((n, d)); ah = rng.standard_normal(n)
        X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
        G = cg.gramian(k, X)
        assert G.sym_partial_supported()
        total = torch.zeros(n, dtype=torch.float64, device="cuda")
        part = torch.empty(n, dtype=torch.float64, device="cuda")
        norms = []
        for r in range(world):
            part.fill_(float("nan"))
            G.sym_partial_(part, a, r, world)
            assert cg.get_info("last_dense_sym") == 1
            assert torch.isfinite(part).all()
            norms.append(float(part.norm()))
            total += part
        ref = o.mul(None, ko, Xh, Xh, ah)
        assert relerr(total.cpu().numpy(), ref) <= 1e-12, (n, d, world)
        if world > 1 and n >= 64 * world:
            assert min(norms) > 0                      # every rank really contributed
    # what the partial form does not serve is refused, not computed some other way
    Gp = cg.gramian(cg.Cauchy() ** 2, X)
    assert not Gp.sym_partial_supported()
