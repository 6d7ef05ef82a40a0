"""The fp16 two-way split of the coordinates in the GENERIC fp32 matrix-core kernels (round 5; csrc/dense_mfma.hpp: gen_row_fragments,
csrc/dense_mfma.hip: mfma_gen_fmt) — MaternP, RQ, Cauchy, IMQ, EQ^p and the one-pass Sum evaluate k(x, y) (reference: src/gramian.jl:63-75 calling
the kernels of src/stationary.jl) from |x|^2 + |y|^2 - 2 x.y accumulated on the matrix cores; inside a tighter radius gate the coordinates
ride as two fp16 pieces (3 products each, four coordinates per MFMA) instead of three bf16 pieces (6 products, two per MFMA).  Checked against
the fp64 oracle on an ADVERSARIAL cloud placed AT the gate (half the points exactly on the gate's sphere: isolated rows whose own diagonal entry
dominates, where the absolute error of |x|^2 + |y|^2 - 2 x.y is largest), in every d class of the compiled instances, on the symmetric and the
general kernels, and against the bf16 split on the same inputs (option "mfma_f16" = 0):
  * BASELINE.json's bar, 1e-5 NORM-wise, holds at the gates (measured <= 7.3e-6 anywhere inside them: profiles/r05_gate_scan.txt);
  * ROW-wise (|err_i| / (|K| |a|)_i) this cloud reaches 1.5e-5 (bf16 split) / 2.5e-5 (fp16 split) at the edge of the gates, at d >= 5 — of the
    EQ kernels too; it scales with the gate, and option "mfma_gate_pct" = 40 brings every split under 1e-5 row-wise on it (asserted below).
    Gaussian clouds (the BASELINE configs, tests/test_gpu_fullsize.py) sit at 1e-8 .. 1e-6 row-wise."""
ROW_AT_GATE = 3.0e-5
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LIMIT = 0.5 * 126.0 / 1.4426950408889634074          # csrc/dense_mfma.hip: mfma_gen_eligible's bound on sensitivity x power x R^2 / l^2


def relerr(b, ref):
    b = np.asarray(b, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return np.linalg.norm(b - ref) / np.linalg.norm(ref)


def rowwise(b, ref, absref):
    return float(np.max(np.abs(np.asarray(b, dtype=np.float64) - ref) / absref))


def _profiles(cg, o):
    """(name, kernel, oracle kernel, sensitivity x power / l^2 of the radius gate)"""
    L = cg.Lengthscale
    m2 = 0.5  # MaternP: max(|phi'(0)/phi(0)|, 0.5) in s; p = 1: 1/2 ... the library's own bound is what the gate uses — 0.5 is its floor
    return [
        ("MaternP(2)", cg.MaternP(2), o.Kernel(o.MATERNP, p=2), None),
        ("MaternP(1;l=2)", L(cg.MaternP(1), 2.0), o.Kernel(o.MATERNP, p=1, lengthscale=2.0), None),
        ("RQ(1.5)", cg.RQ(1.5), o.Kernel(o.RQ, param=1.5), 0.5),
        ("Cauchy(l=1.5)", L(cg.Cauchy(), 1.5), o.Kernel(o.CAUCHY, lengthscale=1.5), 1.0 / 2.25),
        ("IMQ(1.2)", cg.InverseMultiQuadratic(1.2), o.Kernel(o.IMQ, param=1.2), 0.5 / 1.44),
        ("RQ(0.7;l=0.8)", L(cg.RQ(0.7), 0.8), o.Kernel(o.RQ, param=0.7, lengthscale=0.8), 0.5 / 0.64),
    ]


def _cloud(rng, n, d, rho):
    """n points (n even) in adjacent +- pairs — the library's centre (the mean of an evenly spaced sample, here an even-length prefix) is the origin —,
    half of them ON the sphere of radius rho, the rest inside it"""
    h = n // 2
    v = rng.standard_normal((h, d)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    r = np.where(np.arange(h) % 2 == 0, 1.0, rng.uniform(0.0, 1.0, h) ** (1.0 / d))
    P = v * (rho * 0.9995 * r)[:, None]
    P = P.astype(np.float32)
    X = np.empty((n, d), dtype=np.float32); X[0::2] = P; X[1::2] = -P
    return X


@pytest.mark.parametrize("d", [1, 3, 5, 6, 8, 10, 12, 14, 16, 22, 30])
def test_fp16_split_at_its_gate_symmetric_and_general(cg, oracle, d):
    o = oracle
    n = 1536
    rng = np.random.default_rng(5100 + d)
    ah = rng.standard_normal(n).astype(np.float32); a = torch.from_numpy(ah).cuda(); ad = ah.astype(np.float64)
    try:
        for name, k, ko, c in _profiles(cg, o):
            if c is None: continue                      # MaternP: its sensitivity bound is the library's; covered by the radius scan below
            for frac, want in ((0.999 * 72.0 / 126.0, 1), (0.62, 0), (0.995, 0)):
                rho = np.sqrt(frac * LIMIT / c)
                Xh = _cloud(rng, n, d, rho); X = torch.from_numpy(Xh).cuda(); Xd = Xh.astype(np.float64)
                ref = o.mul(None, ko, Xd, Xd, ad); absref = np.abs(o.matrix(ko, Xd, Xd)) @ np.abs(ad)
                for sym in (1, 0):
                    cg.set_option("mfma_sym", sym); cg.set_option("mfma_f16", -1)
                    G = cg.gramian(k, X)
                    y = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda"); G.mul_(y, a)
                    assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_sym") == sym, (name, d, frac)
                    assert cg.get_info("last_mfma_f16") == want, (name, d, frac, sym)
                    b = y.cpu().numpy()
                    assert np.isfinite(b).all()
                    assert relerr(b, ref) <= 1e-5 and rowwise(b, ref, absref) <= ROW_AT_GATE, (name, d, frac, sym, relerr(b, ref), rowwise(b, ref, absref))
                    if want:
                        cg.set_option("mfma_f16", 0)
                        y0 = torch.empty_like(y); G.mul_(y0, a)
                        assert cg.get_info("last_mfma_f16") == 0
                        assert rowwise(y0.cpu().numpy(), ref, absref) <= ROW_AT_GATE, (name, d)
                        assert rowwise(b, y0.cpu().numpy().astype(np.float64), absref) <= ROW_AT_GATE, (name, d, sym)
                        # the scaled gates: this cloud now lies outside the fp16 split's, and a cloud AT the scaled gates holds 1e-5 row-wise
                        cg.set_option("mfma_f16", -1); cg.set_option("mfma_gate_pct", 40)
                        G.mul_(y0, a)
                        assert cg.get_info("last_mfma_f16") == 0, (name, d)
                        for fr2, want2 in ((0.39, 0), (0.4 * 0.999 * 72.0 / 126.0, 1)):
                            X2h = _cloud(rng, n, d, np.sqrt(fr2 * LIMIT / c)); X2d = X2h.astype(np.float64)
                            y2 = torch.empty_like(y); cg.gramian(k, torch.from_numpy(X2h).cuda()).mul_(y2, a)
                            assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_f16") == want2, (name, d, fr2)
                            r2 = o.mul(None, ko, X2d, X2d, ad); ar2 = np.abs(o.matrix(ko, X2d, X2d)) @ np.abs(ad)
                            assert rowwise(y2.cpu().numpy(), r2, ar2) <= 1e-5, (name, d, fr2, sym, rowwise(y2.cpu().numpy(), r2, ar2))
                        cg.set_option("mfma_gate_pct", 100)
    finally:
        cg.set_option("mfma_sym", -1); cg.set_option("mfma_f16", -1); cg.set_option("mfma_gate_pct", 100)


@pytest.mark.parametrize("d", [2, 4, 7, 9, 13, 18, 27])
def test_fp16_split_radius_scan_maternp_and_sum(cg, oracle, d):
    """MaternP(p) and Sums: whatever split the library's gates pick at each radius (both occur over the scan), the result holds 1e-5 norm-wise and
    the row-wise bound of the module docstring; inside 40 % of the gates (option "mfma_gate_pct") 1e-5 row-wise"""
    o = oracle
    n, m = 768, 1280
    rng = np.random.default_rng(5200 + d)
    ah = rng.standard_normal(m).astype(np.float32); a = torch.from_numpy(ah).cuda(); ad = ah.astype(np.float64)
    L = cg.Lengthscale
    ks = [("MaternP(2)", cg.MaternP(2), [(1.0, o.Kernel(o.MATERNP, p=2))]),
          ("MaternP(3;l=1.4)", L(cg.MaternP(3), 1.4), [(1.0, o.Kernel(o.MATERNP, p=3, lengthscale=1.4))]),
          ("M1+RQ+EQ", 0.6 * cg.MaternP(1) + L(cg.RQ(1.2), 1.3) + 0.4 * L(cg.EQ(), 1.7),
           [(0.6, o.Kernel(o.MATERNP, p=1)), (1.0, o.Kernel(o.RQ, param=1.2, lengthscale=1.3)), (0.4, o.Kernel(o.EQ, lengthscale=1.7))])]
    seen = set()
    try:
      for pct, rowtol in ((100, ROW_AT_GATE), (40, 1e-5)):
        cg.set_option("mfma_gate_pct", pct)
        for name, k, terms in ks:
            for rho in (1.5, 3.0, 4.5, 5.5, 6.5, 8.0, 10.0):
                Yh = _cloud(rng, m, d, rho); Xh = _cloud(rng, n, d, 0.8 * rho)
                X = torch.from_numpy(Xh).cuda(); Y = torch.from_numpy(Yh).cuda(); Xd, Yd = Xh.astype(np.float64), Yh.astype(np.float64)
                ref = sum(c * o.mul(None, kk, Xd, Yd, ad) for c, kk in terms)
                absref = sum(abs(c) * (np.abs(o.matrix(kk, Xd, Yd)) @ np.abs(ad)) for c, kk in terms)
                G = cg.gramian(k, X, Y)
                y = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda"); G.mul_(y, a)
                if cg.get_info("last_dense_path") != 2: continue          # beyond the matrix-core gate: direct differences
                f = cg.get_info("last_mfma_f16"); seen.add((name, f))
                b = y.cpu().numpy()
                assert relerr(b, ref) <= 1e-5 and rowwise(b, ref, absref) <= rowtol, (name, d, rho, f, pct, relerr(b, ref), rowwise(b, ref, absref))
                # the same on gramian(k, y)
                cg.set_option("mfma_sym", 1)
                Gs = cg.gramian(k, Y); ys = torch.empty(m, dtype=torch.float32, device="cuda"); Gs.mul_(ys, a)
                cg.set_option("mfma_sym", -1)
                if cg.get_info("last_dense_path") != 2: continue
                refs = sum(c * o.mul(None, kk, Yd, Yd, ad) for c, kk in terms)
                absrefs = sum(abs(c) * (np.abs(o.matrix(kk, Yd, Yd)) @ np.abs(ad)) for c, kk in terms)
                assert rowwise(ys.cpu().numpy(), refs, absrefs) <= rowtol, (name, d, rho, pct, cg.get_info("last_mfma_f16"))
      for name, _, _ in ks:
          assert (name, 1) in seen and (name, 0) in seen, (d, sorted(seen))
    finally:
        cg.set_option("mfma_sym", -1); cg.set_option("mfma_gate_pct", 100)


def test_fp16_split_refused_where_it_has_no_instance(cg):
    """d + 2 positions must fit eight MFMAs of four (d <= 30); beyond that, and for the dot-product kernels, the bf16 split serves"""
    rng = np.random.default_rng(77)
    for d, want in ((14, 1), (15, 1), (24, 1), (30, 1), (31, 0)):
        X = torch.from_numpy((0.3 * rng.standard_normal((1024, d))).astype(np.float32)).cuda(); a = torch.ones(1024, dtype=torch.float32, device="cuda")
        y = torch.empty_like(a); cg.gramian(cg.RQ(1.0), X[:512].contiguous(), X).mul_(y[:512], a)
        assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_f16") == want, d
