"""Generic symmetric matrix-core kernels (MaternP, RQ, Cauchy, IMQ, one-pass Sum) at one or two MFMAs per tile: 4- or 8-wave panels of ONE row tile per wave
(dense_mfma_sym_kernel, option mfma_sym_rt = 1) against 4 waves x TWO row tiles (dense_mfma_sym2_kernel's generic form, 2 / -1), alternating; the two
results against each other and against fp64 oracle rows."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg, covgram_oracle as o, c_oracle
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn, reps):
    ts = []
    for rep in range(5):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts)) * 1e3
L = cg.Lengthscale
for n, d, scale in ((131072, 3, 1.0), (131072, 5, 0.6), (32768, 3, 1.0), (131072, 2, 1.0)):
    rng = np.random.default_rng(3 + d)
    Xh = (scale * rng.standard_normal((n, d))).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); y = torch.empty(n, dtype=torch.float32, device="cuda")
    rows = np.sort(np.random.default_rng(7).choice(n, 128, replace=False)); Xr = Xh[rows].astype(np.float64); Xd = Xh.astype(np.float64); ad = ah.astype(np.float64)
    for name, k, ko in (("MaternP(2)", cg.MaternP(2), [(1.0, o.Kernel(o.MATERNP, p=2))]), ("MaternP(1)", cg.MaternP(1), [(1.0, o.Kernel(o.MATERNP, p=1))]),
                        ("MaternP(3)", cg.MaternP(3), [(1.0, o.Kernel(o.MATERNP, p=3))]), ("RQ(1.5)", cg.RQ(1.5), [(1.0, o.Kernel(o.RQ, param=1.5))]),
                        ("Cauchy", cg.Cauchy(), [(1.0, o.Kernel(o.CAUCHY))]), ("IMQ(1.2)", cg.InverseMultiQuadratic(1.2), [(1.0, o.Kernel(o.IMQ, param=1.2))]),
                        ("EQ+.7RQ+.2M1", L(cg.EQ(), 1.4) + 0.7 * L(cg.RQ(0.8), 0.9) + 0.2 * cg.MaternP(1), [(1.0, o.Kernel(o.EQ, lengthscale=1.4)), (0.7, o.Kernel(o.RQ, param=0.8, lengthscale=0.9)), (0.2, o.Kernel(o.MATERNP, p=1))]),
                        ("1.5M2(.7)+.5EQ(2) fused", 1.5 * L(cg.MaternP(2), 0.7) + 0.5 * L(cg.EQ(), 2.0), [(1.5, o.Kernel(o.MATERNP, p=2, lengthscale=0.7)), (0.5, o.Kernel(o.EQ, lengthscale=2.0))])):
        if n < 131072 and name not in ("MaternP(2)", "Cauchy"): continue
        cg.set_option("mfma_sym", 1); cg.set_option("sum_fused", 1 if "fused" in name else -1)
        G = cg.gramian(k, X)
        out = []; res = {}
        for rt in (1, 2, 1, 2):
            cg.set_option("mfma_sym_rt", rt)
            t = timeit(lambda: G.mul_(y, a), 10)
            res[rt] = y.cpu().numpy().astype(np.float64)
            if rt in res and len(out) >= 2: out[rt - 1] = f"rt={rt} ({cg.get_info('last_mfma_sym_rt')}, f16={cg.get_info('last_mfma_f16')}, path {cg.get_info('last_dense_path')}/{cg.get_info('last_mfma_sym')}): {t:7.1f} us"
            else: out.append("")
        ref = sum(c * c_oracle.mvm(kk, Xr, Xd, ad) for c, kk in ko)
        errs = [np.linalg.norm(res[rt][rows] - ref) / np.linalg.norm(ref) for rt in (1, 2)]
        print(f"n={n} d={d} {name}: " + " | ".join(out) + f" | rel diff {np.linalg.norm(res[1] - res[2]) / np.linalg.norm(res[1]):.1e} | err vs oracle {errs[0]:.1e} / {errs[1]:.1e}", flush=True)
cg.set_option("mfma_sym_rt", -1); cg.set_option("mfma_sym", -1); cg.set_option("sum_fused", -1)
