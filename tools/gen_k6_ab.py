"""Generic symmetric kernels at six MFMAs per tile (Cauchy / IMQ / EQ^2: d = 20 with the fp16 split, d = 11 with bf16): us per MVM (compare across builds: the one-tile-per-stage
4-wave kernel before round 5's change, the staged 8-wave kernel after)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg, covgram_oracle as o, c_oracle
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=10):
    ts = []
    for rep in range(3):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts)) * 1e3
n = 65536
for d, f16 in ((20, -1), (18, -1), (11, 0), (10, 0), (8, -1)):
    rng = np.random.default_rng(d)
    Xh = (rng.standard_normal((n, d)) / np.sqrt(d) * 1.5).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); y = torch.empty_like(a)
    rows = np.sort(np.random.default_rng(7).choice(n, 128, replace=False)); Xr = Xh[rows].astype(np.float64); Xd = Xh.astype(np.float64); ad = ah.astype(np.float64)
    cg.set_option("mfma_f16", f16)
    for name, k, ko in (("Cauchy", cg.Cauchy(), o.Kernel(o.CAUCHY)), ("IMQ(1.2)", cg.InverseMultiQuadratic(1.2), o.Kernel(o.IMQ, param=1.2)), ("EQ^2", cg.EQ() ** 2, o.Kernel(o.EQ, power=2)),
                        ("RQ(1.5)", cg.RQ(1.5), o.Kernel(o.RQ, param=1.5))):
        G = cg.gramian(k, X)
        t = timeit(lambda: G.mul_(y, a))
        ref = c_oracle.mvm(ko, Xr, Xd, ad); got = y.cpu().numpy()[rows].astype(np.float64)
        print(f"n={n} d={d} {name}: {t:8.1f} us (f16 {cg.get_info('last_mfma_f16')}, path {cg.get_info('last_dense_path')}/{cg.get_info('last_mfma_sym')}) err {np.linalg.norm(got - ref) / np.linalg.norm(ref):.1e}", flush=True)
cg.set_option("mfma_f16", -1)
