"""Generic matrix-core kernels: fp16 two-way split (option mfma_f16 = -1: inside its gate) against the bf16 three-way split (0) — us per MVM, MFMAs per tile
implied by d, and norm-wise / row-wise error against fp64 oracle rows.  gramian(k, x) (symmetric kernels) and a row shard (general kernel)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg, covgram_oracle as o, c_oracle
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=10):
    ts = []
    for rep in range(3):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts)) * 1e3
L = cg.Lengthscale
for n, d in ((131072, 3), (65536, 3), (65536, 5), (65536, 8), (65536, 12), (65536, 16), (65536, 24), (65536, 30)):
    rng = np.random.default_rng(40 + d)
    Xh = (rng.standard_normal((n, d)) * (0.9 if n > 65536 else 1.0 if d <= 5 else 0.7)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); y = torch.empty_like(a)
    rows = np.sort(np.random.default_rng(7).choice(n, 128, replace=False))
    Xr = Xh[rows].astype(np.float64); Xd = Xh.astype(np.float64); ad = ah.astype(np.float64)
    per = 8192
    ys = torch.empty(per, dtype=torch.float32, device="cuda")
    for name, k, ko in (("MaternP(2)", cg.MaternP(2), o.Kernel(o.MATERNP, p=2)), ("RQ(1.5)", cg.RQ(1.5), o.Kernel(o.RQ, param=1.5)), ("Cauchy", cg.Cauchy(), o.Kernel(o.CAUCHY)),
                        ("MaternP(1;l=2)", cg.Lengthscale(cg.MaternP(1), 2.0), o.Kernel(o.MATERNP, p=1, lengthscale=2.0)),
                        ("EQ(1.4)+.7RQ(.8;.9)+.2M1", L(cg.EQ(), 1.4) + 0.7 * L(cg.RQ(0.8), 0.9) + 0.2 * cg.MaternP(1),
                         [(1.0, o.Kernel(o.EQ, lengthscale=1.4)), (0.7, o.Kernel(o.RQ, param=0.8, lengthscale=0.9)), (0.2, o.Kernel(o.MATERNP, p=1))])):
        if isinstance(ko, list):
            ref = sum(c * c_oracle.mvm(kk, Xr, Xd, ad) for c, kk in ko); absref = sum(c * c_oracle.mvm(kk, Xr, Xd, np.abs(ad)) for c, kk in ko)
        else:
            ref = c_oracle.mvm(ko, Xr, Xd, ad); absref = c_oracle.mvm(ko, Xr, Xd, np.abs(ad))
        G = cg.gramian(k, X); Gs = cg.gramian(k, X[:per].contiguous(), X)
        out = []
        for f in (0, -1):                                     # (a first, discarded round: the second configuration of a pair otherwise measures ~8 % faster)
            cg.set_option("mfma_f16", f); timeit(lambda: G.mul_(y, a), 5)
        for f in (0, -1):
            cg.set_option("mfma_f16", f)
            t = timeit(lambda: G.mul_(y, a))
            used = cg.get_info("last_mfma_f16"); path = (cg.get_info("last_dense_path"), cg.get_info("last_mfma_sym"))
            got = y.cpu().numpy()[rows].astype(np.float64)
            ts = timeit(lambda: Gs.mul_(ys, a))
            out.append(f"f16={f:2d} (used {used}, path {path}): sym {t:7.1f} us  shard {ts:6.1f} us  err {np.linalg.norm(got - ref) / np.linalg.norm(ref):.1e} rowwise {np.max(np.abs(got - ref) / absref):.1e}")
        print(f"d={d} n={n} {name}: " + " | ".join(out), flush=True)
cg.set_option("mfma_f16", -1)
