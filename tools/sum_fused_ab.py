"""Round 5: the packed-profile matrix-core kernels and the one-pass Sum on the contract-size cloud (d = 3, n = 131072, fp32, x ~ N(0, I)) and neighbours.
us per MVM (median of 5 batches), the path taken, and the error against 128 fp64 oracle rows."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg
import covgram_oracle as o
import c_oracle
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)

def timeit(fn, reps=6):
    ts = []
    for rep in range(5):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts)) * 1e3

def rel(b, ref): return float(np.linalg.norm(b - ref) / np.linalg.norm(ref))

L = cg.Lengthscale
for d in (3, 8):
    n = 131072
    rng = np.random.default_rng(0xC0F + 1 if d == 3 else 11)
    Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); y = torch.empty_like(a)
    rows = np.sort(np.random.default_rng(7).choice(n, 128, replace=False))
    Xr = Xh[rows].astype(np.float64); Xd = Xh.astype(np.float64); ad = ah.astype(np.float64)
    singles = [("MaternP(2)", cg.MaternP(2), o.Kernel(o.MATERNP, p=2)), ("MaternP(1)", cg.MaternP(1), o.Kernel(o.MATERNP, p=1)), ("MaternP(2;l=0.7)", L(cg.MaternP(2), 0.7), o.Kernel(o.MATERNP, p=2, lengthscale=0.7)),
               ("RQ(1.5)", cg.RQ(1.5), o.Kernel(o.RQ, param=1.5)), ("Cauchy", cg.Cauchy(), o.Kernel(o.CAUCHY)), ("EQ", cg.EQ(), o.Kernel(o.EQ))]
    for name, k, ko in singles:
        G = cg.gramian(k, X)
        t = timeit(lambda: G.mul_(y, a))
        info = (cg.get_info("last_dense_path"), cg.get_info("last_mfma_sym"), cg.get_info("last_dense_sym"))
        ref = c_oracle.mvm(ko, Xr, Xd, ad)
        err = rel(y.cpu().numpy()[rows].astype(np.float64), ref)
        line = f"d={d} n={n} gramian({name}, x): default (path, mfma_sym, dense_sym)={info} {t:8.1f} us  err {err:.1e}"
        if name.startswith("MaternP"):
            cg.set_option("dense_variant", 1); t1 = timeit(lambda: G.mul_(y, a)); cg.set_option("dense_variant", 0)
            line += f" | direct differences {t1:8.1f} us"
        print(line, flush=True)
    # the GP model of bench.py's F2 line
    kc = 1.5 * L(cg.MaternP(2), 0.7) + 0.5 * L(cg.EQ(), 2.0)
    G = cg.gramian(kc, X)
    ref = 1.5 * c_oracle.mvm(o.Kernel(o.MATERNP, p=2, lengthscale=0.7), Xr, Xd, ad) + 0.5 * c_oracle.mvm(o.Kernel(o.EQ, lengthscale=2.0), Xr, Xd, ad)
    for sf in (-1, 0):
        cg.set_option("sum_fused", sf)
        t = timeit(lambda: G.mul_(y, a))
        print(f"d={d} n={n} 1.5 MaternP(2; 0.7) + 0.5 EQ(2): sum_fused={sf:2d} (fused={cg.get_info('last_sum_fused')}) {t:8.1f} us  err {rel(y.cpu().numpy()[rows].astype(np.float64), ref):.1e}", flush=True)
    cg.set_option("sum_fused", -1)
    k3 = L(cg.EQ(), 1.4) + 0.7 * L(cg.RQ(0.8), 0.9) + 0.2 * cg.MaternP(1)
    G = cg.gramian(k3, X)
    for sf in (-1, 0):
        cg.set_option("sum_fused", sf)
        t = timeit(lambda: G.mul_(y, a))
        print(f"d={d} n={n} EQ(1.4) + 0.7 RQ(0.8; 0.9) + 0.2 MaternP(1): sum_fused={sf:2d} (fused={cg.get_info('last_sum_fused')}) {t:8.1f} us", flush=True)
    cg.set_option("sum_fused", -1)
    # a row shard (two point sets): 16384 x 131072
    per = 16384
    Gs = cg.gramian(cg.MaternP(2), X[:per].contiguous(), X); ys = torch.empty(per, dtype=torch.float32, device="cuda")
    for dv in (0, 1, 2):
        cg.set_option("dense_variant", dv)
        t = timeit(lambda: Gs.mul_(ys, a), reps=10)
        print(f"d={d} shard {per} x {n} MaternP(2): dense_variant={dv} path {cg.get_info('last_dense_path')} {t:8.1f} us", flush=True)
    cg.set_option("dense_variant", 0)
    Gs = cg.gramian(kc, X[:per].contiguous(), X)
    for sf in (-1, 0):
        cg.set_option("sum_fused", sf)
        t = timeit(lambda: Gs.mul_(ys, a), reps=10)
        print(f"d={d} shard {per} x {n} 1.5 MaternP(2; 0.7) + 0.5 EQ(2): sum_fused={sf:2d} (fused={cg.get_info('last_sum_fused')}, path {cg.get_info('last_dense_path')}) {t:8.1f} us", flush=True)
    cg.set_option("sum_fused", -1)
    del X, a, y
