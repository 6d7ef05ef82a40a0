"""Time composite-kernel (Sum/Product) dense MVMs against their single-profile parts (C2 shape).  usage: composite_bench.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg

def timeit(fn, warm=2, reps=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

n, d = 131072, 3
rng = np.random.default_rng(1)
X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
y = torch.empty(n, dtype=torch.float32, device="cuda")
M2, EQl, RQ, CA = cg.Lengthscale(cg.MaternP(2), 0.7), cg.Lengthscale(cg.EQ(), 2.0), cg.RQ(1.5), cg.Cauchy()
for name, k in (("EQ", cg.EQ()), ("MaternP(2;l)", M2), ("RQ(1.5)", RQ), ("Cauchy", CA),
                ("1.5*MaternP2 + 0.5*EQ", 1.5 * M2 + 0.5 * EQl), ("EQ*Cauchy", cg.EQ() * CA), ("EQ + 0.1", cg.EQ() + 0.1),
                ("MaternP2*RQ^2 + EQ + 0.5", M2 * RQ ** 2 + EQl + 0.5)):
    G = cg.gramian(k, X)
    ms = timeit(lambda: G.mul_(y, a))
    print(f"{name:28s} fp32 n={n} d={d}: {ms:8.3f} ms  {n * n / ms * 1e-6:8.1f} Gpairs/s", flush=True)
