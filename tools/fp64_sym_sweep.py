"""fp64 gramian(k, x) * a: the symmetric direct-difference kernel (upper triangle once) against the all-entries kernel.  Dev tool."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg

def timeit(fn, warm=3, reps=7):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

kernels = [("EQ", cg.EQ()), ("MaternP(2)", cg.MaternP(2)), ("Exp", cg.Exp()), ("RQ(1.5)", cg.RQ(1.5)), ("Cauchy", cg.Cauchy())]
for d in (3, 8):
    for n in (2048, 4096, 6144, 8192, 16384, 32768, 65536):
        rng = np.random.default_rng(5)
        X = torch.from_numpy(rng.standard_normal((n, d)) * (0.3 if d == 8 else 1.0)).cuda()
        a = torch.from_numpy(rng.standard_normal(n)).cuda(); y = torch.empty_like(a); y2 = torch.empty_like(a)
        for name, k in kernels if n in (4096, 16384, 32768) else kernels[:2]:
            G = cg.gramian(k, X)
            cg.set_option("dense_sym", 0); t0 = timeit(lambda: G.mul_(y, a))
            cg.set_option("dense_sym", 1); t1 = timeit(lambda: G.mul_(y2, a))
            err = float((y - y2).norm() / y.norm())
            print(f"d={d} n={n:6d} {name:12s} all entries {t0*1e3:9.1f} us   symmetric {t1*1e3:9.1f} us   x{t0/t1:5.2f}   diff {err:.1e}", flush=True)
cg.set_option("dense_sym", -1)
