"""fp32 gramian(k, x) * a where the matrix cores do not apply: the symmetric direct-difference kernel (dense_sym32_kernel, upper triangle
once) against the all-entries kernel (dense_mvm_kernel).  Dev tool; output kept as profiles/r04_sym32_sweep.txt."""
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

kernels = [("Exp", cg.Exp()), ("GammaExp(1.5)", cg.GammaExp(1.5)), ("MaternP(0)", cg.MaternP(0)), ("EQ l=0.05", cg.Lengthscale(cg.EQ(), 0.05)),
           ("MaternP(2) l=.05", cg.Lengthscale(cg.MaternP(2), 0.05)), ("Cauchy l=.02", cg.Lengthscale(cg.Cauchy(), 0.02))]
sizes = [int(s) for s in sys.argv[1:]] or [8192, 16384, 32768, 65536, 131072]
for d in (3, 8, 16):
    for n in sizes:
        rng = np.random.default_rng(5)
        X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda()
        a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda(); y = torch.empty_like(a); y2 = torch.empty_like(a)
        for name, k in kernels if (n in (32768, 131072) and d == 3) else kernels[:2]:
            G = cg.gramian(k, X)
            cg.set_option("dense_sym", 0); t0 = timeit(lambda: G.mul_(y, a)); p0 = cg.get_info("last_dense_path")
            cg.set_option("dense_sym", 1); t1 = timeit(lambda: G.mul_(y2, a)); s1 = cg.get_info("last_dense_sym")
            err = float((y - y2).norm() / y.norm())
            print(f"d={d:2d} n={n:6d} {name:16s} all entries (path {p0}) {t0*1e3:9.1f} us   symmetric (used {s1}) {t1*1e3:9.1f} us   x{t0/t1:5.2f}   diff {err:.1e}   jsplit {cg.get_info('last_jsplit')}", flush=True)
cg.set_option("dense_sym", -1)
