"""C1 (MaternP(2), d = 3, n = 4096, fp64) and neighbours: the all-entries lane-per-row kernel against the direct-difference symmetric kernel (option dense_sym)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=50):
    ts = []
    for rep in range(5):
        for _ in range(5): fn()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts)) * 1e3
for n in (2048, 4096, 6144, 8192):
    for name, k in (("MaternP(2)", cg.MaternP(2)), ("EQ", cg.EQ()), ("RQ(1.5)", cg.RQ(1.5))):
        rng = np.random.default_rng(1)
        X = torch.from_numpy(rng.standard_normal((n, 3))).cuda(); a = torch.from_numpy(rng.standard_normal(n)).cuda(); y = torch.empty_like(a)
        G = cg.gramian(k, X)
        out = []
        for ds in (0, 1, 0, 1, -1):
            cg.set_option("dense_sym", ds)
            t = timeit(lambda: G.mul_(y, a))
            out.append(f"dense_sym={ds:2d} (ran {cg.get_info('last_dense_sym')}): {t:6.1f} us")
        print(f"n={n} fp64 d=3 {name}: " + " | ".join(out), flush=True)
cg.set_option("dense_sym", -1)
