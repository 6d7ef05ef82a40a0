"""4-wave symmetric matrix-core kernels: four against eight column tiles per stage (option mfma_sym_st)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn, reps):
    ts = []
    for rep in range(5):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts)) * 1e3
for name, n, d, part, k in (("C2 symmetric EQ", 131072, 3, False, cg.EQ()), ("C3 sym partial EQ (rank 3 of 8)", 524288, 8, True, cg.EQ()), ("EQ n=32768 d=3", 32768, 3, False, cg.EQ()),
                          ("MaternP(2) d=3 n=131072", 131072, 3, False, cg.MaternP(2)), ("RQ(1.5) d=3 n=131072", 131072, 3, False, cg.RQ(1.5)), ("MaternP(2) d=3 n=32768", 32768, 3, False, cg.MaternP(2))):
    rng = np.random.default_rng(3 + d)
    X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    G = cg.gramian(k, X); y = torch.empty(n, dtype=torch.float32, device="cuda")
    fn = (lambda: G.sym_partial_(y, a, 3, 8)) if part else (lambda: G.mul_(y, a))
    out = []; res = {}
    for st in (4, 0, 4, 0):
        cg.set_option("mfma_sym_st", st)
        t = timeit(fn, 5 if n > 200000 else 10)
        res[st] = y.clone()
        out.append(f"st={st or 8}: {t:7.1f} us")
    print(f"{name}: " + " | ".join(out) + f" | rel diff {float((res[4] - res[0]).norm() / res[4].norm()):.1e}", flush=True)
cg.set_option("mfma_sym_st", 0)
