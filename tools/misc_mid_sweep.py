"""GP-sized n: lane-per-row dense kernel (profiles that stay off the matrix cores; fp64) and fp32 gradient kernel against the column split."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def sweep(tag, K, a, y):
    res = {}
    for rep in range(3):
        for js in (0, 4, 8, 16, 32, 64, 128, 256):
            cg.set_option("jsplit", js)
            for _ in range(10): K.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(40): K.mul_(y, a)
            e1.record(); e1.synchronize(); res.setdefault(js, []).append(e0.elapsed_time(e1) / 40 * 1e3)
    cg.set_option("jsplit", 0)
    print(tag + ": " + "  ".join(f"{k}:{np.median(v):.1f}" for k, v in res.items()), flush=True)
for (kern, dt, d) in ((cg.Exp(), torch.float32, 3), (cg.Exp(), torch.float32, 8), (cg.EQ(), torch.float64, 3), (cg.MaternP(2), torch.float64, 8)):
    for n in (8192, 16384, 32768):
        rng = np.random.default_rng(n)
        X = torch.from_numpy(rng.standard_normal((n, d))).to(dt).cuda(); a = torch.from_numpy(rng.standard_normal(n)).to(dt).cuda(); y = torch.empty_like(a)
        sweep(f"dense {type(kern).__name__[:6]} {str(dt)[6:]} d={d} n={n}", cg.gramian(kern, X), a, y)
for (kern, d) in ((cg.EQ(), 8), (cg.EQ(), 32), (cg.MaternP(2), 16)):
    for n in (4096, 8192, 16384):
        rng = np.random.default_rng(n)
        X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n * d).astype(np.float32)).cuda(); y = torch.empty_like(a)
        sweep(f"grad32 {type(kern).__name__[:6]} d={d} n={n}", cg.gramian(cg.GradientKernel(kern), X), a, y)
