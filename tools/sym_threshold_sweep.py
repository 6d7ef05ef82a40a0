"""gramian(k, x), fp32, GP-sized n: all entries (mfma_sym = 0) against the symmetric kernels (mfma_sym = 1) — where the default's threshold belongs."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for (kern, d) in ((cg.EQ(), 3), (cg.EQ(), 8), (cg.MaternP(2), 3), (cg.RQ(1.5), 3), (cg.Cauchy(), 3), (cg.MaternP(2), 8)):
    line = []
    for n in (6144, 8192, 10000, 12000, 14000, 16384, 20000, 24000):
        rng = np.random.default_rng(n)
        X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
        y = torch.empty_like(a); G = cg.gramian(kern, X)
        res = {}
        for rep in range(3):
            for sym in (0, 1):
                cg.set_option("mfma_sym", sym)
                for _ in range(10): G.mul_(y, a)
                torch.cuda.synchronize(); e0.record()
                for _ in range(50): G.mul_(y, a)
                e1.record(); e1.synchronize(); res.setdefault(sym, []).append(e0.elapsed_time(e1) / 50 * 1e3)
        line.append(f"{n}: {np.median(res[0]):.1f} / {np.median(res[1]):.1f}")
    cg.set_option("mfma_sym", -1)
    print(f"{type(kern).__name__[:6]} d={d} (all / sym us): " + "  ".join(line), flush=True)
