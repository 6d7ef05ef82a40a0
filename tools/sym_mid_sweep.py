"""Symmetric kernels at GP-sized n: us per MVM against the number of column chunks (option jsplit -> tchunk = ntile / jsplit)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
cg.set_option("mfma_sym", 1)
for (kern, d) in ((cg.EQ(), 3), (cg.EQ(), 8), (cg.MaternP(2), 3)):
    for n in (12000, 16384, 24000, 32768, 50000):
        rng = np.random.default_rng(n)
        X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
        y = torch.empty_like(a); G = cg.gramian(kern, X)
        res = {}
        for rep in range(3):
            for js in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24):
                cg.set_option("jsplit", js)
                for _ in range(10): G.mul_(y, a)
                torch.cuda.synchronize(); e0.record()
                for _ in range(50): G.mul_(y, a)
                e1.record(); e1.synchronize(); res.setdefault(js, []).append(e0.elapsed_time(e1) / 50 * 1e3)
        cg.set_option("jsplit", 0)
        print(f"{type(kern).__name__[:6]} d={d} n={n} (ntile {(n + 31) // 32}): " + "  ".join(f"{k}:{np.median(v):.1f}" for k, v in res.items()), flush=True)
cg.set_option("mfma_sym", -1)
