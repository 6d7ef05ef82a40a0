"""C5 (Exponential Toeplitz, n = 2^22) fp64 / fp32 and the neighbouring sizes: us per MVM back to back, by toeplitz_persist (0 / 1)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for lg in (22, 20, 21, 23):
    n = 1 << lg
    for dt in (torch.float64, torch.float32):
        T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n, dtype=dt))
        a = torch.randn(n, dtype=T.dtype, device="cuda"); y = torch.empty_like(a)
        res = {}
        for rep in range(5):
            for v in (0, 1):
                cg.set_option("toeplitz_persist", v)
                for _ in range(5): T.mul_(y, a)
                torch.cuda.synchronize(); e0.record()
                for _ in range(50): T.mul_(y, a)
                e1.record(); e1.synchronize(); res.setdefault(v, []).append(e0.elapsed_time(e1) / 50 * 1e3)
        print(f"n=2^{lg} {T.dtype}: " + "  ".join(f"persist={k}: median {np.median(v):.1f} us min {min(v):.1f}" for k, v in res.items()), flush=True)
cg.set_option("toeplitz_persist", -1)
