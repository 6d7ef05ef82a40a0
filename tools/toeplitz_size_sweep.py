"""Toeplitz MVM (Exponential on a uniform grid) across sizes, fp64 and fp32: us per MVM and GB/s of the 112 n (fp64) ideal traffic."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for dt in (torch.float64, torch.float32):
    line = []
    for lg in range(10, 24):
        for n in ((1 << lg), (1 << lg) + (1 << (lg - 1))):
            if n > (1 << 23): continue
            T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n, dtype=dt)); a = torch.randn(n, dtype=dt, device="cuda"); y = torch.empty_like(a)
            for _ in range(5): T.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(30): T.mul_(y, a)
            e1.record(); e1.synchronize(); us = e0.elapsed_time(e1) / 30 * 1e3
            line.append(f"{n}:{us:.1f}us({(14 * n * (8 if dt == torch.float64 else 4)) / us * 1e-3:.0f}GB/s)")
            del T
    print(str(dt)[6:] + ": " + "  ".join(line), flush=True)
