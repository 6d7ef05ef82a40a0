"""Kronecker (SeparableProduct on LazyGrid) and low-rank (FiniteBasis-like U V') MVMs across shapes: us per MVM and GB/s of the
compulsory traffic (Kronecker: 2 passes over the tensor per mode; low rank: U and V once)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
for dt in (torch.float64, torch.float32):
    for (side, dims) in ((32, 3), (64, 3), (128, 3), (256, 3), (256, 2), (1024, 2), (4096, 2), (16, 5), (32, 4)):
        ax = torch.linspace(0, 1, side, dtype=dt, device="cuda")
        G = cg.gramian(cg.separable("*", *([cg.Exp()] * dims)), cg.LazyGrid(ax, dims))
        N = side ** dims
        a = torch.randn(N, dtype=dt, device="cuda"); y = torch.empty_like(a)
        us = timeit(lambda: G.mul_(y, a))
        by = dims * 2 * N * (8 if dt == torch.float64 else 4)
        print(f"kron {str(dt)[6:]} {side}^{dims} (N = {N}): {us:.1f} us = {by / us * 1e-3:.0f} GB/s of {dims} x (read + write); {2.0 * N * side * dims / us * 1e-6:.1f} TFLOP/s", flush=True)
