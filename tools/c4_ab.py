"""Interleaved A/B of the split-J grid size for C4 (GradientKernel(EQ), d=32, n=16384, fp64)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
n, d = 16384, 32
X = torch.randn(n, d, dtype=torch.float64, device="cuda"); a = torch.randn(n * d, dtype=torch.float64, device="cuda"); y = torch.empty_like(a)
K = cg.gramian(cg.GradientKernel(cg.EQ()), X)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
res = {}
for rep in range(5):
    for tw in (0, 12288, 16384, 24576, 32768):
        cg.set_option("target_wgs", tw)
        for _ in range(2): K.mul_(y, a)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10): K.mul_(y, a)
        e1.record(); e1.synchronize()
        res.setdefault(tw, []).append(e0.elapsed_time(e1) / 10)
for tw, v in res.items():
    print(f"target_wgs={tw}: median {np.median(v):.3f} ms  min {np.min(v):.3f} ms")
