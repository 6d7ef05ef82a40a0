"""Kernel-level look at the panel (wide) gradient path: usage widegrad_probe.py n d [force_wide]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
n, d = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3: cg.set_option("grad_keep_r", 2)
X = torch.randn(n, d, dtype=torch.float64, device="cuda"); a = torch.randn(n * d, dtype=torch.float64, device="cuda"); y = torch.empty_like(a)
K = cg.gramian(cg.GradientKernel(cg.EQ()), X)
for _ in range(3): K.mul_(y, a)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); K.mul_(y, a); e1.record(); e1.synchronize()
print(f"n={n} d={d}: {e0.elapsed_time(e1):.3f} ms")
