"""Per-kernel view of the small-factor Kronecker cases (16^5, 32^4): run under rocprofv3 --kernel-trace and read with tools/rocpd_stats.py."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
side, dims, dt = int(sys.argv[1]), int(sys.argv[2]), (torch.float32 if sys.argv[3] == "f32" else torch.float64)
ax = torch.linspace(0, 1, side, dtype=dt, device="cuda")
G = cg.gramian(cg.separable("*", *([cg.Exp()] * dims)), cg.LazyGrid(ax, dims))
a = torch.randn(side ** dims, dtype=dt, device="cuda"); y = torch.empty_like(a)
for _ in range(20): G.mul_(y, a)
torch.cuda.synchronize()
