"""One Kronecker shape, a few MVMs (for rocprofv3 --kernel-trace --stats): python tools/kron_one.py <side> <dims> <f32|f64> [reps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
side, dims = int(sys.argv[1]), int(sys.argv[2]); dt = torch.float32 if sys.argv[3] == "f32" else torch.float64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
ax = torch.linspace(0, 1, side, dtype=dt, device="cuda")
G = cg.gramian(cg.separable("*", *([cg.Exp()] * dims)), cg.LazyGrid(ax, dims))
a = torch.randn(side ** dims, dtype=dt, device="cuda"); y = torch.empty_like(a)
for _ in range(reps): G.mul_(y, a)
torch.cuda.synchronize()
print("path", cg.get_info("last_kron_path"))
