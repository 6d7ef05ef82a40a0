"""One named case, a few calls (for rocprofv3 --kernel-trace --stats): python tools/trace_case.py <c1|c5|c5f32|kron128|dotfac|c4|readme16k> [reps]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
case = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(0)
if case == "c1":
    X = torch.from_numpy(rng.standard_normal((4096, 3))).cuda(); a = torch.from_numpy(rng.standard_normal(4096)).cuda(); G = cg.gramian(cg.MaternP(2), X)
elif case == "readme16k":
    X = torch.from_numpy(rng.standard_normal((16384, 3))).cuda(); a = torch.from_numpy(rng.standard_normal(16384)).cuda(); G = cg.gramian(cg.MaternP(2), X)
elif case in ("c5", "c5f32"):
    dt = torch.float64 if case == "c5" else torch.float32      # a RANGE (cg.srange), not a tensor of its values: only a range is a Toeplitz Gramian (src/gramian.jl:167-189)
    G = cg.gramian(cg.Exp(), cg.srange(-1, 1, 1 << 22)); a = torch.randn(1 << 22, dtype=dt, device="cuda")
elif case == "kron128":
    ax = torch.linspace(0, 1, 128, dtype=torch.float64, device="cuda"); G = cg.gramian(cg.separable("*", cg.Exp(), cg.Exp(), cg.Exp()), cg.LazyGrid(ax, 3))
    a = torch.randn(128 ** 3, dtype=torch.float64, device="cuda")
elif case == "dotfac":
    X = torch.from_numpy(rng.standard_normal((1 << 20, 8)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(1 << 20).astype(np.float32)).cuda(); G = cg.gramian(cg.Dot(), X)
elif case == "c4":
    X = torch.from_numpy(rng.standard_normal((16384, 32))).cuda(); a = torch.from_numpy(rng.standard_normal(16384 * 32)).cuda(); G = cg.gramian(cg.GradientKernel(cg.EQ()), X)
y = torch.empty_like(a)
for _ in range(reps): G.mul_(y, a)
torch.cuda.synchronize()
