"""Run K launches of one hot-path config (for rocprofv3 --pmc / --kernel-trace passes).  usage: pmc_run.py dense|grad|toeplitz [K]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
which = sys.argv[1]; K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(0xC0F + 1)
if which == "dense":
    n = 131072
    X = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    G = cg.gramian(cg.EQ(), X); y = torch.empty(n, dtype=torch.float32, device="cuda")
    for _ in range(K): G.mul_(y, a)
elif which == "grad":
    n, d = 16384, 32
    X = torch.from_numpy(rng.standard_normal((n, d))).cuda(); a = torch.from_numpy(rng.standard_normal(n * d)).cuda()
    G = cg.gramian(cg.GradientKernel(cg.EQ()), X); y = torch.empty(n * d, dtype=torch.float64, device="cuda")
    for _ in range(K): G.mul_(y, a)
elif which == "toeplitz":
    n = 1 << 22
    T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n)); a = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(a)
    for _ in range(K): T.mul_(y, a)
torch.cuda.synchronize()
print("done", which, K)
