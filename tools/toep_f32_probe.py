import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
n = 1 << 22
T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n, torch.float32)); a = torch.randn(n, dtype=torch.float32, device="cuda"); y = torch.empty_like(a)
for _ in range(20): T.mul_(y, a)
torch.cuda.synchronize()
