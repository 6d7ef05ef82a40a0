import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
dt = torch.float32
nl, r = 1 << 20, 32
xs = torch.randn(nl, dtype=dt, device="cuda")
Gl = cg.gramian(cg.FiniteBasis([lambda t, i=i: torch.cos(0.37 * i * t) for i in range(r)]), xs)
al = torch.randn(nl, dtype=dt, device="cuda"); yl = torch.empty_like(al)
for _ in range(10): Gl.mul_(yl, al)
torch.cuda.synchronize()
