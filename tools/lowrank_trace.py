"""Low-rank MVM U (U^T a), one shape, a few calls (for rocprofv3 --kernel-trace --stats): python tools/lowrank_trace.py <log2 n> <r> <f32|f64>"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
nl, r = 1 << int(sys.argv[1]), int(sys.argv[2]); dt = torch.float32 if sys.argv[3] == "f32" else torch.float64
xs = torch.randn(nl, dtype=dt, device="cuda")
G = cg.gramian(cg.FiniteBasis([lambda t, i=i: torch.cos(0.37 * i * t) for i in range(r)]), xs)
a = torch.randn(nl, dtype=dt, device="cuda"); y = torch.empty_like(a)
for _ in range(30): G.mul_(y, a)
torch.cuda.synchronize()
