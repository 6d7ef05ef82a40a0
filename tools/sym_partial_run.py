"""K launches of covgram_mvm_sym_partial(rank 3 of 8) at C2 size (for rocprofv3 passes).  usage: sym_partial_run.py [world] [K] [jsplit]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8; K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if len(sys.argv) > 3: cg.set_option("jsplit", int(sys.argv[3]))
n, d = 131072, 3
rng = np.random.default_rng(0xC0F + 1)
X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
G = cg.gramian(cg.EQ(), X); part = torch.empty(n, dtype=torch.float32, device="cuda")
for _ in range(K): G.sym_partial_(part, a, min(3, P - 1), P)
torch.cuda.synchronize(); print("done")
