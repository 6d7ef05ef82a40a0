"""The kernels of ONE row shard's MVM (contract workload as rank 0 of P): run under rocprofv3 --kernel-trace, read with tools/rocpd_stats.py.  usage: shard_kernels.py P"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
P = int(sys.argv[1]); n, d = 131072, 3
rng = np.random.default_rng(20240607)
X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
cg.set_option("mfma_sym", 0)
per = n // P
G = cg.gramian(cg.EQ(), X[:per], X); y = torch.empty(per, dtype=torch.float32, device="cuda")
for _ in range(60): G.mul_(y, a)
torch.cuda.synchronize()
