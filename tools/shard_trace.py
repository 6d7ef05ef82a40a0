"""One rank's row shard of the contract workload (P = 8 / 16: 16384 / 8192 rows x 131072 columns, EQ, d = 3, fp32), K launches — for rocprofv3 --kernel-trace --stats. usage: shard_trace.py P [K] [jsplit]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
P = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n, d = 131072, 3
rng = np.random.default_rng(20240607)
X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
cg.set_option("mfma_sym", 0)
if len(sys.argv) > 3: cg.set_option("jsplit", int(sys.argv[3]))
per = n // P
G = cg.gramian(cg.EQ(), X[:per], X); y = torch.empty(per, dtype=torch.float32, device="cuda")
for _ in range(K): G.mul_(y, a)
torch.cuda.synchronize()
print("instance", cg.get_info("last_mfma_instance"), "lds", cg.get_info("last_mfma_lds"))
