import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import covgram as cg, covgram_oracle as o
n = int(sys.argv[1]); js = int(sys.argv[2])
rng = np.random.default_rng(1)
Xh = rng.standard_normal((n, 3)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
G = cg.gramian(cg.EQ(), X); y = torch.empty(n, dtype=torch.float32, device="cuda")
cg.set_option("mfma_sym", 1); cg.set_option("jsplit", js)
for _ in range(10): G.mul_(y, a)
torch.cuda.synchronize()
rows = np.arange(0, n, 997)
ref = o.mul(None, o.Kernel(o.EQ), Xh[rows], Xh, ah, dtype=np.float32)
print("rel-err", float(np.linalg.norm(y.cpu().numpy()[rows] - ref) / np.linalg.norm(ref)))
