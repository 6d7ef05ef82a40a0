"""One fp64 gramian(k, x) * a configuration (MaternP(2), d = 8, n = 32768) on the all-entries and on the symmetric kernel, 10 calls each:
the target of `rocprofv3 --kernel-trace --stats -- python3 tools/fp64_sym_profile.py`.  Dev tool."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
n, d = 32768, 8
rng = np.random.default_rng(5)
X = torch.from_numpy(rng.standard_normal((n, d)) * 0.3).cuda()
a = torch.from_numpy(rng.standard_normal(n)).cuda(); y = torch.empty_like(a)
G = cg.gramian(cg.MaternP(2), X)
for mode in (0, 1):
    cg.set_option("dense_sym", mode)
    for _ in range(10): G.mul_(y, a)
    torch.cuda.synchronize()
cg.set_option("dense_sym", -1)
