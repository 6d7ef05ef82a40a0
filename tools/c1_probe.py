import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
n = 4096
rng = np.random.default_rng(0xC0F)
X = torch.from_numpy(rng.standard_normal((n, 3))).cuda(); a = torch.from_numpy(rng.standard_normal(n)).cuda(); y = torch.empty_like(a)
G = cg.gramian(cg.MaternP(2), X)
for _ in range(20): G.mul_(y, a)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): G.mul_(y, a)
e1.record(); e1.synchronize()
print(f"C1: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per MVM back-to-back")
