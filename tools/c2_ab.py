"""Interleaved A/B of the grid size for C2 on the matrix-core EQ path."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
n = 131072
rng = np.random.default_rng(0xC0F + 1)
X = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
G = cg.gramian(cg.EQ(), X); y = torch.empty(n, dtype=torch.float32, device="cuda")
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
res = {}
for rep in range(5):
    for tw in (0, 8192, 12288, 24576, 32768, 65536):
        cg.set_option("target_wgs", tw)
        for _ in range(2): G.mul_(y, a)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10): G.mul_(y, a)
        e1.record(); e1.synchronize()
        res.setdefault(tw, []).append(e0.elapsed_time(e1) / 10)
for tw, v in res.items():
    print(f"target_wgs={tw}: median {np.median(v):.3f} ms  min {np.min(v):.3f} ms")
