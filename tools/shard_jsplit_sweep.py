"""Row shard of the contract workload (rank 0 of P): us per MVM by the column split ("jsplit" option; 0 = the library's choice)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
n, d = 131072, 3
rng = np.random.default_rng(20240607)
X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
cg.set_option("mfma_sym", 0)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def t(G, y):
    ts = []
    for rep in range(5):
        for _ in range(10): G.mul_(y, a)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): G.mul_(y, a)
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    return float(np.median(ts))
G1 = cg.gramian(cg.EQ(), X); y1 = torch.empty(n, dtype=torch.float32, device="cuda")
base = t(G1, y1)
print(f"P=1: {base:.1f} us")
for P in (8, 16):
    per = n // P
    G = cg.gramian(cg.EQ(), X[:per], X); y = torch.empty(per, dtype=torch.float32, device="cuda")
    for js in (0, 4, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64):
        cg.set_option("jsplit", js)
        us = t(G, y)
        print(f"P={P} jsplit={js}: {us:.1f} us ({base / P / us:.3f} of ideal)", flush=True)
    cg.set_option("jsplit", 0)
