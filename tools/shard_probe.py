"""One rank's share of the C2 MVM at N = 8 (16384 rows x 131072 columns), timed as a back-to-back loop: shows the fixed per-MVM
costs (pack, reduce, launch gaps) that bound strong scaling.  usage: shard_probe.py [rows]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = 131072
rng = np.random.default_rng(0xC0F + 1)
X = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
G = cg.gramian(cg.EQ(), X[:rows], X); y = torch.empty(rows, dtype=torch.float32, device="cuda")
for _ in range(10): G.mul_(y, a)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
K = 200
e0.record()
for _ in range(K): G.mul_(y, a)
e1.record(); e1.synchronize()
print(f"rows={rows}: {e0.elapsed_time(e1) / K * 1e3:.1f} us per MVM (ideal 1/8 of 1660 us = 207 us)")
res = {}
for rep in range(6):
    for tw in (0, 4096, 8192, 12288):
        cg.set_option("target_wgs", tw)
        for _ in range(3): G.mul_(y, a)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100): G.mul_(y, a)
        e1.record(); e1.synchronize()
        res.setdefault(tw, []).append(e0.elapsed_time(e1) / 100 * 1e3)
for tw, v in res.items():
    print(f"  target_wgs={tw}: median {np.median(v):.1f} us  min {np.min(v):.1f} us")
