"""Contract workload (EQ, d = 3, n = 131072, fp32, all n*m entries) as rank 0 of P sees it: the row shard's MVM (every kernel of
the call: weight pack, matrix-core kernel, split-J reduction) timed back to back on one GPU, against 1/P of the P = 1 time."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
n, d = 131072, 3
rng = np.random.default_rng(20240607)
X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
cg.set_option("mfma_sym", 0)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
base = None
for P in (1, 2, 4, 8, 16):
    per = n // P
    G = cg.gramian(cg.EQ(), X[:per], X); y = torch.empty(per, dtype=torch.float32, device="cuda")
    ts = []
    for rep in range(5):
        for _ in range(10): G.mul_(y, a)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): G.mul_(y, a)
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    t = float(np.median(ts))
    if base is None: base = t
    print(f"P={P}: rows {per}: {t:.1f} us per MVM  (ideal {base / P:.1f} us, {base / P / t:.3f} of it)", flush=True)
