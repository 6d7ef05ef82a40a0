"""One rank's row shard of the contract workload by column split (option jsplit): P = 8, 16 -> 16384 / 8192 rows x 131072 columns."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
n, d = 131072, 3
rng = np.random.default_rng(20240607)
X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
cg.set_option("mfma_sym", 0)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for P in (8, 16, 4):
    per = n // P
    G = cg.gramian(cg.EQ(), X[:per], X); y = torch.empty(per, dtype=torch.float32, device="cuda")
    out = []
    for js in (0, 16, 24, 32, 48, 64, 96, 128, 256):
        cg.set_option("jsplit", js)
        ts = []
        for rep in range(5):
            for _ in range(10): G.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): G.mul_(y, a)
            e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        out.append(f"{js}: {np.median(ts):6.1f}")
    print(f"P={P} rows {per}: us by jsplit (0 = auto)  " + " | ".join(out), flush=True)
cg.set_option("jsplit", 0)
