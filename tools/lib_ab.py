"""One build of the library (COVGRAM_LIB, default the in-tree one) on the contract shapes: run once per build, alternating, on one box (tools/lib_ab.sh)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
out = []
for name, n, per, d, reps in (("C2 general", 131072, 131071, 3, 10), ("C2 symmetric", 131072, 0, 3, 10), ("C3 shard", 524288, 65536, 8, 5), ("C3 sym partial", 524288, -1, 8, 5),
                             ("16384 x 131072 d=3", 131072, 16384, 3, 20), ("d=16 n=65536", 65536, 65535, 16, 10), ("d=8 l=0.8 (bf16 split)", 131072, 131071, 8, 5)):
    rng = np.random.default_rng(3 + d)
    X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    k = cg.Lengthscale(cg.EQ(), 0.8) if "l=0.8" in name else cg.EQ()
    if per > 0:
        G = cg.gramian(k, X[:per].contiguous(), X); y = torch.empty(per, dtype=torch.float32, device="cuda"); fn = lambda: G.mul_(y, a)
    elif per == 0:
        G = cg.gramian(k, X); y = torch.empty(n, dtype=torch.float32, device="cuda"); fn = lambda: G.mul_(y, a)
    else:
        G = cg.gramian(k, X); y = torch.empty(n, dtype=torch.float32, device="cuda"); fn = lambda: G.sym_partial_(y, a, 3, 8)
    ts = []
    for rep in range(5):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    out.append(f"{name}: {np.median(ts) * 1e3:.1f}")
print(" | ".join(out), flush=True)
