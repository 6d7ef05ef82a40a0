import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import covgram as cg, covgram_oracle as o
for dt in (torch.float64, torch.float32):
    for n in (100000, 500000, 2000000, 6000000):
        x = cg.srange(-1, 1, n, dt)
        G = cg.gramian(cg.Exp(), x)
        rng = np.random.default_rng(n)
        a = rng.standard_normal(n).astype(np.float64 if dt == torch.float64 else np.float32)
        y = (G @ torch.from_numpy(a).cuda()).cpu().numpy()
        vc, _ = o.toeplitz_vectors(o.Kernel(o.EXP), o.srange(-1, 1, n))
        ref = o.toeplitz_mul(None, vc, None, a.astype(np.float64))
        N = 1
        while N < 2 * n - 1: N *= 2
        print(dt, n, "Mp =", N // 2 // 1024, "rel err", float(np.linalg.norm(y - ref) / np.linalg.norm(ref)), flush=True)
