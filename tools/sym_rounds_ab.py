"""Symmetric matrix-core kernels: rounds of resident workgroups the work list aims at (option target_wgs = 768 x rounds; automatic: 5 for the two-row-tile kernels since this measurement, 8 before and for the others), interleaved; us per MVM."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
CASES = ((131072, 3, cg.EQ()), (131072, 8, cg.EQ()), (65536, 3, cg.EQ()), (131072, 3, cg.MaternP(2)), (262144, 3, cg.EQ()))
if len(sys.argv) > 1: CASES = ((131072, 8, cg.MaternP(2)), (65536, 8, cg.MaternP(2)), (131072, 12, cg.EQ()), (131072, 16, cg.EQ()), (65536, 12, cg.Cauchy()), (131072, 24, cg.EQ()))
for n, d, kern in CASES:
    rng = np.random.default_rng(1)
    X = torch.from_numpy((rng.standard_normal((n, d)) * min(1.0, 2.0 / np.sqrt(d))).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    G = cg.gramian(kern, X); y = torch.empty(n, dtype=torch.float32, device="cuda")
    res = {}
    for rep in range(3):
        for k in (0, 6, 5, 4, 10):
            cg.set_option("target_wgs", 768 * k)
            for _ in range(4): G.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): G.mul_(y, a)
            e1.record(); e1.synchronize()
            if rep: res.setdefault(k, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    cg.set_option("target_wgs", 0)
    print(f"n={n} d={d} {type(kern).__name__[:7]}: " + "  ".join(f"rounds={k or 'auto'}: {min(v):.0f}" for k, v in res.items()), flush=True)
