"""C3's symmetric partial (rank 3 of 8) and C2's: rounds of resident workgroups the work list aims at (option target_wgs = 768 x rounds; automatic: 4), interleaved."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for n, d in ((524288, 8), (131072, 3)):
    rng = np.random.default_rng(3 + d)
    X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    G = cg.gramian(cg.EQ(), X); y = torch.empty(n, dtype=torch.float32, device="cuda")
    res = {}
    for rep in range(3):
        for k in (0, 2, 3, 5, 6):
            cg.set_option("target_wgs", 768 * k)
            for _ in range(3): G.sym_partial_(y, a, 3, 8)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): G.sym_partial_(y, a, 3, 8)
            e1.record(); e1.synchronize()
            if rep: res.setdefault(k, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    cg.set_option("target_wgs", 0)
    print(f"n={n} d={d} partial 3 of 8: " + "  ".join(f"rounds={k or 'auto(4)'}: {min(v):.0f}" for k, v in res.items()), flush=True)
