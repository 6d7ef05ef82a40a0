"""C1 (MaternP(2), d = 3, n = 4096, fp64): time per MVM against the column split."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for n in (4096, 8192, 2048):
    rng = np.random.default_rng(0xC0F)
    X = torch.from_numpy(rng.standard_normal((n, 3))).cuda(); a = torch.from_numpy(rng.standard_normal(n)).cuda()
    G = cg.gramian(cg.MaternP(2), X); y = torch.empty_like(a)
    res = {}
    for rep in range(3):
        for js in (0, 4, 8, 16, 32, 64, 128):
            cg.set_option("jsplit", js)
            for _ in range(10): G.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(100): G.mul_(y, a)
            e1.record(); e1.synchronize(); res.setdefault(js, []).append(e0.elapsed_time(e1) / 100 * 1e3)
    cg.set_option("jsplit", 0)
    print(f"n={n}: " + "  ".join(f"{js}:{np.median(v):.1f}us" for js, v in res.items()), flush=True)
