"""Dense EQ fp32 at GP-sized n: time per MVM against the column split and the LDS-sharing switch."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
cg.set_option("mfma_sym", 0)
for d in (3, 8):
    for n in (8192, 16384, 32768):
        rng = np.random.default_rng(n)
        X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
        y = torch.empty_like(a); G = cg.gramian(cg.EQ(), X)
        res = {}
        for rep in range(3):
            for lds in (-1, 0, 1):
                for js in (0, 4, 8, 16, 32, 64):
                    cg.set_option("mfma_lds", lds); cg.set_option("jsplit", js)
                    for _ in range(10): G.mul_(y, a)
                    torch.cuda.synchronize(); e0.record()
                    for _ in range(100): G.mul_(y, a)
                    e1.record(); e1.synchronize(); res.setdefault((lds, js), []).append(e0.elapsed_time(e1) / 100 * 1e3)
        cg.set_option("mfma_lds", -1); cg.set_option("jsplit", 0)
        print(f"d={d} n={n}: " + "  ".join(f"{k}:{np.median(v):.1f}" for k, v in res.items()), flush=True)
