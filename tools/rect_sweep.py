"""Rectangular Gramians (prediction shapes: few rows, many columns), fp32: us per MVM against the column split."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for (kern, d) in ((cg.EQ(), 3), (cg.EQ(), 8), (cg.MaternP(2), 3)):
    for (n, m) in ((256, 65536), (1024, 65536), (4096, 65536), (1024, 16384), (4096, 16384), (16384, 4096)):
        rng = np.random.default_rng(n + m)
        X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); Y = torch.from_numpy(rng.standard_normal((m, d)).astype(np.float32)).cuda()
        a = torch.from_numpy(rng.standard_normal(m).astype(np.float32)).cuda(); y = torch.empty(n, dtype=torch.float32, device="cuda")
        G = cg.gramian(kern, X, Y)
        res = {}
        for rep in range(3):
            for js in (0, 4, 8, 16, 32, 64, 128, 256):
                cg.set_option("jsplit", js)
                for _ in range(10): G.mul_(y, a)
                torch.cuda.synchronize(); e0.record()
                for _ in range(50): G.mul_(y, a)
                e1.record(); e1.synchronize(); res.setdefault(js, []).append(e0.elapsed_time(e1) / 50 * 1e3)
        cg.set_option("jsplit", 0)
        print(f"{type(kern).__name__[:6]} d={d} {n}x{m} (path {cg.get_info('last_dense_path')}): " + "  ".join(f"{k}:{np.median(v):.1f}" for k, v in res.items()), flush=True)
