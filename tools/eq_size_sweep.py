"""Dense EQ fp32 MVM across sizes (general kernel: all n*m entries; symmetric default): pairs per second per size — looks for cliffs."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for d in (3, 8):
    for n in (4096, 8192, 16384, 20000, 32768, 50000, 65536, 100000, 131072, 200000, 262144):
        rng = np.random.default_rng(n)
        X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
        y = torch.empty_like(a)
        out = []
        for sym in (0, -1):
            cg.set_option("mfma_sym", sym)
            G = cg.gramian(cg.EQ(), X)
            for _ in range(5): G.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            reps = max(3, int(2e10 / (n * n)))
            for _ in range(reps): G.mul_(y, a)
            e1.record(); e1.synchronize(); ms = e0.elapsed_time(e1) / reps
            out.append(f"{'all' if sym == 0 else 'default'} {ms:.4f} ms = {n * n / ms * 1e-9:.2f} Tpairs/s (path {cg.get_info('last_dense_path')}, sym {cg.get_info('last_mfma_sym')})")
        print(f"d={d} n={n}: " + " | ".join(out), flush=True)
cg.set_option("mfma_sym", -1)
