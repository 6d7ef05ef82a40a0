"""fp32 EQ, d = 24 / 32 (six / eight MFMAs per tile, fp16 split), gramian(k, x): the one-tile-per-stage 4-wave kernel (dense_mfma_sym_wide_kernel) against the
staged 8-wave kernel at the same fragment length (option mfma_sym_st = 16), alternating."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=10):
    ts = []
    for rep in range(3):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts)) * 1e3
for n in (32768, 131072):
    for d in (20, 24, 28, 32):
        rng = np.random.default_rng(d)
        X = torch.from_numpy((rng.standard_normal((n, d)) / np.sqrt(d) * 2.0).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
        G = cg.gramian(cg.EQ(), X); y = torch.empty_like(a)
        out = []; res = {}
        for st in (0, 16, 0, 16):
            cg.set_option("mfma_sym_st", st)
            t = timeit(lambda: G.mul_(y, a)); res[st] = y.clone()
            out.append(f"st={st:2d}: {t:8.1f} us (f16={cg.get_info('last_mfma_f16')}, sym={cg.get_info('last_mfma_sym')})")
        print(f"n={n} d={d}: " + " | ".join(out) + f" | rel diff {float((res[0] - res[16]).norm() / res[0].norm()):.1e}", flush=True)
cg.set_option("mfma_sym_st", 0)
