"""Symmetric EQ kernel: one row tile per wave (8-wave workgroups) against two (4-wave workgroups, dense_mfma_sym2.hpp), option mfma_sym_rt."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn, reps):
    ts = []
    for rep in range(5):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts)) * 1e3
for name, n, d, part, l in (("C2 symmetric", 131072, 3, False, 1.0), ("C3 sym partial (rank 3 of 8)", 524288, 8, True, 1.0), ("d=8 n=131072", 131072, 8, False, 1.0), ("d=6 l=0.8 (bf16 split)", 131072, 6, False, 0.8),
                          ("d=16 n=65536", 65536, 16, False, 1.5), ("n=32768 d=3", 32768, 3, False, 1.0), ("C2 partial rank 1 of 8", 131072, 3, True, 1.0)):
    rng = np.random.default_rng(3 + d)
    X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    G = cg.gramian(cg.Lengthscale(cg.EQ(), l), X); y = torch.empty(n, dtype=torch.float32, device="cuda")
    fn = (lambda: G.sym_partial_(y, a, 3 if n > 200000 else 1, 8)) if part else (lambda: G.mul_(y, a))
    out = []; res = {}
    for rt in (1, 2, 1, 2):
        cg.set_option("mfma_sym_rt", rt)
        t = timeit(fn, 5 if n > 200000 else 10)
        res[rt] = y.clone()
        out.append(f"rt={rt} ({cg.get_info('last_mfma_sym_rt')}, f16={cg.get_info('last_mfma_f16')}): {t:7.1f} us")
    diff = float((res[1] - res[2]).norm() / res[1].norm())
    print(f"{name}: " + " | ".join(out) + f" | rel diff {diff:.1e}", flush=True)
cg.set_option("mfma_sym_rt", -1)
