"""gramian(k, x) * a over a grid of (kernel, d, n), fp32: us per MVM and evaluated-pairs rate relative to the same kernel's rate at the largest n — a map for finding
planner outliers (a size or dimension where the routing rule picks a slow form)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def t_of(fn, reps):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
ns = (4096, 8192, 12000, 16384, 24000, 32768, 65536, 131072)
for name, k in (("EQ", cg.EQ()), ("MaternP2", cg.MaternP(2)), ("Cauchy", cg.Cauchy()), ("Exp", cg.Exp())):
    for d in (1, 2, 3, 4, 5, 8, 12, 16):
        cells = []
        for n in ns:
            rng = np.random.default_rng(n + d)
            X = torch.from_numpy((rng.standard_normal((n, d)) * min(1.0, 2.0 / np.sqrt(d))).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda(); y = torch.empty_like(a)
            G = cg.gramian(k, X)
            t = t_of(lambda: G.mul_(y, a), 20 if n <= 32768 else 8)
            cells.append((t, float(n) * n / t, f"{cg.get_info('last_dense_path')}{'s' if cg.get_info('last_mfma_sym') or cg.get_info('last_dense_sym') else 'g'}{'h' if cg.get_info('last_mfma_f16') else ''}"))
        top = max(c[1] for c in cells)
        print(f"{name:8s} d={d:2d}: " + "  ".join(f"{n}: {c[0]:7.1f}us {c[1] / top:4.2f} {c[2]:4s}" for n, c in zip(ns, cells)), flush=True)
