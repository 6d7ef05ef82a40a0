"""Kronecker MVM: the mode kernel's column tiling aimed at one workgroup per CU (option kron_fill = 1) against two (2): us per MVM back to back."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=50):
    ts = []
    for rep in range(5):
        for _ in range(5): fn()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts)) * 1e3
for dt in (torch.float64, torch.float32):
    for side, dims in ((128, 3), (100, 3), (64, 3), (32, 4), (48, 3), (256, 2), (64, 4), (32, 5), (96, 3), (48, 4), (128, 2)):
        ax = torch.linspace(0, 1, side, dtype=dt, device="cuda")
        G = cg.gramian(cg.separable("*", *([cg.Exp()] * dims)), cg.LazyGrid(ax, dims))
        N = side ** dims
        a = torch.randn(N, dtype=dt, device="cuda"); y = torch.empty_like(a)
        out = []; res = {}
        for f in ((1, 3, 1, 3) if len(sys.argv) > 1 else (1, 2, 1, 2)):
            cg.set_option("kron_fill", f)
            t = timeit(lambda: G.mul_(y, a)); res[f] = y.clone()
            out.append(f"fill={f}: {t:6.1f} us (path {cg.get_info('last_kron_path')})")
        print(f"kron {str(dt)[6:]} {side}^{dims}: " + " | ".join(out) + f" | rel diff {float((res[1] - res[list(res)[-1]]).norm() / res[1].norm()):.1e}", flush=True)
cg.set_option("kron_fill", 1)
