"""C5: the row kernel's spectrum copy as reals (symmetric matrix, default) against the complex copy (option toeplitz_real_spectrum = 0
at handle creation), interleaved on one box."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd"))
import covgram as cg
n = 1 << 22
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for dt in (torch.float64, torch.float32):
    Ts = {}
    for real in (1, 0):
        cg.set_option("toeplitz_real_spectrum", real)
        Ts[real] = cg.gramian(cg.Exp(), cg.srange(-1, 1, n, dtype=dt))
    cg.set_option("toeplitz_real_spectrum", 1)
    a = torch.randn(n, dtype=dt, device="cuda"); ys = {r: torch.empty_like(a) for r in Ts}
    res = {}
    for rep in range(5):
        for real, T in Ts.items():
            for _ in range(5): T.mul_(ys[real], a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): T.mul_(ys[real], a)
            e1.record(); e1.synchronize(); res.setdefault(real, []).append(e0.elapsed_time(e1) / 50 * 1e3)
    diff = float((ys[1] - ys[0]).norm() / ys[0].norm())
    print(f"C5 {dt}: " + "  ".join(f"real_spectrum={k}: median {np.median(v):.1f} us min {min(v):.1f} us" for k, v in res.items()) + f"  rel diff {diff:.1e}")
