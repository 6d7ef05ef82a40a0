"""Kronecker MVM (csrc/kron.hip) across shapes: us per MVM (HIP events, median of per-call timings and back-to-back mean), the
reference's algorithmic bytes (2 tensor passes per mode) per second, and TFLOP/s of 2 N sum(n_i).
usage: python tools/kron_bench.py [quick]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg

def timeit(fn, reps=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    per = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return per[len(per) // 2], per[0], e0.elapsed_time(e1) / reps * 1e3

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
shapes = [(128, 3)] if quick else [(32, 3), (64, 3), (128, 3), (256, 3), (256, 2), (512, 2), (1024, 2), (16, 5), (32, 4), (100, 3), (48, 3)]
for dt in (torch.float64, torch.float32):
    for (side, dims) in shapes:
        ax = torch.linspace(0, 1, side, dtype=dt, device="cuda")
        G = cg.gramian(cg.separable("*", *([cg.Exp()] * dims)), cg.LazyGrid(ax, dims))
        N = side ** dims
        a = torch.randn(N, dtype=dt, device="cuda"); y = torch.empty_like(a)
        G.mul_(y, a)
        # check against the dense factor applied mode by mode in torch (fp64)
        F = G._dense_factors()[0].t().to(torch.float64)
        t = a.to(torch.float64).reshape([side] * dims)
        for axn in range(dims):
            t = torch.movedim(torch.tensordot(F, t, dims=([1], [axn])), 0, axn)
        err = float((y.to(torch.float64) - t.reshape(-1)).norm() / t.norm())
        med, mn, mean = timeit(lambda: G.mul_(y, a))
        by = dims * 2 * N * (8 if dt == torch.float64 else 4)
        print(f"kron {str(dt)[6:]} {side}^{dims} (N = {N}): median {med:.1f} us, min {mn:.1f}, back-to-back {mean:.1f} | {by / mean * 1e-6:.2f} TB/s of {dims} x (read + write), "
              f"{2.0 * N * side * dims / mean * 1e-6:.1f} TFLOP/s | rel err {err:.1e}", flush=True)
