"""Composite (product) gradient Gramian at the C4 shape: panel path (coefficient + apply kernels, option grad_keep_r = 2 / auto) against
the lane-per-row kernel with the per-pair interpreter (grad_keep_r = 0)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for (kern, n, d, dt) in ((cg.EQ() * cg.RQ(1.0), 16384, 32, torch.float64), (cg.EQ() * cg.Cauchy(), 16384, 32, torch.float64), (cg.MaternP(2) * cg.EQ(), 16384, 8, torch.float64), (cg.EQ() * cg.RQ(1.0), 8192, 16, torch.float64),
                          (cg.EQ() * cg.RQ(1.0), 16384, 48, torch.float64), (cg.EQ() * cg.RQ(1.0), 16384, 32, torch.float32), (cg.MaternP(2) * cg.EQ(), 16384, 8, torch.float32), (cg.EQ() * cg.RQ(1.0), 16384, 64, torch.float32)):
    rng = np.random.default_rng(1)
    X = torch.from_numpy(rng.standard_normal((n, d))).to(dt).cuda(); a = torch.from_numpy(rng.standard_normal(n * d)).to(dt).cuda()
    K = cg.gramian(cg.GradientKernel(kern), X); y = torch.empty_like(a)
    res = {}; outs = {}
    for opt in (-1, 2, 0):
        cg.set_option("grad_keep_r", opt)
        for _ in range(2): K.mul_(y, a)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): K.mul_(y, a)
        e1.record(); e1.synchronize(); res[opt] = e0.elapsed_time(e1) / 5; outs[opt] = y.clone()
    cg.set_option("grad_keep_r", -1)
    print(f"{str(dt)[6:]} n={n} d={d}: auto {res[-1]:.2f} ms  panel {res[2]:.2f} ms  lane-per-row {res[0]:.2f} ms  rel diff {float((outs[0] - outs[2]).norm() / outs[2].norm()):.1e}", flush=True)
