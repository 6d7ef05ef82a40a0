"""C4 (GradientKernel(EQ), d = 32, n = 16384, fp64) and neighbours: direct differences (6 fp64 instructions per dimension and
pair) against the expanded form (4), interleaved rounds in one process."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg
import covgram_oracle as o, c_oracle
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for (n, d, kern, ko, vg) in ((16384, 32, cg.EQ(), o.Kernel(o.EQ), 0), (16384, 32, cg.MaternP(2), o.Kernel(o.MATERNP, p=2), 0), (16384, 32, cg.EQ(), o.Kernel(o.EQ), 1),
                             (32768, 8, cg.EQ(), o.Kernel(o.EQ), 0), (16384, 48, cg.EQ(), o.Kernel(o.EQ), 0), (65536, 3, cg.EQ(), o.Kernel(o.EQ), 0)):
    rng = np.random.default_rng(0xC0F + 3)
    Xh = rng.standard_normal((n, d)); ah = rng.standard_normal(n * (d + vg))
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    K = cg.gramian((cg.ValueGradientKernel if vg else cg.GradientKernel)(kern), X); y = torch.empty_like(a)
    res = {0: [], 1: []}; outs = {}
    for rep in range(4):
        for ex in (0, 1):
            cg.set_option("grad_expand", ex)
            for _ in range(2): K.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(5): K.mul_(y, a)
            e1.record(); e1.synchronize(); res[ex].append(e0.elapsed_time(e1) / 5)
            outs[ex] = y.cpu().numpy().copy()
    err = ""
    if not vg:
        rows = np.sort(np.random.default_rng(3).choice(n, 64, replace=False))
        ref = c_oracle.grad_mvm(ko, Xh[rows], Xh, ah)
        err = "  rel-err vs oracle: direct %.1e expanded %.1e" % tuple(np.linalg.norm(outs[e].reshape(n, d)[rows].reshape(-1) - ref) / np.linalg.norm(ref) for e in (0, 1))
    print(f"{'ValueGradient' if vg else 'Gradient'}({type(kern).__name__}) n={n} d={d} fp64: direct {np.median(res[0]):.3f} ms  expanded {np.median(res[1]):.3f} ms ({np.median(res[1]) / np.median(res[0]):.3f}x){err}", flush=True)
cg.set_option("grad_expand", -1)
