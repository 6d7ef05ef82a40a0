"""C4 and neighbours with the library named by COVGRAM_LIB (default: the in-tree build): run once per build, alternating, on one box."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
out = []
for (n, d, kern, vg) in ((16384, 32, cg.EQ(), 0), (16384, 32, cg.EQ(), 1), (32768, 8, cg.EQ(), 0), (16384, 48, cg.EQ(), 0), (65536, 3, cg.EQ(), 0), (16384, 32, cg.RQ(1.5), 0),
                          (16384, 32, cg.MaternP(2), 0), (16384, 32, cg.Exp(), 0), (16384, 32, cg.GammaExponential(1.3), 0),
                          (20000, 32, cg.EQ(), 0), (4096, 32, cg.EQ(), 0), (50000, 8, cg.EQ(), 0), (8192, 16, cg.EQ(), 0)):
    rng = np.random.default_rng(0xC0F + 3)
    X = torch.from_numpy(rng.standard_normal((n, d))).cuda(); a = torch.from_numpy(rng.standard_normal(n * (d + vg))).cuda()
    K = cg.gramian((cg.ValueGradientKernel if vg else cg.GradientKernel)(kern), X); y = torch.empty_like(a)
    ts = []
    for rep in range(5):
        for _ in range(2): K.mul_(y, a)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): K.mul_(y, a)
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
    out.append(f"{'VG' if vg else 'G'}({type(kern).__name__[:4]}) n={n} d={d}: {np.median(ts):.3f}")
# dense fp64 MVMs (direct-difference kernel)
for (n, d, kern) in ((32768, 3, cg.EQ()), (32768, 3, cg.MaternP(2)), (32768, 3, cg.Exp())):
    rng = np.random.default_rng(5)
    X = torch.from_numpy(rng.standard_normal((n, d))).cuda(); a = torch.from_numpy(rng.standard_normal(n)).cuda()
    K = cg.gramian(kern, X); y = torch.empty_like(a)
    ts = []
    for rep in range(5):
        for _ in range(2): K.mul_(y, a)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): K.mul_(y, a)
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
    out.append(f"dense({type(kern).__name__[:4]}) n={n} d={d}: {np.median(ts):.3f}")
print("lib_ab " if os.environ.get("COVGRAM_LIB") else "in-tree", " | ".join(out), flush=True)
