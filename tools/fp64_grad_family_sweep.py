"""fp64 GradientKernel MVM per profile at the C4 shape (n = 16384, d = 32) and at d = 8 (n = 16384): ms per MVM.  Dev tool."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg

def timeit(fn, warm=3, reps=7):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

kernels = [("EQ", cg.EQ()), ("Exp", cg.Exp()), ("RQ(1.5)", cg.RQ(1.5)), ("GammaExp(1.5)", cg.GammaExp(1.5)), ("Cauchy", cg.Cauchy()),
           ("IMQ(1)", cg.InverseMultiQuadratic(1.0)), ("MaternP(1)", cg.MaternP(1)), ("MaternP(2)", cg.MaternP(2)), ("MaternP(3)", cg.MaternP(3))]
n = 16384
for d in (32, 8):
    rng = np.random.default_rng(6)
    X = torch.from_numpy(rng.standard_normal((n, d)) * (4.0 / np.sqrt(d))).cuda()
    a = torch.from_numpy(rng.standard_normal(n * d)).cuda(); y = torch.empty_like(a)
    for name, k in kernels:
        K = cg.gramian(cg.GradientKernel(k), X)
        ms = timeit(lambda: K.mul_(y, a))
        print(f"grad d={d} {name:16s} {ms:8.3f} ms", flush=True)
