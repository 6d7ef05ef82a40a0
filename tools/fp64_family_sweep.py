"""fp64 dense MVM rate per profile (the reference's default element type is Float64): n = 32768 against itself through the general
(all-entries) path, d = 3 and 8; prints Tpairs/s and the implied fp64 lane-instructions per pair at the chip's fp64 issue rate
(1024 SIMDs x 64 lanes / 4 cycles x 2.4 GHz = 3.93e13 per second).  Dev tool."""
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
           ("IMQ(1)", cg.InverseMultiQuadratic(1.0)), ("MaternP(1)", cg.MaternP(1)), ("MaternP(2)", cg.MaternP(2)), ("MaternP(3)", cg.MaternP(3)),
           ("Dot^2", cg.Dot() ** 2), ("ExpDot", cg.ExponentialDot()), ("Lengthscale(MaternP(2),0.7)", cg.Lengthscale(cg.MaternP(2), 0.7))]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for d in (3, 8):
    rng = np.random.default_rng(5)
    X = torch.from_numpy(rng.standard_normal((n, d)) * (0.3 if d == 8 else 1.0)).cuda()
    Y = torch.from_numpy(rng.standard_normal((n, d)) * (0.3 if d == 8 else 1.0)).cuda()
    a = torch.from_numpy(rng.standard_normal(n)).cuda(); y = torch.empty_like(a)
    for name, k in kernels:
        G = cg.gramian(k, X, Y)
        ms = timeit(lambda: G.mul_(y, a))
        rate = n * n / (ms * 1e-3)
        print(f"d={d} {name:28s} {ms:8.3f} ms  {rate*1e-12:6.3f} Tpairs/s  ~{3.93e13/rate:5.1f} fp64 issue slots per pair", flush=True)
