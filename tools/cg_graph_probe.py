"""CG through the hot path, eager loop vs the iteration body replayed as a HIP graph (covgram.cg(graph=True))."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
for name, k, n, d, dt, sig in (("C1-shaped MaternP(2) fp64", cg.MaternP(2), 4096, 3, torch.float64, 1e-2),
                               ("EQ fp32", cg.EQ(), 16384, 3, torch.float32, 1e-1),
                               ("EQ fp32", cg.EQ(), 131072, 3, torch.float32, 1e-1)):
    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.standard_normal((n, d))).to(dt).cuda()
    b = torch.from_numpy(rng.standard_normal(n)).to(dt).cuda()
    S = cg.gramian(k, X) + sig * torch.ones(n, device="cuda", dtype=dt)
    for graph in (False, True):
        best = None
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            x, info = cg.cg(S, b, reltol=1e-6 if dt == torch.float32 else 1e-10, maxiter=200, graph=graph)
            torch.cuda.synchronize(); dtm = time.perf_counter() - t0
            best = dtm if best is None else min(best, dtm)
        print(f"{name} n={n}: graph={graph}  {info['iterations']} iterations  {best * 1e3:8.2f} ms total  {best / info['iterations'] * 1e6:8.1f} us/iteration  converged={info['converged']}")
