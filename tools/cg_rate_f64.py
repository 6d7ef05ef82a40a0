"""(G + sigma^2 I) x = b by CG in fp64 (the reference's default element type), MaternP(2) and EQ, d = 3: time per iteration with the
all-entries kernel (dense_sym = 0) and with the symmetric direct-difference kernel (default from n = 8192), plain loop and HIP-graph
replay.  Dev tool."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
for name, k in (("MaternP(2)", cg.MaternP(2)), ("EQ", cg.EQ())):
    for n in (8192, 16384, 32768):
        rng = np.random.default_rng(n)
        X = torch.from_numpy(rng.standard_normal((n, 3))).cuda(); b = torch.from_numpy(rng.standard_normal(n)).cuda()
        G = cg.gramian(k, X)
        A = G + 0.1 * torch.ones(n, device="cuda", dtype=torch.float64)
        out = []
        for mode in (0, -1):
            cg.set_option("dense_sym", mode)
            for graph in (False, True):
                for _ in range(2):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    x, info = cg.cg(A, b, reltol=1e-30, maxiter=100, graph=graph)
                    torch.cuda.synchronize(); el = time.perf_counter() - t0
                out.append(f"{'sym' if mode else 'all'} {'graph' if graph else 'loop'} {el / max(info['iterations'], 1) * 1e6:7.1f} us/it")
        cg.set_option("dense_sym", -1)
        print(f"{name:10s} n={n:6d} fp64: " + "   ".join(out), flush=True)
