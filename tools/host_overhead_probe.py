"""Host-side cost of one mul! call (python + ctypes + libcovgram host code) vs its device time, C1 shape."""
import os, sys, time, cProfile, pstats, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
n = 4096
rng = np.random.default_rng(0xC0F)
X = torch.from_numpy(rng.standard_normal((n, 3))).cuda(); a = torch.from_numpy(rng.standard_normal(n)).cuda(); y = torch.empty_like(a)
for name, k in (("EQ", cg.EQ()), ("MaternP(2)", cg.MaternP(2)), ("EQ again", cg.EQ()), ("MaternP(2) again", cg.MaternP(2))):
    G = cg.gramian(k, X)
    for _ in range(20): G.mul_(y, a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): G.mul_(y, a)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host issue {1e6 * (t1 - t0) / 200:.1f} us per call, wall {1e6 * (t2 - t0) / 200:.1f} us per call")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): G.mul_(y, a)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
