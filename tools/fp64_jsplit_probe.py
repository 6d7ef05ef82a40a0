"""fp64 dense MVM (general path, n = 32768, d = 3) against the column split: how much of the per-pair time is stall that more
resident waves would hide.  Dev tool."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
def timeit(fn, warm=3, reps=7):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
rng = np.random.default_rng(5)
for d in (3, 8):
    X = torch.from_numpy(rng.standard_normal((n, d))).cuda(); Y = torch.from_numpy(rng.standard_normal((n, d))).cuda()
    a = torch.from_numpy(rng.standard_normal(n)).cuda(); y = torch.empty_like(a)
    for name, k in (("EQ", cg.EQ()), ("MaternP(2)", cg.MaternP(2))):
        G = cg.gramian(k, X, Y)
        row = []
        for js in (0, 4, 8, 16, 32, 64, 128, 0):
            cg.set_option("jsplit", js)
            ms = timeit(lambda: G.mul_(y, a))
            row.append(f"jsplit {js} -> {cg.get_info('last_jsplit')}: {ms:.3f} ms ({3.93e13 * ms * 1e-3 / (n * n):.1f} slots)")
        cg.set_option("jsplit", 0)
        print(f"d={d} {name}: " + "  ".join(row), flush=True)
