"""Per-rank cost of the multi-GPU symmetric form on ONE GPU: time covgram_mvm_sym_partial for every rank r of world P
(cyclic 256-row panels of the upper triangle) at C2 size, next to the row-sharded general kernel's shard.  The all-reduce /
all-gather is not included."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
n, d = 131072, 3
rng = np.random.default_rng(0xC0F + 1)
X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
G = cg.gramian(cg.EQ(), X)
part = torch.empty(n, dtype=torch.float32, device="cuda")
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
full = timeit(lambda: G.sym_partial_(part, a, 0, 1))
print(f"world=1: {full:.1f} us")
for P in (2, 4, 8):
    ts = [timeit(lambda r=r: G.sym_partial_(part, a, r, P)) for r in range(P)]
    per = (n + P - 1) // P
    Gs = cg.gramian(cg.EQ(), X[:per], X); ys = torch.empty(per, dtype=torch.float32, device="cuda")
    rs = timeit(lambda: Gs.mul_(ys, a))
    print(f"world={P}: symmetric partial per rank max {max(ts):.1f} us  min {min(ts):.1f} us  (ideal {full / P:.1f});  row shard of the general kernel {rs:.1f} us   per rank: " + " ".join(f"{t:.0f}" for t in ts))
for js in (128, 64, 43, 32, 16):
    cg.set_option("jsplit", js)
    ts = [timeit(lambda r=r: G.sym_partial_(part, a, r, 8)) for r in range(8)]
    print(f"world=8, column chunk of {-(-4096 // js)} tiles: per rank " + " ".join(f"{t:.0f}" for t in ts))
cg.set_option("jsplit", 0)
