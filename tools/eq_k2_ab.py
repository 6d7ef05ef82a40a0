"""d = 7 against d = 8 (and 5 / 6, 15 / 16) on the fp32 EQ matrix-core kernels: odd d carries the norm pseudo-coordinate in the idle
half of its last MFMA, even d pays one more MFMA per tile.  Run once per library build (COVGRAM_LIB=...) on the same box."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
from covgram import _ffi
if os.environ.get("COVGRAM_LIB"):   # an older build: drop the entry points it does not have yet
    import ctypes
    _l = ctypes.CDLL(os.environ["COVGRAM_LIB"])
    for _k in list(_ffi.PROTOTYPES):
        if not hasattr(_l, _k): _ffi.PROTOTYPES.pop(_k)

def timeit(fn, warm=3, reps=10, inner=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        for _ in range(inner): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / inner)
    return float(np.median(ts)), float(np.min(ts))

print(os.environ.get("COVGRAM_LIB", "lib/libcovgram.so"))
for rep in range(2):
    for (n, m, d, sym) in ((65536, 524288, 7, 0), (65536, 524288, 8, 0), (65536, 65536, 5, 0), (65536, 65536, 6, 0), (65536, 65536, 7, 0), (65536, 65536, 8, 0),
                           (131072, 131072, 3, 0), (131072, 131072, 4, 0), (200000, 200000, 7, 1), (200000, 200000, 8, 1), (65536, 65536, 15, 0), (65536, 65536, 16, 0)):
        rng = np.random.default_rng(d)
        Y = torch.from_numpy(rng.standard_normal((m, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(m).astype(np.float32)).cuda()
        G = cg.gramian(cg.EQ(), Y) if n == m else cg.gramian(cg.EQ(), Y[:n], Y)
        y = torch.empty(n, dtype=torch.float32, device="cuda")
        cg.set_option("mfma_sym", sym)
        med, mn = timeit(lambda: G.mul_(y, a))
        print(f"n={n} m={m} d={d} sym={sym}: median {med:.4f} min {mn:.4f} ms", flush=True)
