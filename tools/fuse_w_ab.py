"""General matrix-core EQ kernel: column weights formed in the kernel (option mfma_fuse_w = 1, the default) against the weight-pack launch in front of it (0); us per MVM, interleaved."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for n, per, d, reps in ((131072, 131071, 3, 10), (131072, 16384, 3, 50), (131072, 8192, 3, 50), (16384, 16383, 3, 100), (4096, 4095, 3, 200), (2048, 2047, 3, 200), (524288, 65536, 8, 5), (131072, 16384, 8, 50), (32768, 32767, 16, 20), (65536, 256, 3, 200)):
    X = torch.from_numpy(np.random.default_rng(1).standard_normal((n, d)).astype(np.float32) * (0.6 if d > 8 else 1.0)).cuda(); a = torch.randn(n, dtype=torch.float32, device="cuda")
    G = cg.gramian(cg.EQ(), X[:per].contiguous(), X); y = torch.empty(per, dtype=torch.float32, device="cuda")
    res = {0: [], 1: []}; out = {}
    for rnd in range(5):
        for o in (0, 1):
            cg.set_option("mfma_fuse_w", o); res[o].append(timed(lambda: G.mul_(y, a), reps)); out[o] = y.clone()
    cg.set_option("mfma_fuse_w", -1)
    print(f"{per} x {n} d={d}: pack launch {np.median(res[0]) * 1e3:8.1f} us | in the kernel {np.median(res[1]) * 1e3:8.1f} us   identical {bool(torch.equal(out[0], out[1]))}", flush=True)
