"""C2 (EQ, d=3, n=131072, fp32): direct-difference lane-per-row kernel vs the matrix-core path, time and accuracy."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg
import covgram_oracle as o

def timeit(fn, warm=3, reps=12):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))

print(torch.cuda.get_device_name(0))
for (n, d) in ((131072, 3), (65536, 8), (32768, 32)):
    rng = np.random.default_rng(0xC0F + 1)
    Xh = (rng.standard_normal((n, d)) / (1.0 if d == 3 else np.sqrt(d / 3.0))).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); y = torch.empty(n, dtype=torch.float32, device="cuda")
    rows = np.random.default_rng(1).choice(n, 1024, replace=False)
    ref = o.mul(None, o.Kernel(o.EQ), Xh[rows], Xh, ah, dtype=np.float32)
    G = cg.gramian(cg.EQ(), X)
    for label, opts in (("direct differences (dense_variant=1)", {"dense_variant": 1}),
                        ("bf16x3 MFMA, 2 row tiles/wave (auto)", {"dense_variant": 0, "rows_per_lane": 0}),
                        ("bf16x3 MFMA, 1 row tile/wave", {"dense_variant": 0, "rows_per_lane": 1})):
        for k, v in opts.items(): cg.set_option(k, v)
        med, mn = timeit(lambda: G.mul_(y, a))
        err = float(np.linalg.norm(y.cpu().numpy()[rows].astype(np.float64) - ref) / np.linalg.norm(ref))
        print(f"n={n} d={d}  {label:40s} median {med:7.3f} ms  min {mn:7.3f} ms  {1e3 / med:7.1f} MVM/s  rel-err {err:.2e}", flush=True)
    cg.set_option("dense_variant", 0); cg.set_option("rows_per_lane", 0)
