"""Matrix-core EQ kernel at d = 17 ... 32 (fp16 split: six / eight MFMAs per tile): one row tile per wave on the split-tile LDS form (default) against two row tiles per wave
(option rows_per_lane = 2), interleaved; us per MVM."""
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
opts = [int(v) for v in sys.argv[1:]] or [0, 2]
for n, per, d, l, reps in ((32768, 32767, 32, 1.0, 10), (32768, 32767, 24, 1.0, 10), (65536, 65535, 20, 1.0, 5), (65536, 65535, 16, 1.0, 5), (65536, 65535, 12, 1.0, 5), (32768, 32767, 32, 0.85, 10)):
    X = torch.from_numpy((np.random.default_rng(1).standard_normal((n, d)) * (0.7 if d >= 24 else 1.0)).astype(np.float32)).cuda()
    a = torch.randn(n, dtype=torch.float32, device="cuda")
    G = cg.gramian(cg.Lengthscale(cg.EQ(), l), X[:per].contiguous(), X); y = torch.empty(per, dtype=torch.float32, device="cuda")
    res = {o: [] for o in opts}; out = {}
    for rnd in range(3):
        for o in opts:
            cg.set_option("rows_per_lane", o); cg.set_option("mfma_lds", 1 if o == 2 else -1)
            res[o].append(timed(lambda: G.mul_(y, a), reps)); out[o] = y.clone()
            info = (cg.get_info("last_dense_path"), cg.get_info("last_mfma_f16"), cg.get_info("last_mfma_lds"))
    cg.set_option("rows_per_lane", 0); cg.set_option("mfma_lds", -1)
    base = out[opts[0]].double()
    print(f"{per} x {n} d={d} l={l} (path, fp16, lds of the last variant {info}): " + " | ".join(f"rows_per_lane={o}: {np.median(res[o]) * 1e3:8.1f} us (diff {float((out[o].double() - base).norm() / base.norm()):.1e})" for o in opts), flush=True)
