"""C3's row-shard kernel (EQ, d = 8, fp32, 65536 rows x 524288 columns on the matrix cores): row tiles per wave, interleaved A/B.
rows_per_lane = 0: the default instance dense_mfma_eq_kernel<4, 2, 4, 1> (two row tiles per wave, three waves per SIMD);
rows_per_lane = 4: <4, 4, 4, 1> (four row tiles per wave share every column fragment: half the LDS reads, DMA issue and barriers per pair; two waves per SIMD).
Also d = 5 ... 7 and a C2-sized shard of 16384 rows."""
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

cases = [(524288, 65536, 8, 5), (524288, 65536, 6, 5), (131072, 16384, 8, 20), (131072, 131071, 7, 5), (32768, 32767, 8, 20), (131072, 131071, 3, 5), (131072, 16384, 3, 20), (65536, 65535, 2, 10), (131072, 131071, 4, 5)]
opts = [int(v) for v in sys.argv[1:]] or [2, 0, 4]
for n, per, d, reps in cases:
    X = torch.from_numpy(np.random.default_rng(0xC0F + 2).standard_normal((n, d)).astype(np.float32)).cuda()
    a = torch.from_numpy(np.random.default_rng(3).standard_normal(n).astype(np.float32)).cuda()
    G = cg.gramian(cg.EQ(), X[:per].contiguous(), X); y = torch.empty(per, dtype=torch.float32, device="cuda")
    res = {o: [] for o in opts}; out = {}
    for rnd in range(3):
        for o in opts:
            cg.set_option("rows_per_lane", o)
            res[o].append(timed(lambda: G.mul_(y, a), reps)); out[o] = y.clone()
            assert cg.get_info("last_dense_path") == 2
    cg.set_option("rows_per_lane", 0)
    base = out[opts[0]].double()
    line = f"rows {per} x cols {n} d={d}: " + " | ".join(f"rows_per_lane={o}: {np.median(res[o]) * 1e3:8.1f} us (diff {float((out[o].double() - base).norm() / base.norm()):.1e})" for o in opts)
    print(line, flush=True)
