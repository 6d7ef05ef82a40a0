"""Sustained shader clock under C3's row-shard kernel (and C2's), from the stamping build of the same loop (option mfma_stamp)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
for n, per, d, pp in ((524288, 65536, 8, 0), (131072, 131071, 3, 0), (524288, 65536, 8, 0)):
    X = torch.from_numpy(np.random.default_rng(1).standard_normal((n, d)).astype(np.float32)).cuda()
    a = torch.randn(n, dtype=torch.float32, device="cuda")
    G = cg.gramian(cg.EQ(), X[:per].contiguous(), X); y = torch.empty(per, dtype=torch.float32, device="cuda")
    cg.set_option("rows_per_lane", 2)
    for _ in range(3): G.mul_(y, a)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): G.mul_(y, a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    cg.set_option("mfma_stamp", 1)
    for _ in range(3): G.mul_(y, a)
    khz = cg.get_info("last_clock_khz")
    cg.set_option("mfma_stamp", 0); cg.set_option("rows_per_lane", 0)
    pairs = per * n
    cyc = ms * 1e-3 * khz * 1e3 * 1024 / (pairs / 64)          # SIMD cycles per 64 pairs: 1024 SIMDs
    print(f"rows {per} x cols {n} d={d}: {ms * 1e3:.1f} us, clock under the stamped kernel {khz / 1e6:.3f} GHz -> {cyc:.2f} SIMD cycles per 64 pairs (at 2.4 GHz it would read {ms * 1e-3 * 2.4e9 * 1024 / (pairs / 64):.2f})")
