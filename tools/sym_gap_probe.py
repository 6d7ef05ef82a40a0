"""Why does the symmetric loop of bench.py slow down after the CPU baseline?  Host time per call (no sync) and GPU time per step,
for the general and the symmetric path, with and without the OpenMP CPU baseline having run first in this process."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import bench
rng = np.random.default_rng(bench.SEED)
Xh = rng.standard_normal((bench.N_POINTS, bench.DIM)).astype(np.float32); ah = rng.standard_normal(bench.N_POINTS).astype(np.float32)
if len(sys.argv) > 1 and sys.argv[1] == "cpu":
    if len(sys.argv) > 2: os.environ["OMP_WAIT_POLICY"] = sys.argv[2]
    print("cpu baseline:", bench.cpu_baseline(Xh, ah)["value"], os.environ.get("OMP_WAIT_POLICY"))
torch.cuda.set_device(0)
import covgram as cg
X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); b = torch.empty_like(a)
for sym in (0, -1, 0, -1):
    cg.set_option("mfma_sym", sym)
    G = cg.gramian(cg.EQ(), X)
    for _ in range(20): G.mul_(b, a)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(50):
        h0 = time.perf_counter(); G.mul_(b, a); host.append(time.perf_counter() - h0)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"mfma_sym={sym}: {el / 50 * 1e3:.3f} ms per step; host call median {np.median(host) * 1e6:.0f} us, max {max(host) * 1e6:.0f} us, sum {sum(host) * 1e3:.1f} ms", flush=True)
