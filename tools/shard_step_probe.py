"""Contract workload (EQ, d = 3, n = 131072, fp32, all n*m entries) as rank 0 of P sees it: the row shard's MVM (every kernel of
the call: weight pack, matrix-core kernel, split-J reduction) timed back to back on one GPU, against 1/P of the P = 1 time."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
n, d = 131072, 3
rng = np.random.default_rng(20240607)
X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
cg.set_option("mfma_sym", 0)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
base = None
# round 4: option "inkernel_reduce" (the last-arriving workgroup of a row block sums the split-J slab: no reduce launch) A/B, interleaved
def measure(G, y):
    ts = {0: [], 1: []}
    for rep in range(5):
        for ikr in (0, 1):
            cg.set_option("inkernel_reduce", ikr)
            for _ in range(10): G.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): G.mul_(y, a)
            e1.record(); e1.synchronize(); ts[ikr].append(e0.elapsed_time(e1) / 50 * 1e3)
    return float(np.median(ts[0])), float(np.median(ts[1]))
for P in (1, 2, 4, 8, 16):
    per = n // P
    G = cg.gramian(cg.EQ(), X[:per], X); y = torch.empty(per, dtype=torch.float32, device="cuda")
    t0, t = measure(G, y)
    if base is None: base = t
    print(f"P={P}: rows {per}: {t:.1f} us per MVM  (ideal {base / P:.1f} us, {base / P / t:.3f} of it)   with the separate reduce launch: {t0:.1f} us", flush=True)
# small problems: the whole gramian(k, x) MVM, fp32 EQ on the matrix cores and fp64 MaternP(2) on the lane-per-row kernel (C1 is n = 4096 fp64)
cg.set_option("mfma_sym", -1)
for dt, k, name in ((torch.float32, cg.EQ(), "EQ fp32"), (torch.float64, cg.MaternP(2), "MaternP(2) fp64")):
    for nn in (2048, 4096, 8192, 16384):
        Xs = X[:nn].to(dt).contiguous(); as_ = a[:nn].to(dt).contiguous(); ys = torch.empty_like(as_)
        G = cg.gramian(k, Xs)
        cg.set_option("dense_sym", 0)
        t0, t = measure_small = None, None
        ts = {0: [], 1: []}
        for rep in range(5):
            for ikr in (0, 1):
                cg.set_option("inkernel_reduce", ikr)
                for _ in range(20): G.mul_(ys, as_)
                torch.cuda.synchronize(); e0.record()
                for _ in range(200): G.mul_(ys, as_)
                e1.record(); e1.synchronize(); ts[ikr].append(e0.elapsed_time(e1) / 200 * 1e3)
        print(f"{name} n={nn}: {np.median(ts[1]):.1f} us per MVM   with the separate reduce launch: {np.median(ts[0]):.1f} us   (jsplit {cg.get_info('last_jsplit')}, in-kernel {cg.get_info('last_inkernel_reduce')})", flush=True)
cg.set_option("dense_sym", -1); cg.set_option("inkernel_reduce", -1)
