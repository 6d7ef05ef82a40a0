"""Measure every hot-path config of BASELINE.json (C1..C5) plus the Kronecker / low-rank rows on one MI355X and print one
JSON line per config: time, rate, rel-err vs the fp64 oracle (row subset for the large dense cases) and the roofline
the config is bound by.  Dev/evidence tool — bench.py is the contract benchmark (config C2).

    python tools/bench_configs.py > profiles/rNN_all_configs.jsonl
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg
import covgram_oracle as o

FP32_PEAK, FP64_PEAK, HBM_PEAK = 157.3e12, 78.6e12, 8.0e12


def timeit(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def rel(b, ref):
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(b - ref) / np.linalg.norm(ref))


def emit(**kw):
    print(json.dumps(kw), flush=True)


def dense(tag, kern, kern_o, n, d, dtype, check_rows=1024, note=""):
    npdt = np.float32 if dtype == torch.float32 else np.float64
    rng = np.random.default_rng(0xC0F + int(tag[1]))
    Xh = rng.standard_normal((n, d)).astype(npdt); ah = rng.standard_normal(n).astype(npdt)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    G = cg.gramian(kern, X); y = torch.empty(n, dtype=dtype, device="cuda")
    med, mn = timeit(lambda: G.mul_(y, a))
    rows = np.random.default_rng(1).choice(n, min(check_rows, n), replace=False)
    err = rel(y.cpu().numpy()[rows], o.mul(None, kern_o, Xh[rows], Xh, ah, dtype=npdt))
    # gramian(k, x) takes the library's symmetric kernels where they apply: price the pairs it EVALUATED (upper triangle incl. the
    # diagonal blocks), not the n^2 entries the product covers
    sym = cg.get_info("last_mfma_sym") == 1 or cg.get_info("last_dense_sym") == 1
    pairs = float(n) * (n + (32 if dtype == torch.float32 else 64)) / 2 if sym else float(n) * n
    flops = pairs * (3 * d + 3 + (2 if sym else 0))
    peak = FP32_PEAK if dtype == torch.float32 else FP64_PEAK
    emit(config=tag, what=f"dense {type(kern).__name__} mul!, d={d}, n={n}, {str(dtype)[6:]}{note}" + (" (symmetric kernel: upper triangle once)" if sym else ""),
         ms_median=med, ms_min=mn, mvm_per_s=1e3 / med, pairs_per_s=float(n) * n / (med * 1e-3), evaluated_pairs_per_s=pairs / (med * 1e-3),
         rel_err_vs_fp64_oracle=err, checked_rows=len(rows),
         roofline={"bound": "valu", "achieved_TFLOPs": flops / (med * 1e-3) * 1e-12, "peak_TFLOPs": peak * 1e-12,
                   "frac": flops / (med * 1e-3) / peak, "algorithmic_flops": flops, "evaluated_pairs": pairs,
                   "compulsory_bytes": (n * d + n * (d + 1) + n) * (4 if dtype == torch.float32 else 8)})


def main():
    print(json.dumps({"device": torch.cuda.get_device_name(0), "note": "one MI355X, inputs resident in HBM, torch.cuda.Event timing of the whole mul! call (pack + main + reduce kernels)"}))
    # C1: MaternP(2), d=3, n=4096, fp64 — the reference's CPU-runnable case, here on the GPU
    dense("C1", cg.MaternP(2), o.Kernel(o.MATERNP, p=2), 4096, 3, torch.float64, check_rows=4096)
    # C2: EQ, d=3, n=131072, fp32 (bench.py's workload)
    dense("C2", cg.EQ(), o.Kernel(o.EQ), 131072, 3, torch.float32)
    # C3: EQ, d=8, n=524288, fp32 is the 8-GPU config; one GPU's row shard = 65536 rows x all 524288 columns
    n3, d3, shard = 524288, 8, 65536
    rng = np.random.default_rng(0xC0F + 2)
    Xh = rng.standard_normal((n3, d3)).astype(np.float32); ah = rng.standard_normal(n3).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    G = cg.gramian(cg.EQ(), X[:shard], X); y = torch.empty(shard, dtype=torch.float32, device="cuda")
    med, mn = timeit(lambda: G.mul_(y, a), warm=2, reps=5)
    rows = np.random.default_rng(1).choice(shard, 512, replace=False)
    err = rel(y.cpu().numpy()[rows], o.mul(None, o.Kernel(o.EQ), Xh[rows], Xh, ah, dtype=np.float32))
    flops = float(shard) * n3 * (3 * d3 + 3)
    emit(config="C3-shard", what="dense EQ mul!, d=8, n=524288 fp32: ONE rank's row shard of the 8-GPU config (65536 rows x 524288 columns)",
         ms_median=med, ms_min=mn, pairs_per_s=float(shard) * n3 / (med * 1e-3), rel_err_vs_fp64_oracle=err, checked_rows=512,
         projected_8gpu_mvm_per_s=1e3 / (med + 0.03), note="projection = shard time + ~30 us all-gather of 256 KiB per rank; the 8-GPU run itself is the driver's",
         roofline={"bound": "valu", "achieved_TFLOPs": flops / (med * 1e-3) * 1e-12, "peak_TFLOPs": FP32_PEAK * 1e-12, "frac": flops / (med * 1e-3) / FP32_PEAK})
    # the form the 8-GPU run actually takes (covgram.ShardedGramian): rank r's cyclic panels of the upper triangle + one all-reduce
    Gf = cg.gramian(cg.EQ(), X); part = torch.empty(n3, dtype=torch.float32, device="cuda"); tot = torch.zeros(n3, dtype=torch.float32, device="cuda")
    per_rank = []
    for r in range(8):
        med_r, _ = timeit(lambda r=r: Gf.sym_partial_(part, a, r, 8), warm=2, reps=4)
        per_rank.append(med_r); tot += part
    err = rel(tot.cpu().numpy()[rows], o.mul(None, o.Kernel(o.EQ), Xh[rows], Xh, ah, dtype=np.float32))
    emit(config="C3-sym-partial", what="dense EQ mul!, d=8, n=524288 fp32: rank r of 8 of the symmetric form (cyclic 256-row panels of the upper "
         "triangle, covgram_mvm_sym_partial), every rank timed on this one GPU; the 8 partials summed and checked",
         ms_per_rank=per_rank, ms_median=max(per_rank), rel_err_vs_fp64_oracle=err, checked_rows=512,
         projected_8gpu_mvm_per_s=1e3 / (max(per_rank) + 0.06), note="projection = slowest rank + ~60 us all-reduce of 2 MiB; the 8-GPU run itself is the driver's")
    del G, Gf, X, a, y, part, tot
    # C4: GradientKernel(EQ), d=32, n=16384, fp64
    n4, d4 = 16384, 32
    rng = np.random.default_rng(0xC0F + 3)
    Xh = rng.standard_normal((n4, d4)); ah = rng.standard_normal(n4 * d4)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    K = cg.gramian(cg.GradientKernel(cg.EQ()), X); y = torch.empty(n4 * d4, dtype=torch.float64, device="cuda")
    med, mn = timeit(lambda: K.mul_(y, a), warm=2, reps=6)
    rows = np.sort(np.random.default_rng(1).choice(n4, 128, replace=False))
    ref = o.grad_mul(None, o.Kernel(o.EQ), Xh[rows], Xh, ah)
    got = y.cpu().numpy().reshape(n4, d4)[rows].reshape(-1)
    flops = float(n4) * n4 * (10 * d4 + 12)
    emit(config="C4", what="GradientKernel(EQ) mul!, d=32, n=16384, fp64", ms_median=med, ms_min=mn, mvm_per_s=1e3 / med,
         blocks_per_s=float(n4) * n4 / (med * 1e-3), rel_err_vs_fp64_oracle=rel(got, ref), checked_rows=128,
         roofline={"bound": "valu(fp64)", "achieved_TFLOPs": flops / (med * 1e-3) * 1e-12, "peak_TFLOPs": FP64_PEAK * 1e-12,
                   "frac": flops / (med * 1e-3) / FP64_PEAK, "algorithmic_flops": flops, "compulsory_bytes": 3 * n4 * d4 * 8})
    del K, X, a, y
    # C5: Exponential on a uniform 1-D grid, n = 2^22, fp64 — Toeplitz via rocFFT
    n5 = 1 << 22
    T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n5))
    ah = np.random.default_rng(0xC0F + 4).standard_normal(n5)
    a = torch.from_numpy(ah).cuda(); y = torch.empty_like(a)
    med, mn = timeit(lambda: T.mul_(y, a))
    vc = T.vc.cpu().numpy()
    ref = o.toeplitz_mul(None, vc, None, ah)           # numpy.fft circulant embedding (host)
    idx = np.random.default_rng(2).choice(n5, 64, replace=False)
    direct = np.array([np.dot(vc[np.abs(i - np.arange(n5))], ah) for i in idx])   # 64 rows of the dense O(n^2) product
    bytes_alg = 112.0 * n5
    emit(config="C5", what="Exponential on range(-1,1,2^22): SymmetricToeplitz mul!, fp64, rocFFT R2C/C2R at N=2n, cached spectrum",
         ms_median=med, ms_min=mn, mvm_per_s=1e3 / med, rel_err_vs_numpy_fft=rel(y.cpu().numpy(), ref),
         rel_err_vs_dense_rows=rel(y.cpu().numpy()[idx], direct), checked_rows=64,
         roofline={"bound": "hbm", "achieved_GBps": bytes_alg / (med * 1e-3) * 1e-9, "peak_GBps": HBM_PEAK * 1e-9,
                   "frac": bytes_alg / (med * 1e-3) / HBM_PEAK, "algorithmic_bytes": bytes_alg})
    del T, a, y
    # C5, fp32 variant (BASELINE.json configs[4]: "also f32 variant")
    T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n5, torch.float32))
    a32 = torch.from_numpy(ah.astype(np.float32)).cuda(); y32 = torch.empty_like(a32)
    med, mn = timeit(lambda: T.mul_(y32, a32))
    emit(config="C5-f32", what="the same Toeplitz MVM in fp32", ms_median=med, ms_min=mn, mvm_per_s=1e3 / med,
         rel_err_vs_numpy_fft=rel(y32.cpu().numpy(), ref),
         roofline={"bound": "hbm", "achieved_GBps": 0.5 * bytes_alg / (med * 1e-3) * 1e-9, "peak_GBps": HBM_PEAK * 1e-9,
                   "frac": 0.5 * bytes_alg / (med * 1e-3) / HBM_PEAK, "algorithmic_bytes": 0.5 * bytes_alg})
    del T, a32, y32
    # Kronecker: the README case (README.md:205-210): Exp^{(x)3} on a 128^3 grid, dense 128x128 factors, fp64
    ax = torch.linspace(0, 1, 128, dtype=torch.float64, device="cuda")
    Gk = cg.gramian(cg.separable("*", cg.Exp(), cg.Exp(), cg.Exp()), cg.LazyGrid(ax, 3))
    ak = torch.randn(128 ** 3, dtype=torch.float64, device="cuda"); yk = torch.empty_like(ak)
    Gk.mul_(yk, ak)
    med, mn = timeit(lambda: Gk.mul_(yk, ak))
    F = o.matrix(o.Kernel(o.EXP), ax.cpu().numpy(), ax.cpu().numpy())
    refk = o.kron_mul(None, [F, F, F], ak.cpu().numpy())
    bytes_k = 3 * 2 * 128 ** 3 * 8.0
    emit(config="Kron-README", what="(Exp x Exp x Exp) on a 128^3 LazyGrid (2,097,152 points), fp64; reference: 22.6 ms (README.md:205-210)",
         ms_median=med, ms_min=mn, rel_err_vs_fp64_oracle=rel(yk.cpu().numpy(), refk),
         roofline={"bound": "hbm", "achieved_GBps": bytes_k / (med * 1e-3) * 1e-9, "peak_GBps": HBM_PEAK * 1e-9, "frac": bytes_k / (med * 1e-3) / HBM_PEAK,
                   "algorithmic_bytes": bytes_k, "algorithmic_flops": 2.0 * 128 ** 3 * 3 * 128})
    del Gk, ak, yk
    # low rank: FiniteBasis with r = 32 basis functions on n = 2^20 points, fp32
    nl, r = 1 << 20, 32
    xs = torch.randn(nl, dtype=torch.float32, device="cuda")
    Gl = cg.gramian(cg.FiniteBasis([lambda t, i=i: torch.cos(0.37 * i * t) for i in range(r)]), xs)
    al = torch.randn(nl, dtype=torch.float32, device="cuda"); yl = torch.empty_like(al)
    med, mn = timeit(lambda: Gl.mul_(yl, al))
    U = np.stack([np.cos(0.37 * i * xs.cpu().numpy().astype(np.float64)) for i in range(r)], 1)
    bytes_l = 2.0 * nl * r * 4
    emit(config="LowRank", what=f"FiniteBasis (r={r}) Gramian U(U'a), n=2^20, fp32", ms_median=med, ms_min=mn,
         rel_err_vs_fp64_oracle=rel(yl.cpu().numpy(), o.lowrank_mul(None, U, U, al.cpu().numpy())),
         roofline={"bound": "hbm", "achieved_GBps": bytes_l / (med * 1e-3) * 1e-9, "peak_GBps": HBM_PEAK * 1e-9, "frac": bytes_l / (med * 1e-3) / HBM_PEAK,
                   "algorithmic_bytes": bytes_l})
    del Gl, al, yl, xs
    # ---- composite kernels and ValueGradientKernel ------------------------------------------------
    kc = 1.5 * cg.Lengthscale(cg.MaternP(2), 0.7) + 0.5 * cg.Lengthscale(cg.EQ(), 2.0)
    kco = o.Composite(((o.Kernel(o.MATERNP, p=2, lengthscale=0.7, scale=1.5),), (o.Kernel(o.EQ, lengthscale=2.0, scale=0.5),)), o.ISOTROPIC, 1.0)
    dense("F2-composite", kc, kco, 131072, 3, torch.float32, note="; composite 1.5*MaternP(2; l=0.7) + 0.5*EQ(l=2) one symmetric matrix-core MVM per term (the library's rule for two terms)")
    for tag, kern, ko, mul_o, blk in (("F2-valgrad", cg.ValueGradientKernel(cg.EQ()), o.Kernel(o.EQ), o.valgrad_mul, d4 + 1),
                                      ("F2-composite-grad", cg.GradientKernel(cg.EQ() * cg.RQ(1.0)), o.Composite(((o.Kernel(o.EQ), o.Kernel(o.RQ, param=1.0)),)), o.grad_mul, d4)):
        rng = np.random.default_rng(0xC0F + 7)
        Xh = rng.standard_normal((n4, d4)); ah = rng.standard_normal(n4 * blk)
        X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
        K = cg.gramian(kern, X); y = torch.empty(n4 * blk, dtype=torch.float64, device="cuda")
        med, mn = timeit(lambda: K.mul_(y, a), warm=2, reps=6)
        rows = np.sort(np.random.default_rng(1).choice(n4, 64, replace=False))
        ref = mul_o(None, ko, Xh[rows], Xh, ah)
        got = y.cpu().numpy().reshape(n4, blk)[rows].reshape(-1)
        emit(config=tag, what=f"{type(kern).__name__} of {type(kern.k).__name__} mul!, d=32, n=16384, fp64", ms_median=med, ms_min=mn,
             blocks_per_s=float(n4) * n4 / (med * 1e-3), rel_err_vs_fp64_oracle=rel(got, ref), checked_rows=64)
        del K, X, a, y


if __name__ == "__main__":
    main()
