"""Interleaved A/B of the matrix-core EQ path on a symmetric Gramian gramian(EQ, x): all n^2 tiles (mfma_sym = 0) vs the
upper triangle evaluated once and used for row and column sums (mfma_sym = 1); rel-err of both against the fp64 oracle on
sampled rows."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg
import covgram_oracle as o
cases = [(131072, 3), (65536, 8), (32768, 3), (262144, 3), (100003, 5)]
KERNELS = {"EQ": (lambda l: cg.Lengthscale(cg.EQ(), l), lambda l: o.Kernel(o.EQ, lengthscale=l)),
           "MaternP2": (lambda l: cg.Lengthscale(cg.MaternP(2), l), lambda l: o.Kernel(o.MATERNP, p=2, lengthscale=l)),
           "RQ": (lambda l: cg.Lengthscale(cg.RQ(1.5), l), lambda l: o.Kernel(o.RQ, param=1.5, lengthscale=l)),
           "Cauchy": (lambda l: cg.Lengthscale(cg.Cauchy(), l), lambda l: o.Kernel(o.CAUCHY, lengthscale=l)),
           "Dot2": (lambda l: cg.Dot() ** 2, lambda l: o.Kernel(o.DOT, power=2))}
kname = "EQ"
args = sys.argv[1:]
if args and args[0] in KERNELS: kname = args.pop(0)
if args: cases = [tuple(int(v) for v in s.split("x")) for s in args]
for n, d in cases:
    rng = np.random.default_rng(0xC0F + 1)
    Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    l = (1.0 if d <= 3 else 2.0) * (1.0 if kname == "EQ" else 2.0)
    if kname == "Dot2": Xh = (Xh / np.sqrt(d)).astype(np.float32); X = torch.from_numpy(Xh).cuda()
    G = cg.gramian(KERNELS[kname][0](l), X); y = torch.empty(n, dtype=torch.float32, device="cuda")
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    res = {}; outs = {}; used = {}
    for rep in range(5):
        for v in (0, 1):
            cg.set_option("mfma_sym", v)
            for _ in range(3): G.mul_(y, a)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20): G.mul_(y, a)
            e1.record(); e1.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 20)
            outs[v] = y.cpu().numpy().astype(np.float64); used[v] = cg.get_info("last_mfma_sym")
    cg.set_option("mfma_sym", -1)
    rows = np.random.default_rng(1).choice(n, 512, replace=False)
    ref = o.mul(None, KERNELS[kname][1](l), Xh[rows], Xh, ah, dtype=np.float32)
    err = {v: float(np.linalg.norm(outs[v][rows] - ref) / np.linalg.norm(ref)) for v in (0, 1)}
    print(f"{kname} n={n} d={d}: full {np.median(res[0]):.4f} ms (rel-err {err[0]:.2e})   symmetric {np.median(res[1]):.4f} ms (rel-err {err[1]:.2e}, used={used[1]})"
          f"   full-vs-sym {float(np.linalg.norm(outs[0] - outs[1]) / np.linalg.norm(outs[0])):.2e}", flush=True)
