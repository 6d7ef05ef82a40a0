"""Interleaved A/B of the matrix-core EQ path: one wave per workgroup with its own fragment loads (mfma_lds = 0) vs
four waves sharing the column tiles through LDS (mfma_lds = 1), and what the automatic rule (-1) picks."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
cases = [(131072, 3), (131072, 2), (100000, 4), (65536, 6), (65536, 8), (5000, 3), (20011, 7), (30000, 3), (50000, 3), (80000, 3), (30000, 8), (50000, 8)]
KERNELS = {"EQ": lambda l: cg.Lengthscale(cg.EQ(), l), "MaternP2": lambda l: cg.Lengthscale(cg.MaternP(2), 2 * l),
           "RQ": lambda l: cg.Lengthscale(cg.RQ(1.5), 2 * l), "Cauchy": lambda l: cg.Lengthscale(cg.Cauchy(), 2 * l), "Dot2": lambda l: cg.Dot() ** 2}
kname = "EQ"
args = sys.argv[1:]
if args and args[0] in KERNELS: kname = args.pop(0)
if args: cases = [tuple(int(v) for v in s.split("x")) for s in args]
cg.set_option("mfma_sym", 0)                      # the general kernels (two point sets, row shards), not the symmetric ones
for n, d in cases:
    rng = np.random.default_rng(0xC0F + 1)
    X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    if kname == "Dot2": X = X / (d ** 0.5)
    G = cg.gramian(KERNELS[kname](1.0 if d <= 3 else 2.0), X); y = torch.empty(n, dtype=torch.float32, device="cuda")
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    res = {}; outs = {}
    for rep in range(7):
        for v in (0, 1, -1):
            cg.set_option("mfma_lds", v)
            for _ in range(3): G.mul_(y, a)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20): G.mul_(y, a)
            e1.record(); e1.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 20)
            outs[v] = y.clone()
    print(f"{kname} n={n} d={d}: " + "  ".join(f"{ {0: 'wave', 1: 'lds4', -1: 'auto'}[v]} {np.median(t):.4f} ms" for v, t in res.items()),
          " max|diff|", float((outs[0] - outs[1]).abs().max()), "path", cg.get_info("last_dense_path"))
