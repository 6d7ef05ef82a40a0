"""Round 4, C5 (Exponential Toeplitz, n = 2^22): persistent row-kernel workgroups that prefetch the next row pair (option toeplitz_persist:
0 = one pair per workgroup, 1 = one workgroup per CU, k > 1 = k workgroups) — and, option toeplitz_colpersist, the same for the column
kernels — interleaved A/B; results checked against the one-pair kernels bit for bit (same arithmetic, same order)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd"))
import covgram as cg
n = 1 << 22
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
variants = [(0, 0), (1, 0), (512, 0), (128, 0)]
try:
    cg.set_option("toeplitz_persist", 0)
    variants += [(0, 1), (1, 1), (1, 2)]
except Exception:
    pass
for dt in (torch.float64, torch.float32):
    T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n, dtype=dt))
    a = torch.randn(n, dtype=T.dtype, device="cuda"); y = torch.empty_like(a)
    res, outs = {}, {}
    for rep in range(5):
        for v in variants:
            cg.set_option("toeplitz_persist", v[0])
            # (the column kernel's persistent variant of round 4 — option toeplitz_colpersist — did not pay and was removed with its option)
            for _ in range(5): T.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): T.mul_(y, a)
            e1.record(); e1.synchronize(); res.setdefault(v, []).append(e0.elapsed_time(e1) / 50 * 1e3)
            outs[v] = y.clone()
    same = {v: bool(torch.equal(outs[v], outs[variants[0]])) for v in variants}
    print(f"C5 {T.dtype}: " + "  ".join(f"(row persist, col persist)={k}: median {np.median(v):.1f} us min {min(v):.1f} (identical {same[k]})" for k, v in res.items()), flush=True)
cg.set_option("toeplitz_persist", 1)
