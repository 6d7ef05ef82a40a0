"""C5 (Exponential Toeplitz, n = 2^22) fp64 and fp32: time per MVM, back-to-back."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd"))
import covgram as cg
n = 1 << 22
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for dt in (torch.float64, torch.float32):
    T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n, dtype=dt))
    a = torch.randn(n, dtype=T.dtype, device="cuda"); y = torch.empty_like(a)
    res = {}
    for rep in range(5):
        for fused in ((1, 16), (1, 4), (2, 4)):   # interleaved A/B: (toeplitz_fused, toeplitz_colfft) = the default, radix-4 column FFT, round 1's kernels
            cg.set_option("toeplitz_fused", fused[0]); cg.set_option("toeplitz_colfft", fused[1])
            for _ in range(5): T.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): T.mul_(y, a)
            e1.record(); e1.synchronize(); res.setdefault(fused, []).append(e0.elapsed_time(e1) / 50 * 1e3)
    cg.set_option("toeplitz_fused", 1); cg.set_option("toeplitz_colfft", 16)
    print(f"C5 {T.dtype}: " + "  ".join(f"(fused, colfft)={k}: median {np.median(v):.1f} us min {min(v):.1f} us" for k, v in res.items()))
