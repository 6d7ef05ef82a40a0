"""One-pass Sum kernels against one MVM per term on the GENERAL form (a row shard 16384 x 131072, d = 3, fp32, x ~ N(0, I)): us per MVM, alternating
the two routes twice (the first pair discarded).  Basis of the rule in csrc/dense_mfma.hip: sum_fused_applies."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
L = cg.Lengthscale
n, d, per = 131072, 3, 16384
rng = np.random.default_rng(0xC0F + 1)
X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
ys = torch.empty(per, dtype=torch.float32, device="cuda")
cases = [("1.5 MaternP(2; 0.7) + 0.5 EQ(2)", 1.5 * L(cg.MaternP(2), 0.7) + 0.5 * L(cg.EQ(), 2.0)),
         ("RQ(0.8; 0.9) + 0.2 MaternP(1)", L(cg.RQ(0.8), 0.9) + 0.2 * cg.MaternP(1)),
         ("MaternP(2) + 0.3 MaternP(1)", cg.MaternP(2) + 0.3 * cg.MaternP(1)),
         ("EQ(1.4) + 0.7 RQ(0.8; 0.9) + 0.2 MaternP(1)", L(cg.EQ(), 1.4) + 0.7 * L(cg.RQ(0.8), 0.9) + 0.2 * cg.MaternP(1)),
         ("RQ(1) + 0.5 Cauchy + MaternP(2)", cg.RQ(1.0) + 0.5 * cg.Cauchy() + cg.MaternP(2))]
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for name, kc in cases:
    Gs = cg.gramian(kc, X[:per].contiguous(), X)
    res = {}
    for rnd in range(3):
        for sf in (1, 0):
            cg.set_option("sum_fused", sf)
            for _ in range(5): Gs.mul_(ys, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): Gs.mul_(ys, a)
            e1.record(); e1.synchronize()
            assert cg.get_info("last_sum_fused") == sf
            if rnd: res.setdefault(sf, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    cg.set_option("sum_fused", -1); Gs.mul_(ys, a)
    print(f"{name:46s} one pass {min(res[1]):7.1f} us | one MVM per term {min(res[0]):7.1f} us | automatic rule: {'one pass' if cg.get_info('last_sum_fused') else 'per term'}", flush=True)
