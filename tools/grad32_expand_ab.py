"""fp32 GradientKernel MVM: expanded form (grad_expand = 1) against direct differences (0) — us per MVM and error vs 64 fp64 oracle block rows."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg, covgram_oracle as o, c_oracle
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for n, d in ((16384, 32), (16384, 8), (16384, 16), (8192, 48), (32768, 12)):
    rng = np.random.default_rng(5 + d)
    Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n * d).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); y = torch.empty_like(a)
    rows = np.sort(np.random.default_rng(3).choice(n, 64, replace=False))
    for name, k, ko in (("EQ", cg.EQ(), o.Kernel(o.EQ)), ("MaternP(2)", cg.MaternP(2), o.Kernel(o.MATERNP, p=2)), ("RQ(1.5)", cg.RQ(1.5), o.Kernel(o.RQ, param=1.5))):
        K = cg.gramian(cg.GradientKernel(k), X)
        ref = c_oracle.grad_mvm(ko, Xh[rows].astype(np.float64), Xh.astype(np.float64), ah.astype(np.float64))
        out = []
        for ex in (0, 1, -1):
            cg.set_option("grad_expand", ex)
            ts = []
            for rep in range(3):
                for _ in range(3): K.mul_(y, a)
                torch.cuda.synchronize(); e0.record()
                for _ in range(10): K.mul_(y, a)
                e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 10 * 1e3)
            got = y.cpu().numpy().reshape(n, d)[rows].reshape(-1).astype(np.float64)
            out.append(f"expand={ex:2d} (ran {cg.get_info('last_grad_expand')}): {np.median(ts):7.1f} us err {np.linalg.norm(got - ref) / np.linalg.norm(ref):.1e}")
        print(f"n={n} d={d} GradientKernel({name}) fp32: " + " | ".join(out), flush=True)
cg.set_option("grad_expand", -1)
