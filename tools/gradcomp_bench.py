"""C4-shaped gradient MVMs: EQ on the lane-per-row and on the panel path, and a composite (EQ*RQ) on both."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg
import covgram_oracle as o

def timeit(fn, warm=2, reps=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

n, d = 16384, 32
rng = np.random.default_rng(3)
Xh = rng.standard_normal((n, d)); ah = rng.standard_normal(n * d)
X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); y = torch.empty_like(a)
rows = np.sort(np.random.default_rng(1).choice(n, 32, replace=False))
for name, k, ko in (("EQ", cg.EQ(), o.Kernel(o.EQ)), ("EQ*RQ(1)", cg.EQ() * cg.RQ(1.0), o.Composite(((o.Kernel(o.EQ), o.Kernel(o.RQ, param=1.0)),))),
                    ("MaternP(2)+0.5EQ", cg.MaternP(2) + 0.5 * cg.EQ(), o.Composite(((o.Kernel(o.MATERNP, p=2),), (o.Kernel(o.EQ, scale=0.5),))))):
    K = cg.gramian(cg.GradientKernel(k), X)
    ref = o.grad_mul(None, ko, Xh[rows], Xh, ah)
    for label, opt in (("lane-per-row", 0), ("panel (coef + apply)", 2)):
        cg.set_option("grad_keep_r", opt)
        ms = timeit(lambda: K.mul_(y, a))
        got = y.cpu().numpy().reshape(n, d)[rows].reshape(-1)
        print(f"{name:18s} {label:22s} {ms:8.3f} ms   rel-err {np.linalg.norm(got - ref) / np.linalg.norm(ref):.2e}", flush=True)
    cg.set_option("grad_keep_r", -1)
