"""fp32 gradient MVMs with the library named by COVGRAM_LIB (default: the in-tree build): run once per build, alternating."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
out = []
for (n, d, kern) in ((32768, 32, cg.EQ()), (32768, 8, cg.EQ()), (65536, 3, cg.EQ()), (32768, 16, cg.RQ(1.5)), (32768, 32, cg.MaternP(2)), (16384, 64, cg.EQ())):
    rng = np.random.default_rng(0xC0F + 3)
    X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n * d).astype(np.float32)).cuda()
    K = cg.gramian(cg.GradientKernel(kern), X); y = torch.empty_like(a)
    ts = []
    for rep in range(5):
        for _ in range(2): K.mul_(y, a)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): K.mul_(y, a)
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
    out.append(f"G32({type(kern).__name__[:4]}) n={n} d={d}: {np.median(ts):.3f}")
print("lib_ab " if os.environ.get("COVGRAM_LIB") else "in-tree", " | ".join(out), flush=True)
