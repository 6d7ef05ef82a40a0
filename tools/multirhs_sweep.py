"""Dense Gramian times a matrix (p right-hand sides), fp32 and fp64: ms per product against p."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for (kern, dt, n, d) in ((cg.EQ(), torch.float32, 32768, 3), (cg.MaternP(2), torch.float32, 32768, 3), (cg.EQ(), torch.float64, 16384, 3), (cg.EQ(), torch.float32, 32768, 8)):
    rng = np.random.default_rng(n)
    X = torch.from_numpy(rng.standard_normal((n, d))).to(dt).cuda()
    G = cg.gramian(kern, X)
    line = []
    for p in (1, 2, 4, 8, 16, 32, 64):
        A = torch.from_numpy(rng.standard_normal((n, p))).to(dt).cuda()
        for _ in range(3): B = G @ A
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): B = G @ A
        e1.record(); e1.synchronize(); line.append(f"p={p}: {e0.elapsed_time(e1) / 10:.3f}")
    print(f"{type(kern).__name__[:6]} {str(dt)[6:]} n={n} d={d} (ms): " + "  ".join(line), flush=True)
