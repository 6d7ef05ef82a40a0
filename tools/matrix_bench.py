"""Matrix(G) (covgram_matrix): us per call and TB/s written, fp32 / fp64, n = 16384 and 32768 (fp32)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for dt, n, d, k in ((torch.float32, 16384, 3, cg.EQ()), (torch.float32, 32768, 3, cg.EQ()), (torch.float64, 16384, 3, cg.EQ()), (torch.float32, 16384, 3, cg.MaternP(2)), (torch.float32, 16384, 16, cg.EQ())):
    X = torch.randn(n, d, dtype=dt, device="cuda"); G = cg.gramian(k, X)
    for _ in range(3): M = G.to_dense()
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): M = G.to_dense()
    e1.record(); e1.synchronize(); t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"{str(dt)[6:]} n={n} d={d} {type(k).__name__}: {t * 1e6:8.1f} us  {n * n * X.element_size() / t * 1e-12:5.2f} TB/s written (incl. torch.empty)", flush=True)
