"""Matrix(G) (dense instantiation, src/gramian.jl:102-114): ms and GB/s written, fp32 / fp64, a few profiles and shapes."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for (kern, dt, n, m, d) in ((cg.EQ(), torch.float32, 16384, 16384, 3), (cg.EQ(), torch.float64, 16384, 16384, 3), (cg.MaternP(2), torch.float64, 16384, 16384, 3),
                            (cg.EQ(), torch.float32, 16384, 16384, 32), (cg.EQ(), torch.float64, 4096, 4096, 3), (cg.MaternP(2), torch.float32, 32768, 8192, 8)):
    rng = np.random.default_rng(1)
    X = torch.from_numpy(rng.standard_normal((n, d))).to(dt).cuda(); Y = torch.from_numpy(rng.standard_normal((m, d))).to(dt).cuda()
    G = cg.gramian(kern, X, Y)
    by = n * m * (4 if dt == torch.float32 else 8)
    res = {}; Ms = {}
    for var in (1, 0):
        cg.set_option("matrix_variant", var)
        for _ in range(3): M = G.to_dense()
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): M = G.to_dense()
        e1.record(); e1.synchronize(); res[var] = e0.elapsed_time(e1) / 10; Ms[var] = M
    same = bool(torch.equal(Ms[0], Ms[1]))
    print(f"{type(kern).__name__[:6]} {str(dt)[6:]} {n}x{m} d={d}: generic {res[1]:.3f} ms ({by / res[1] * 1e-6:.0f} GB/s)  rows in registers {res[0]:.3f} ms ({by / res[0] * 1e-6:.0f} GB/s)  identical: {same}", flush=True)
    del M, Ms
