"""dense_mfma_mrhs_kernel: G @ A for p right-hand sides against the 4-at-a-time VALU form (option mfma_mrhs = 0) and the fp64 oracle; timing."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg, covgram_oracle as o
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
rng = np.random.default_rng(3)
# correctness at ragged shapes
for (kern, ko, n, m, d, p) in ((cg.EQ(), o.Kernel(o.EQ), 1000, 1537, 3, 13), (cg.MaternP(2), o.Kernel(o.MATERNP, p=2), 777, 2050, 5, 40),
                               (cg.RQ(1.5), o.Kernel(o.RQ, param=1.5), 2049, 999, 8, 70), (cg.Dot() ** 2, o.Kernel(o.DOT, power=2), 515, 1025, 4, 33)):
    X = rng.standard_normal((n, d)).astype(np.float32) * 0.7; Y = rng.standard_normal((m, d)).astype(np.float32) * 0.7
    A = rng.standard_normal((m, p)).astype(np.float32)
    G = cg.gramian(kern, torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
    outs = {}
    for opt in (0, -1):
        cg.set_option("mfma_mrhs", opt)
        outs[opt] = (G @ torch.from_numpy(A).cuda()).cpu().numpy()
    cg.set_option("mfma_mrhs", -1)
    ref = np.stack([o.mul(None, ko, X, Y, A[:, c], dtype=np.float32) for c in range(p)], 1)
    print(type(kern).__name__[:8], n, m, d, p, "rel err mrhs", float(np.linalg.norm(outs[-1] - ref) / np.linalg.norm(ref)), "valu", float(np.linalg.norm(outs[0] - ref) / np.linalg.norm(ref)), "path", cg.get_info("last_dense_path"), flush=True)
# timing
for (kern, n, d) in ((cg.EQ(), 32768, 3), (cg.MaternP(2), 32768, 3), (cg.EQ(), 32768, 8)):
    X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); G = cg.gramian(kern, X)
    line = []
    for p in (8, 12, 16, 32, 64, 128):
        A = torch.from_numpy(rng.standard_normal((n, p)).astype(np.float32)).cuda()
        ts = {}
        for opt in (0, -1):
            cg.set_option("mfma_mrhs", opt)
            for _ in range(3): B = G @ A
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): B = G @ A
            e1.record(); e1.synchronize(); ts[opt] = e0.elapsed_time(e1) / 10
        line.append(f"p={p}: {ts[0]:.3f} / {ts[-1]:.3f}")
    cg.set_option("mfma_mrhs", -1)
    print(f"{type(kern).__name__[:6]} n={n} d={d} (VALU / matrix cores, ms): " + "  ".join(line), flush=True)
