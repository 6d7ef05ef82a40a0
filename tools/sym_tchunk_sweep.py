"""Column chunk length of the symmetric matrix-core kernels (tiles per workgroup; option jsplit sets it): us per MVM, EQ and MaternP(2), against the automatic choice."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for n, kern in [(int(v), cg.EQ()) for v in sys.argv[1:]] or ((131072, cg.EQ()), (131072, cg.MaternP(2)), (65536, cg.EQ()), (65536, cg.MaternP(2)), (32768, cg.MaternP(2))):
    rng = np.random.default_rng(1)
    X = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    G = cg.gramian(kern, X); y = torch.empty(n, dtype=torch.float32, device="cuda")
    T = (n + 31) // 32
    out = []
    for tc in (1024, 512, 256, 192, 128, 96, 64, 0):
        cg.set_option("mfma_sym", 1); cg.set_option("jsplit", 0 if tc == 0 else max(1, -(-T // tc)))
        for _ in range(3): G.mul_(y, a)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): G.mul_(y, a)
        e1.record(); e1.synchronize(); out.append(f"tc={tc or 'auto'}: {e0.elapsed_time(e1) / 30 * 1e3:.0f}")
    cg.set_option("mfma_sym", 0); cg.set_option("jsplit", 0)
    for _ in range(3): G.mul_(y, a)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): G.mul_(y, a)
    e1.record(); e1.synchronize()
    print(f"n={n} {type(kern).__name__[:7]} T={T}: " + "  ".join(out) + f"  | general {e0.elapsed_time(e1) / 30 * 1e3:.0f} us", flush=True)
