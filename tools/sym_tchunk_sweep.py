import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo/covariancefunctions.jl_amd")
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for n in [int(v) for v in sys.argv[1:]] or (40000, 49152, 65536, 90000):
    rng = np.random.default_rng(1)
    X = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    G = cg.gramian(cg.EQ(), X); y = torch.empty(n, dtype=torch.float32, device="cuda")
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
    print(f"n={n} T={T}: " + "  ".join(out) + f"  | general {e0.elapsed_time(e1) / 30 * 1e3:.0f} us", flush=True)
