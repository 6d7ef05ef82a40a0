"""(G + sigma^2 I) x = b by CG at GP-sized n: time per iteration against the bare MVM (fp32 EQ, d = 3): plain loop and HIP-graph replay."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
CASES = ((8192, torch.float32), (16384, torch.float32), (32768, torch.float32), (16384, torch.float64))
if len(sys.argv) > 1:                                                          # e.g. `cg_rate.py 16384 float32` (one case: kernel traces)
    CASES = ((int(sys.argv[1]), getattr(torch, sys.argv[2] if len(sys.argv) > 2 else "float32")),)
for (n, dt) in CASES:
    rng = np.random.default_rng(n)
    X = torch.from_numpy(rng.standard_normal((n, 3))).to(dt).cuda(); b = torch.from_numpy(rng.standard_normal(n)).to(dt).cuda()
    G = cg.gramian(cg.EQ(), X)
    A = G + 0.1 * torch.ones(n, device="cuda", dtype=dt)                       # G + sigma^2 I stays lazy
    y = torch.empty_like(b)
    for _ in range(10): G.mul_(y, b)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): G.mul_(y, b)
    e1.record(); e1.synchronize(); mvm = e0.elapsed_time(e1) / 50 * 1e3
    out = [f"MVM {mvm:.1f} us"]
    for graph in (False, True):
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            x, info = cg.cg(A, b, reltol=1e-30, maxiter=2000 if graph else 200, graph=graph)   # (the graph's capture is a one-off of a few ms)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
        out.append(f"{'graph' if graph else 'loop'}: {el / max(info['iterations'], 1) * 1e6:.1f} us / iteration ({info['iterations']} its)")
    print(f"n={n} {str(dt)[6:]}: " + "  ".join(out), flush=True)
