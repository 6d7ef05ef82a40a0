"""fp64 dense Gramian MVM by point dimension (lane-per-row direct-difference kernel, all entries and the symmetric form): us per MVM and
fp64 issue slots per pair (1024 SIMDs x 2.4 GHz / 4 cycles = 6.14e11 DP instructions per second = the chip's DP issue rate)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def t_us(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for name, k in (("EQ", cg.EQ()), ("MaternP(2)", cg.MaternP(2))):
    for d in (3, 8, 16, 24, 32, 48, 64):
        rng = np.random.default_rng(d)
        X = torch.from_numpy(rng.standard_normal((n, d)) / np.sqrt(d) * 1.5).cuda(); a = torch.from_numpy(rng.standard_normal(n)).cuda(); y = torch.empty_like(a)
        G = cg.gramian(k, X)
        out = []
        for opt in (0, 1, 2, 3):
            cg.set_option("dense_sym", 1 if opt in (1, 3) else 0)
            cg.set_option("dense_bcast", 1 if opt >= 2 else 0)
            us = t_us(lambda: G.mul_(y, a))
            ev = n * (n + 64) / 2 if opt in (1, 3) else n * n
            used = cg.get_info("last_dense_bcast")
            out.append(f"{('all entries', 'symmetric', 'broadcast' if used else 'broadcast n/a', 'sym + broadcast' if used else 'sym (broadcast n/a)')[opt]} {us:8.1f} us = {us * 1e-6 * 6.144e11 * 64 / ev:6.1f} slots per evaluated pair")
        print(f"{name:10s} d={d:2d} n={n}: " + " | ".join(out), flush=True)
cg.set_option("dense_sym", -1); cg.set_option("dense_bcast", -1)
