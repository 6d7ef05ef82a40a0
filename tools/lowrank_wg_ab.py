"""Low-rank MVM: row slabs per CU of the first pass (option lowrank_wgs) after the round-5 four-group trips; us per MVM back to back."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for dt in (torch.float32, torch.float64):
    for nl, r in ((1 << 20, 32), (1 << 20, 16), (1 << 20, 8), (1 << 19, 64), (1 << 17, 32)):
        xs = torch.randn(nl, dtype=dt, device="cuda")
        G = cg.gramian(cg.FiniteBasis([lambda t, i=i: torch.cos(0.37 * i * t) for i in range(r)]), xs)
        a = torch.randn(nl, dtype=dt, device="cuda"); y = torch.empty_like(a)
        out = []
        for w in (1, 2, 3, 4, 2):
            cg.set_option("lowrank_wgs", w)
            for _ in range(5): G.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): G.mul_(y, a)
            e1.record(); e1.synchronize(); out.append(f"wgs={w}: {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us")
        print(f"{str(dt)[6:]} n={nl} r={r} ({nl * r * a.element_size() / 2**20:.0f} MiB): " + " | ".join(out), flush=True)
cg.set_option("lowrank_wgs", 0)
