"""C4: time per MVM against the column split (option jsplit) — whole rounds of workgroups against ragged ones."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
import sys as _s
SHAPES = ((2048, 32, cg.EQ()), (4096, 32, cg.EQ()), (8192, 32, cg.EQ()), (4096, 8, cg.EQ()), (8192, 8, cg.EQ()), (16384, 8, cg.EQ()), (8192, 3, cg.EQ())) if len(_s.argv) > 1 and _s.argv[1] == 'small' else ((16384, 32, cg.EQ()), (16384, 32, cg.MaternP(2)), (16384, 32, cg.RQ(1.5)), (16384, 48, cg.EQ()), (20000, 32, cg.EQ()), (65536, 3, cg.EQ()), (32768, 8, cg.EQ()))
for (n, d, kern) in SHAPES:
    rng = np.random.default_rng(0xC0F + 3)
    X = torch.from_numpy(rng.standard_normal((n, d))).cuda(); a = torch.from_numpy(rng.standard_normal(n * d)).cuda()
    K = cg.gramian(cg.GradientKernel(kern), X); y = torch.empty_like(a)
    res = {}
    for rep in range(3):
        for js in (0, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128):
            cg.set_option("jsplit", js)
            for _ in range(2): K.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): K.mul_(y, a)
            e1.record(); e1.synchronize(); res.setdefault(js, []).append(e0.elapsed_time(e1) / 20)
    cg.set_option("jsplit", 0)
    print(f"{type(kern).__name__[:6]} n={n} d={d}: " + "  ".join(f"{js}:{np.median(v):.3f}" for js, v in res.items()), flush=True)
