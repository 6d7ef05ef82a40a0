import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import covgram as cg, covgram_oracle as o
from quickbench import timeit
rng = np.random.default_rng(0xC0F + 3)
for (ng, dg, dt) in ((16384, 32, torch.float64), (16384, 32, torch.float32), (16384, 16, torch.float64), (16384, 24, torch.float64), (8192, 64, torch.float32), (8192, 48, torch.float64)):
    Xg = torch.from_numpy(rng.standard_normal((ng, dg))).cuda().to(dt); ag = torch.from_numpy(rng.standard_normal(ng * dg)).cuda().to(dt)
    K = cg.gramian(cg.GradientKernel(cg.EQ()), Xg); yg = torch.empty(ng * dg, dtype=dt, device="cuda")
    for keep in (-1, 0, 1):
        for tw in (0, 16384):
            cg.set_option("grad_keep_r", keep); cg.set_option("target_wgs", tw)
            med, mn = timeit(lambda: K.mul_(yg, ag), warm=2, reps=5)
            print(f"grad EQ n={ng} d={dg} {dt} keep_r={keep} target_wgs={tw}: min {mn:.3f} ms -> {ng*ng/(mn*1e-3)*1e-9:.2f} Gblocks/s, {ng*ng*5*dg/(mn*1e-3)*1e-12:.2f} T lane-ops/s", flush=True)
    cg.set_option("grad_keep_r", -1); cg.set_option("target_wgs", 0)
