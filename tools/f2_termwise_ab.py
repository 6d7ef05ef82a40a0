import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo/covariancefunctions.jl_amd")
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def t_us(fn, reps=6):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
n, d = 131072, 3
X = torch.from_numpy(np.random.default_rng(1).standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.randn(n, dtype=torch.float32, device="cuda"); y = torch.empty_like(a)
kc = 1.5 * cg.Lengthscale(cg.MaternP(2), 0.7) + 0.5 * cg.Lengthscale(cg.EQ(), 2.0)
G = cg.gramian(kc, X)
for tw in (1, 0, -1):
    cg.set_option("composite_termwise", tw)
    t = t_us(lambda: G.mul_(y, a)); print("composite_termwise", tw, f"{t:.1f} us", cg.get_info("last_dense_path"), cg.get_info("last_mfma_sym"), cg.get_info("last_dense_sym")); 
    if tw == 1: y1 = y.clone()
    else: print("   diff vs termwise", float((y - y1).norm() / y1.norm()))
