"""A few README rows re-measured (fp32 gramian(k, x) * a at n = 131072, d = 3 by profile; EQ at d = 31, n = 32768): us per MVM back to back."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
def timed(fn, reps):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
n, d = 131072, 3
X = torch.from_numpy(np.random.default_rng(1).standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.randn(n, dtype=torch.float32, device="cuda"); y = torch.empty_like(a)
for name, k in (("EQ", cg.EQ()), ("RQ(1.5)", cg.RQ(1.5)), ("Cauchy", cg.Cauchy()), ("Dot^2", cg.Dot() ** 2), ("MaternP(2)", cg.MaternP(2)), ("Exp", cg.Exp()), ("GammaExp(1.5)", cg.GammaExp(1.5)), ("EQ l=0.05", cg.Lengthscale(cg.EQ(), 0.05))):
    G = cg.gramian(k, X); t = np.median([timed(lambda: G.mul_(y, a), 5) for _ in range(3)])
    print(f"gramian({name}, x) n={n} d={d}: {t:8.1f} us  (path {cg.get_info('last_dense_path')}, mfma_sym {cg.get_info('last_mfma_sym')}, dense_sym {cg.get_info('last_dense_sym')}, fp16 {cg.get_info('last_mfma_f16')})", flush=True)
n, d = 32768, 31
X = torch.from_numpy((np.random.default_rng(2).standard_normal((n, d)) * 0.6).astype(np.float32)).cuda(); a = torch.randn(n, dtype=torch.float32, device="cuda"); y = torch.empty_like(a)
G = cg.gramian(cg.EQ(), X); t = np.median([timed(lambda: G.mul_(y, a), 10) for _ in range(3)])
print(f"gramian(EQ, x) n={n} d={d} (x ~ 0.6 N(0, I)): {t:8.1f} us (mfma_sym {cg.get_info('last_mfma_sym')}, fp16 {cg.get_info('last_mfma_f16')})")
cg.set_option("mfma_sym", 0); t = np.median([timed(lambda: G.mul_(y, a), 10) for _ in range(3)]); cg.set_option("mfma_sym", -1)
print(f"   all entries: {t:8.1f} us (fp16 {cg.get_info('last_mfma_f16')})")
for p in (32, 64, 128):
    n, d = 32768, 3
    X = torch.from_numpy(np.random.default_rng(3).standard_normal((n, d)).astype(np.float32)).cuda(); A = torch.randn(n, p, dtype=torch.float32, device="cuda")
    G = cg.gramian(cg.EQ(), X); t = np.median([timed(lambda: G @ A, 5) for _ in range(3)])
    print(f"gramian(EQ, x) * A, n={n} d={d} p={p}: {t:8.1f} us")
