"""fp32 gramian(k, x) * a for the generic matrix-core profiles: the symmetric matrix-core kernel (dense_mfma_sym_kernel<FAM>: distance on the matrix pipe,
profile per entry in scalar fp32) against the symmetric direct-difference kernel of round 4 (dense_sym32_kernel: packed fp32 math, option dense_variant = 1)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def t_us(fn, reps=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
for n in (32768, 131072):
    for d in (3, 8, 16):
        rng = np.random.default_rng(d)
        X = torch.from_numpy((rng.standard_normal((n, d)) * (1.0 if d == 3 else 0.6)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda(); y = torch.empty_like(a)
        for name, k in (("MaternP(2)", cg.MaternP(2)), ("MaternP(1)", cg.MaternP(1)), ("RQ(1.5)", cg.RQ(1.5)), ("Cauchy", cg.Cauchy()), ("IMQ", cg.InverseMultiQuadratic(1.0)), ("EQ", cg.EQ()),
                        ("MaternP(2)*EQ", cg.MaternP(2) * cg.EQ())):
            G = cg.gramian(k, X)
            cg.set_option("dense_variant", 0); t0 = t_us(lambda: G.mul_(y, a)); p0 = (cg.get_info("last_dense_path"), cg.get_info("last_mfma_sym"), cg.get_info("last_dense_sym")); y0 = y.clone()
            cg.set_option("dense_variant", 1); t1 = t_us(lambda: G.mul_(y, a)); p1 = (cg.get_info("last_dense_path"), cg.get_info("last_mfma_sym"), cg.get_info("last_dense_sym"))
            cg.set_option("dense_variant", 2); t2 = t_us(lambda: G.mul_(y, a)); p2 = (cg.get_info("last_dense_path"), cg.get_info("last_mfma_sym"), cg.get_info("last_dense_sym")); d2 = float((y-y0).norm()/y0.norm())
            print(f"n={n} d={d:2d} {name:14s}: default (path, mfma_sym, dense_sym)={p0} {t0:8.1f} us | direct differences {p1} {t1:8.1f} us | matrix cores forced {p2} {t2:8.1f} us (diff {d2:.1e})", flush=True)
# row shard (two point sets: no symmetric form): matrix cores (dense_variant = 2) against the lane-per-row kernel
for n in (131072,):
    for d in (3, 8):
        rng = np.random.default_rng(d)
        X = torch.from_numpy((rng.standard_normal((n, d)) * (1.0 if d == 3 else 0.6)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
        y = torch.empty(n // 8, dtype=torch.float32, device="cuda")
        for name, k in (("MaternP(2)", cg.MaternP(2)), ("MaternP(1)", cg.MaternP(1))):
            G = cg.gramian(k, X[: n // 8], X)
            cg.set_option("dense_variant", 2); t0 = t_us(lambda: G.mul_(y, a)); p0 = cg.get_info("last_dense_path")
            cg.set_option("dense_variant", 1); t1 = t_us(lambda: G.mul_(y, a)); p1 = cg.get_info("last_dense_path")
            cg.set_option("dense_variant", 0); t2 = t_us(lambda: G.mul_(y, a)); p2 = cg.get_info("last_dense_path")
            print(f"shard {n // 8} x {n} d={d} {name}: matrix cores (path {p0}) {t0:.1f} us | lane-per-row (path {p1}) {t1:.1f} us | default (path {p2}) {t2:.1f} us", flush=True)
cg.set_option("dense_variant", 0)
