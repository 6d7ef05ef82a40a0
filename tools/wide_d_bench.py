"""fp32 EQ at wide points (d = 12 .. 32) on the matrix cores: us per MVM for gramian(k, x) (symmetric kernels) and a 16384-row shard (general kernel), the MFMAs
per tile, and the matrix-pipe floor (32 cycles per v_mfma_f32_32x32x16 at the measured ~2.2 GHz)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=10):
    ts = []
    for rep in range(3):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts)) * 1e3
for n in (32768, 131072):
    for d in (8, 12, 16, 24, 32):
        rng = np.random.default_rng(d)
        X = torch.from_numpy((rng.standard_normal((n, d)) / np.sqrt(d) * 2.0).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
        k = cg.Lengthscale(cg.EQ(), 1.0)
        G = cg.gramian(k, X); y = torch.empty_like(a)
        ts = timeit(lambda: G.mul_(y, a)); inst = cg.get_info("last_mfma_instance"); f16 = cg.get_info("last_mfma_f16"); path = (cg.get_info("last_dense_path"), cg.get_info("last_mfma_sym"))
        per = 16384
        Gs = cg.gramian(k, X[:per].contiguous(), X); ys = torch.empty(per, dtype=torch.float32, device="cuda")
        tg = timeit(lambda: Gs.mul_(ys, a)); instg = cg.get_info("last_mfma_instance")
        k2 = (d + 3) // 4 if f16 else (d + 1) // 2
        floor_s = (n * n / 2 / 1024) * k2 * 32 / 1024 / 2.2e9 * 1e6; floor_g = (per * n / 1024) * k2 * 32 / 1024 / 2.2e9 * 1e6
        print(f"n={n} d={d:2d} f16={f16} K2~{k2}: symmetric {ts:8.1f} us (instance {inst}, path {path}, MFMA floor {floor_s:6.1f}) | shard {per} rows {tg:7.1f} us (instance {instg}, floor {floor_g:6.1f})", flush=True)
