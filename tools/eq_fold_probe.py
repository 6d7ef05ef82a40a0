"""Times the fp32 EQ matrix-core kernels at the BASELINE shapes (C2 general / symmetric, C3 row shard / symmetric partial,
d = 16 / 32) and checks sampled rows against the fp64 oracle — the A/B of the norm-in-exponent form against round 1's numbers
(profiles/r01_all_configs.jsonl).   python tools/eq_fold_probe.py > profiles/r02_eq_fold_probe.txt"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg
import covgram_oracle as o


def timeit(fn, warm=3, reps=10, inner=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        for _ in range(inner): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / inner)
    return float(np.median(ts)), float(np.min(ts))


def rel(b, ref):
    return float(np.linalg.norm(np.asarray(b, np.float64) - ref) / np.linalg.norm(ref))


def case(tag, n, m, d, sym, seed, rows=256):
    rng = np.random.default_rng(seed)
    Yh = rng.standard_normal((m, d)).astype(np.float32); ah = rng.standard_normal(m).astype(np.float32)
    Y = torch.from_numpy(Yh).cuda(); a = torch.from_numpy(ah).cuda()
    G = cg.gramian(cg.EQ(), Y) if n == m else cg.gramian(cg.EQ(), Y[:n], Y)
    y = torch.empty(n, dtype=torch.float32, device="cuda")
    cg.set_option("mfma_sym", sym)
    med, mn = timeit(lambda: G.mul_(y, a))
    path = (cg.get_info("last_dense_path"), cg.get_info("last_mfma_sym"), cg.get_info("last_mfma_lds"))
    r = np.random.default_rng(1).choice(n, rows, replace=False)
    err = rel(y.cpu().numpy()[r], o.mul(None, o.Kernel(o.EQ), Yh[r], Yh, ah, dtype=np.float32))
    print(f"{tag}: n={n} m={m} d={d} sym={sym} path={path} median {med:.4f} ms min {mn:.4f} ms rel-err {err:.2e}", flush=True)
    cg.set_option("mfma_sym", -1)


print(torch.cuda.get_device_name(0))
case("C2 general", 131072, 131072, 3, 0, 0xC0F + 1)
case("C2 symmetric", 131072, 131072, 3, 1, 0xC0F + 1)
case("C3 row shard", 65536, 524288, 8, 0, 0xC0F + 2, rows=128)
case("d=8 symmetric", 200000, 200000, 8, 1, 5, rows=128)
case("d=8 general", 65536, 65536, 8, 0, 5)
case("d=2 general", 131072, 131072, 2, 0, 6)
case("d=4 general", 131072, 131072, 4, 0, 6)
case("d=6 general", 65536, 65536, 6, 0, 6)
case("d=16 general", 65536, 65536, 16, 0, 7)
case("d=16 symmetric", 65536, 65536, 16, 1, 7)
case("d=32 general", 32768, 32768, 32, 0, 8)
case("d=31 symmetric", 32768, 32768, 31, 1, 8)
# C3 as symmetric partial: rank 0 and rank 7 of 8
rng = np.random.default_rng(0xC0F + 2)
n3 = 524288
X = torch.from_numpy(rng.standard_normal((n3, 8)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n3).astype(np.float32)).cuda()
Gf = cg.gramian(cg.EQ(), X); part = torch.empty(n3, dtype=torch.float32, device="cuda")
for r in (0, 7):
    med, mn = timeit(lambda: Gf.sym_partial_(part, a, r, 8), warm=2, reps=5, inner=2)
    print(f"C3 symmetric partial rank {r} of 8: median {med:.4f} ms min {mn:.4f} ms", flush=True)
