"""General matrix-core EQ kernel: the bf16 three-way split (option mfma_f16 = 0) against the fp16 two-way split (= 2: wherever the matrix-core gate admits the cloud): time per MVM and error against the fp64 C oracle
(256 sample rows; norm-wise and worst row against sum_j |a_j| k_ij), on the contract shapes and on clouds scaled to the edge of the gate."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg, covgram_oracle as o, c_oracle

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def cloud(n, d, kind, rng):
    X = rng.standard_normal((n, d))
    if kind > 0:        # every point on a shell of g^2 |x|^2 in [0.5 kind, kind]: the worst case of the gate
        X = X / np.linalg.norm(X, axis=1, keepdims=True) * np.sqrt(rng.uniform(0.5 * kind, kind, (n, 1)) / np.log2(np.e))
    return X.astype(np.float32)

cases = [("C2: d=3 n=131072", 131072, 131071, 3, 0, 5), ("C3 shard: d=8 65536 x 524288", 524288, 65536, 8, 0, 5), ("d=8 16384 x 131072", 131072, 16384, 8, 0, 20),
         ("d=16 n=65536", 65536, 65535, 16, 0, 10), ("d=32 n=32768", 32768, 32767, 32, 0, 10), ("d=5 n=65536", 65536, 65535, 5, 0, 10),
         ("d=8 shell g2R2<=30", 65536, 16384, 8, 30, 10), ("d=8 shell g2R2<=60", 65536, 16384, 8, 60, 10), ("d=8 shell g2R2<=90", 65536, 16384, 8, 90, 10),
         ("d=8 shell g2R2<=110", 65536, 16384, 8, 110, 10), ("d=3 shell g2R2<=110", 65536, 16384, 3, 110, 10), ("d=3 shell g2R2<=60", 65536, 16384, 3, 60, 10), ("d=16 shell g2R2<=72", 65536, 16384, 16, 72, 10)]
only = sys.argv[1:]
for name, n, per, d, kind, reps in cases:
    if only and not any(s in name for s in only): continue
    rng = np.random.default_rng(7 + d + kind)
    Xh = cloud(n, d, kind, rng); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda()
    G = cg.gramian(cg.EQ(), X[:per].contiguous(), X); y = torch.empty(per, dtype=torch.float32, device="cuda")
    rows = np.sort(rng.choice(per, 256, replace=False))
    Xd = Xh.astype(np.float64); ad = ah.astype(np.float64)
    ref = c_oracle.mvm(o.Kernel(o.EQ), Xd[rows], Xd, ad).ravel()
    absref = c_oracle.mvm(o.Kernel(o.EQ), Xd[rows], Xd, np.abs(ad)).ravel()
    out = []
    for f16 in (0, 1):
        cg.set_option("mfma_f16", 2 * f16)
        ms = np.median([timed(lambda: G.mul_(y, a), reps) for _ in range(3)])
        assert cg.get_info("last_dense_path") == 2 and cg.get_info("last_mfma_f16") == f16, (name, cg.get_info("last_dense_path"), cg.get_info("last_mfma_f16"))
        b = y.cpu().numpy().astype(np.float64)[rows]
        out.append(f"{'fp16x2' if f16 else 'bf16x3'} {ms * 1e3:8.1f} us  norm-wise {np.linalg.norm(b - ref) / np.linalg.norm(ref):.2e}  row-wise {np.max(np.abs(b - ref) / absref):.2e}")
    cg.set_option("mfma_f16", -1)
    print(f"{name:32s}: " + "  |  ".join(out), flush=True)
