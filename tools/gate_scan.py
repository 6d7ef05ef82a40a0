"""Row-wise error of the fp32 matrix-core kernels against the fp64 oracle as a function of the radius-gate quantity, on the ADVERSARIAL cloud of
tests/test_gpu_gen_f16.py (+- pairs, half the points exactly on the sphere of the gate radius — isolated rows whose diagonal entry dominates):
for each profile, d and split (bf16 three-way: option mfma_f16 = 0; fp16 two-way anywhere the matrix-core gate admits: 2), symmetric and general
kernels, the worst row-wise and the norm-wise error at fractions of the matrix-core gate (sensitivity x power x R^2 / l^2 over 0.5 x 126 / log2 e)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import covgram as cg, covgram_oracle as o
from test_gpu_gen_f16 import _cloud, LIMIT
n = 2048
profiles = [("EQ", cg.EQ(), o.Kernel(o.EQ), 0.5), ("RQ(1.5)", cg.RQ(1.5), o.Kernel(o.RQ, param=1.5), 0.5), ("Cauchy", cg.Cauchy(), o.Kernel(o.CAUCHY), 1.0),
            ("MaternP(2)", cg.MaternP(2), o.Kernel(o.MATERNP, p=2), 5.0 / 6.0), ("MaternP(1)", cg.MaternP(1), o.Kernel(o.MATERNP, p=1), 0.5 * 3 / 1 / 2 * 1.0)]
fracs = (0.3, 0.4, 0.5, 0.571, 0.7, 0.85, 0.995)
for d in (1, 3, 5, 8, 13):
    rng = np.random.default_rng(d)
    ah = rng.standard_normal(n).astype(np.float32); a = torch.from_numpy(ah).cuda(); ad = ah.astype(np.float64)
    for name, k, ko, c in profiles:
        for f16 in (0, 2):
            for sym in (1, 0):
                cells = []
                for fr in fracs:
                    Xh = _cloud(rng, n, d, np.sqrt(fr * LIMIT / c)); X = torch.from_numpy(Xh).cuda(); Xd = Xh.astype(np.float64)
                    ref = o.mul(None, ko, Xd, Xd, ad); absref = np.abs(o.matrix(ko, Xd, Xd)) @ np.abs(ad)
                    cg.set_option("mfma_sym", sym); cg.set_option("mfma_f16", f16)
                    y = torch.empty_like(a); cg.gramian(k, X).mul_(y, a)
                    used = cg.get_info("last_mfma_f16"); path = cg.get_info("last_dense_path")
                    b = y.cpu().numpy().astype(np.float64)
                    cells.append(f"{np.max(np.abs(b - ref) / absref) * 1e6:5.1f}/{np.linalg.norm(b - ref) / np.linalg.norm(ref) * 1e6:4.1f}{'h' if used else 'b'}{'' if path == 2 else '!'}")
                print(f"d={d:2d} {name:11s} f16={f16} sym={sym}: " + "  ".join(cells), flush=True)
print("cells: rowwise / normwise error x 1e6, h = fp16 split ran, b = bf16, ! = not a matrix-core kernel; columns = gate fraction", fracs)
cg.set_option("mfma_sym", -1); cg.set_option("mfma_f16", -1)
