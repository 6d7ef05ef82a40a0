"""Factored dot-product Gramian gramian(Dot(), x) * a = X (X^T a) (csrc/lowrank.hip: dot_vta1_kernel + dot_xz_kernel): us per MVM back to back, the
algorithmic bytes (two passes over the points + a + y) per second, and the error against torch fp64."""
import os, subprocess, sys
CASES = [(dt, n, d) for dt in ("float32", "float64") for n, d in ((1 << 20, 8), (1 << 20, 3), (1 << 20, 16), (1 << 20, 32), (1 << 20, 64), (1 << 17, 3), (1 << 22, 3), (100003, 5))]
if len(sys.argv) == 1:      # one fresh process per case (this parent never touches the GPU): a 256 MB point set allocated late in a long sweep of one process
    for dt, n, d in CASES:  # measured 10x slower than the same case alone — allocator / page state, not the kernels
        subprocess.run([sys.executable, os.path.abspath(__file__), dt, str(n), str(d)], check=False)
    sys.exit(0)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
only = sys.argv[1:] and (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
for dt in (torch.float32, torch.float64):
    for n, d in sorted({(c[1], c[2]) for c in CASES}):
        if only and (str(dt)[6:], n, d) != only: continue
        rng = np.random.default_rng(n + d)
        X = torch.from_numpy(rng.standard_normal((n, d))).to(dt).cuda(); a = torch.from_numpy(rng.standard_normal(n)).to(dt).cuda(); y = torch.empty_like(a)
        G = cg.gramian(cg.Dot(), X)
        for _ in range(5): G.mul_(y, a)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): G.mul_(y, a)
        e1.record(); e1.synchronize(); t = e0.elapsed_time(e1) / 50 * 1e-3
        ref = X.double() @ (X.double().T @ a.double())
        by = (2 * n * d + 2 * n) * a.element_size()
        print(f"{str(dt)[6:]} n={n} d={d}: {t * 1e6:7.1f} us  {by / t * 1e-12:5.2f} TB/s  rel err {float((y.double() - ref).norm() / ref.norm()):.1e}", flush=True)
