"""Device Levinson / Durbin / Trench timings (src/toeplitz.jl direct solvers, csrc/toeplitz_direct.hip) beside the PCG-over-FFT solve."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg
import covgram_oracle as o
for n in (1024, 4096, 16384):
    T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n))
    b = torch.randn(n, dtype=torch.float64, device="cuda")
    for _ in range(2): x = cg.levinson(T, b)
    torch.cuda.synchronize(); t0 = time.perf_counter(); x = cg.levinson(T, b); torch.cuda.synchronize(); tl = time.perf_counter() - t0
    xp, info = cg.toeplitz_solve(T, b); torch.cuda.synchronize()     # (first call: rocFFT plans)
    t0 = time.perf_counter(); xp, info = cg.toeplitz_solve(T, b); torch.cuda.synchronize(); tp = time.perf_counter() - t0
    t0 = time.perf_counter(); y = cg.durbin((T.vc[1:] / T.vc[0]).contiguous()); torch.cuda.synchronize(); td = time.perf_counter() - t0
    res = float(torch.linalg.vector_norm(T @ x - b) / torch.linalg.vector_norm(b))
    line = f"n={n}: levinson {tl * 1e3:.2f} ms (residual {res:.1e}), durbin {td * 1e3:.2f} ms, PCG over FFT MVM {tp * 1e3:.2f} ms ({info['iterations']} it)"
    if n <= 4096:
        t0 = time.perf_counter(); B = cg.trench(T); torch.cuda.synchronize(); tt = time.perf_counter() - t0
        line += f", trench {tt * 1e3:.2f} ms (|B T - I| {float(torch.linalg.matrix_norm(B @ T.to_dense() - torch.eye(n, dtype=torch.float64, device='cuda'))):.1e})"
    print(line, flush=True)
