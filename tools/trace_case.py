"""One named case, a few calls (for rocprofv3 --kernel-trace --stats): python tools/trace_case.py <c1|c5|c5f32|kron128|dotfac|c4|readme16k|toep64k|cg131k|matrix16k|valgrad|c4f32|circ1m> [reps]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
case = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(0)
if case == "c1":
    X = torch.from_numpy(rng.standard_normal((4096, 3))).cuda(); a = torch.from_numpy(rng.standard_normal(4096)).cuda(); G = cg.gramian(cg.MaternP(2), X)
elif case == "readme16k":
    X = torch.from_numpy(rng.standard_normal((16384, 3))).cuda(); a = torch.from_numpy(rng.standard_normal(16384)).cuda(); G = cg.gramian(cg.MaternP(2), X)
elif case in ("c5", "c5f32"):
    dt = torch.float64 if case == "c5" else torch.float32      # a RANGE (cg.srange), not a tensor of its values: only a range is a Toeplitz Gramian (src/gramian.jl:167-189)
    G = cg.gramian(cg.Exp(), cg.srange(-1, 1, 1 << 22)); a = torch.randn(1 << 22, dtype=dt, device="cuda")
elif case == "kron128":
    ax = torch.linspace(0, 1, 128, dtype=torch.float64, device="cuda"); G = cg.gramian(cg.separable("*", cg.Exp(), cg.Exp(), cg.Exp()), cg.LazyGrid(ax, 3))
    a = torch.randn(128 ** 3, dtype=torch.float64, device="cuda")
elif case == "dotfac":
    X = torch.from_numpy(rng.standard_normal((1 << 20, 8)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(1 << 20).astype(np.float32)).cuda(); G = cg.gramian(cg.Dot(), X)
elif case == "c4":
    X = torch.from_numpy(rng.standard_normal((16384, 32))).cuda(); a = torch.from_numpy(rng.standard_normal(16384 * 32)).cuda(); G = cg.gramian(cg.GradientKernel(cg.EQ()), X)
elif case == "wide64":
    X = torch.from_numpy(rng.standard_normal((16384, 32)) / 4).cuda(); a = torch.from_numpy(rng.standard_normal(16384)).cuda(); G = cg.gramian(cg.EQ(), X)
elif case == "mp32k":
    X = torch.from_numpy(rng.standard_normal((32768, 3))).cuda(); a = torch.from_numpy(rng.standard_normal(32768)).cuda(); G = cg.gramian(cg.MaternP(2), X)
elif case == "toep64k":
    G = cg.gramian(cg.EQ(), cg.srange(-1, 1, 65536)); a = torch.randn(65536, dtype=torch.float64, device="cuda")
elif case == "circ1m":
    G = cg.gramian(cg.Cosine(2.0) if hasattr(cg, "Cosine") else cg.EQ(), cg.srange(0, 1, 1 << 20)); a = torch.randn(1 << 20, dtype=torch.float64, device="cuda")
elif case == "valgrad":
    X = torch.from_numpy(rng.standard_normal((16384, 32))).cuda(); a = torch.from_numpy(rng.standard_normal(16384 * 33)).cuda(); G = cg.gramian(cg.ValueGradientKernel(cg.EQ()), X)
elif case == "c4f32":
    X = torch.from_numpy(rng.standard_normal((16384, 32)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(16384 * 32).astype(np.float32)).cuda(); G = cg.gramian(cg.GradientKernel(cg.EQ()), X)
elif case == "matrix16k":
    X = torch.from_numpy(rng.standard_normal((16384, 3)).astype(np.float32)).cuda(); G = cg.gramian(cg.EQ(), X)
    for _ in range(reps): M = G.to_dense()
    torch.cuda.synchronize(); sys.exit(0)
elif case == "cg131k":
    X = torch.from_numpy(rng.standard_normal((131072, 3)).astype(np.float32)).cuda(); b = torch.from_numpy(rng.standard_normal(131072).astype(np.float32)).cuda()
    from covgram import solve
    A = cg.gramian(cg.EQ(), X) + torch.full((131072,), 0.1, dtype=torch.float32, device="cuda")     # G + sigma^2 I (src/gramian.jl:55-60)
    x, info = solve.cg(A, b, reltol=1e-30, maxiter=reps)
    torch.cuda.synchronize(); print(info); sys.exit(0)
y = torch.empty_like(a)
for _ in range(reps): G.mul_(y, a)
torch.cuda.synchronize()
