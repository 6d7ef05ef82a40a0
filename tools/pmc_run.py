"""Run K launches of one hot-path config (for rocprofv3 --pmc / --kernel-trace passes).
usage: pmc_run.py dense|densegen|shard8|grad|toeplitz|toeplitz4|toeplitz32|c1|f64sym|f64all|f64eq3|kron64|kron32|lowrank [K]     (dense: the library's default = symmetric kernel; densegen: all n*m entries)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
which = sys.argv[1]; K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(0xC0F + 1)
if which == "dense":
    n = 131072
    X = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    G = cg.gramian(cg.EQ(), X); y = torch.empty(n, dtype=torch.float32, device="cuda")
    for _ in range(K): G.mul_(y, a)
elif which == "densegen":
    n = 131072
    X = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    cg.set_option("mfma_sym", 0)
    G = cg.gramian(cg.EQ(), X); y = torch.empty(n, dtype=torch.float32, device="cuda")
    for _ in range(K): G.mul_(y, a)
elif which in ("matern", "materngen", "f2", "f2gen", "maternshard"):   # round 5: packed-profile matrix-core kernels and the one-pass Sum at the contract size
    n = 131072
    Xh = rng.standard_normal((n, 3)).astype(np.float32)
    X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    k = cg.MaternP(2) if which.startswith("matern") else 1.5 * cg.Lengthscale(cg.MaternP(2), 0.7) + 0.5 * cg.Lengthscale(cg.EQ(), 2.0)
    if which.endswith("gen"): cg.set_option("mfma_sym", 0); cg.set_option("dense_variant", 2)
    if which == "maternshard":
        cg.set_option("dense_variant", 2)
        G = cg.gramian(k, X[:16384].contiguous(), X); y = torch.empty(16384, dtype=torch.float32, device="cuda")
    else:
        G = cg.gramian(k, X); y = torch.empty(n, dtype=torch.float32, device="cuda")
    for _ in range(K): G.mul_(y, a)
elif which in ("c3sym", "c2sym"):                # the symmetric EQ kernel: rank 3 of 8's cyclic panels of C3 (d = 8, n = 524288) / all of C2's triangle
    n, d = (524288, 8) if which == "c3sym" else (131072, 3)
    X = torch.from_numpy(np.random.default_rng(0xC0F + 2).standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    G = cg.gramian(cg.EQ(), X); y = torch.empty(n, dtype=torch.float32, device="cuda")
    for _ in range(K):
        if which == "c3sym": G.sym_partial_(y, a, 3, 8)
        else: G.mul_(y, a)
elif which == "shard8":
    n, per = 524288, 65536
    X = torch.from_numpy(np.random.default_rng(0xC0F + 2).standard_normal((n, 8)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    G = cg.gramian(cg.EQ(), X[:per], X); y = torch.empty(per, dtype=torch.float32, device="cuda")
    for _ in range(K): G.mul_(y, a)
elif which == "grad":
    n, d = 16384, 32
    X = torch.from_numpy(rng.standard_normal((n, d))).cuda(); a = torch.from_numpy(rng.standard_normal(n * d)).cuda()
    G = cg.gramian(cg.GradientKernel(cg.EQ()), X); y = torch.empty(n * d, dtype=torch.float64, device="cuda")
    for _ in range(K): G.mul_(y, a)
elif which == "grad32":                          # the C4 shape in fp32: GradientKernel(EQ), d = 32, n = 16384 (the scalar-stream lane-per-row kernel)
    n, d = 16384, 32
    X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n * d).astype(np.float32)).cuda()
    G = cg.gramian(cg.GradientKernel(cg.EQ()), X); y = torch.empty(n * d, dtype=torch.float32, device="cuda")
    for _ in range(K): G.mul_(y, a)
elif which == "c1":
    n = 4096
    X = torch.from_numpy(rng.standard_normal((n, 3))).cuda(); a = torch.from_numpy(rng.standard_normal(n)).cuda()
    G = cg.gramian(cg.MaternP(2), X); y = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(K): G.mul_(y, a)
elif which in ("f64sym", "f64all"):             # fp64 gramian(k, x) * a, MaternP(2), d = 8, n = 32768: symmetric / all-entries direct-difference kernel
    n, d = 32768, 8
    X = torch.from_numpy(np.random.default_rng(5).standard_normal((n, d)) * 0.3).cuda(); a = torch.from_numpy(rng.standard_normal(n)).cuda()
    cg.set_option("dense_sym", 1 if which == "f64sym" else 0)
    G = cg.gramian(cg.MaternP(2), X); y = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(K): G.mul_(y, a)
elif which == "f64eq3":                          # fp64 EQ, d = 3, n = 32768 against a second point set: the general lane-per-row kernel
    n = 32768
    X = torch.from_numpy(np.random.default_rng(5).standard_normal((n, 3))).cuda(); Y = torch.from_numpy(np.random.default_rng(6).standard_normal((n, 3))).cuda()
    a = torch.from_numpy(rng.standard_normal(n)).cuda()
    G = cg.gramian(cg.EQ(), X, Y); y = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(K): G.mul_(y, a)
elif which in ("kron64", "kron32"):             # README.md:205-210: 128^3 grid, three 128 x 128 factors
    dt = torch.float64 if which == "kron64" else torch.float32
    ax = torch.linspace(0, 1, 128, dtype=dt, device="cuda")
    G = cg.gramian(cg.separable("*", cg.Exp(), cg.Exp(), cg.Exp()), cg.LazyGrid(ax, 3))
    a = torch.randn(128 ** 3, dtype=dt, device="cuda"); y = torch.empty_like(a)
    for _ in range(K): G.mul_(y, a)
elif which == "lowrank":                          # FiniteBasis (r = 32) on n = 2^20 points, fp32: U (U' a)
    nl, r = 1 << 20, 32
    xs = torch.randn(nl, dtype=torch.float32, device="cuda")
    Gl = cg.gramian(cg.FiniteBasis([lambda t, i=i: torch.cos(0.37 * i * t) for i in range(r)]), xs)
    al = torch.randn(nl, dtype=torch.float32, device="cuda"); yl = torch.empty_like(al)
    for _ in range(K): Gl.mul_(yl, al)
elif which == "toeplitz32":
    n = 1 << 22
    T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n, dtype=torch.float32)); a = torch.randn(n, dtype=torch.float32, device="cuda"); y = torch.empty_like(a)
    for _ in range(K): T.mul_(y, a)
elif which in ("toeplitz", "toeplitz4"):          # toeplitz4: the radix-4 fused row kernel (round 1) instead of the radix-16 one
    if which == "toeplitz4": cg.set_option("toeplitz_fused", 2)
    n = 1 << 22
    T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n)); a = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(a)
    for _ in range(K): T.mul_(y, a)
torch.cuda.synchronize()
print("done", which, K)
