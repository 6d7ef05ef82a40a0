"""First-light timing of the dense EQ MVM (BASELINE config 2) with option sweeps.  Dev tool, not the contract bench."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg, covgram_oracle as o

def timeit(fn, warm=3, reps=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))

def dense(n, d, dtype, kern=None, label="EQ"):
    rng = np.random.default_rng(0xC0F + 1)
    X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32 if dtype == torch.float32 else np.float64)).cuda()
    a = torch.from_numpy(rng.standard_normal(n).astype(np.float32 if dtype == torch.float32 else np.float64)).cuda()
    G = cg.gramian(kern or cg.EQ(), X)
    y = torch.empty(n, dtype=dtype, device="cuda")
    med, mn = timeit(lambda: G.mul_(y, a))
    pairs = n * n / (mn * 1e-3)
    print(f"{label} n={n} d={d} {dtype}: median {med:.3f} ms  min {mn:.3f} ms  -> {1e3/mn:.1f} MVM/s  {pairs*1e-12:.3f} Tpairs/s", flush=True)
    return G, X, a, y

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    print(torch.cuda.get_device_name(0))
    n = 131072
    for tw in (0, 16384, 65536):
        cg.set_option("target_wgs", tw)
        print(f"target_wgs={tw}: ", end="")
        G, X, a, y = dense(n, 3, torch.float32)
    cg.set_option("target_wgs", 0)
    # accuracy on a row subset against the fp64 oracle
    G, X, a, y = dense(n, 3, torch.float32)
    rows = np.random.default_rng(1).choice(n, 2048, replace=False)
    ref = o.mul(None, o.Kernel(o.EQ), X.cpu().numpy()[rows], X.cpu().numpy(), a.cpu().numpy(), dtype=np.float32)
    b = y.cpu().numpy()[rows]
    print("C2 rel-err vs fp64 oracle (2048 rows):", np.linalg.norm(b - ref) / np.linalg.norm(ref))
    dense(524288 // 4, 8, torch.float32, label="EQ(C3 shard-sized n/4)")
    dense(16384, 3, torch.float64, cg.MaternP(2), "MaternP2 f64")
    dense(16384, 32, torch.float32, label="EQ d32 f32")
    # gradient config 4
    rng = np.random.default_rng(0xC0F + 3)
    ng, dg = 16384, 32
    Xg = torch.from_numpy(rng.standard_normal((ng, dg))).cuda(); ag = torch.from_numpy(rng.standard_normal(ng * dg)).cuda()
    K = cg.gramian(cg.GradientKernel(cg.EQ()), Xg); yg = torch.empty(ng * dg, dtype=torch.float64, device="cuda")
    med, mn = timeit(lambda: K.mul_(yg, ag), warm=2, reps=5)
    print(f"GradientKernel(EQ) n={ng} d={dg} f64: median {med:.3f} ms min {mn:.3f} ms -> {1e3/mn:.2f} MVM/s, {ng*ng/(mn*1e-3)*1e-9:.2f} Gblocks/s")
    Xg32 = Xg.float(); ag32 = ag.float(); K32 = cg.gramian(cg.GradientKernel(cg.EQ()), Xg32); yg32 = torch.empty(ng * dg, dtype=torch.float32, device="cuda")
    med, mn = timeit(lambda: K32.mul_(yg32, ag32), warm=2, reps=5)
    print(f"GradientKernel(EQ) n={ng} d={dg} f32: median {med:.3f} ms min {mn:.3f} ms")
    # toeplitz config 5
    nt = 1 << 22
    T = cg.gramian(cg.Exp(), cg.srange(-1, 1, nt)); at = torch.randn(nt, dtype=torch.float64, device="cuda"); yt = torch.empty_like(at)
    med, mn = timeit(lambda: T.mul_(yt, at))
    print(f"Toeplitz Exp n=2^22 f64: median {med:.3f} ms min {mn:.3f} ms -> {1e3/mn:.1f} MVM/s; ideal 470MB -> {470e6/(mn*1e-3)*1e-12:.2f} TB/s algorithmic")
    # wide-d dense (d > 64): README-like EQ d=1024, n=16384 would be 2.7e8 pairs x 2048 lane-ops
    for (nw, dw, dt) in ((16384, 128, torch.float32), (8192, 1024, torch.float32), (8192, 1024, torch.float64)):
        rngw = np.random.default_rng(5)
        Xw = torch.from_numpy(rngw.standard_normal((nw, dw)) / np.sqrt(dw)).to(dt).cuda(); aw = torch.randn(nw, dtype=dt, device="cuda")
        Gw = cg.gramian(cg.EQ(), Xw); yw = torch.empty(nw, dtype=dt, device="cuda")
        med, mn = timeit(lambda: Gw.mul_(yw, aw), warm=2, reps=5)
        print(f"wide EQ n={nw} d={dw} {dt}: median {med:.3f} ms min {mn:.3f} ms -> {nw*nw*2.0*dw/(mn*1e-3)*1e-12:.2f} T lane-ops/s")
    # wide-d gradient: the README case (README.md:231-245): MaternP(2), d = 1024, n = 1024, fp64 (reference: 0.394 s)
    for (ng, dg, dt) in ((1024, 1024, torch.float64), (1024, 1024, torch.float32), (4096, 256, torch.float64)):
        rngw = np.random.default_rng(6)
        Xw = torch.from_numpy(rngw.standard_normal((ng, dg)) / np.sqrt(dg)).to(dt).cuda(); aw = torch.randn(ng * dg, dtype=dt, device="cuda")
        Kw = cg.gramian(cg.GradientKernel(cg.MaternP(2)), Xw); yw = torch.empty(ng * dg, dtype=dt, device="cuda")
        med, mn = timeit(lambda: Kw.mul_(yw, aw), warm=2, reps=5)
        print(f"wide grad MaternP(2) n={ng} d={dg} {dt}: median {med:.3f} ms min {mn:.3f} ms -> {ng*ng*6.0*dg/(mn*1e-3)*1e-12:.2f} T lane-ops/s")
