"""fp32 Levinson / Durbin on the device against the fp64 chain: residual and error by size (tolerances of tests/test_gpu_parity.py)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg
import covgram_oracle as o
def relerr(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
for m in (65, 257, 511, 513, 1025, 2049, 4097, 16384):
    rng = np.random.default_rng(4100 + m)
    xs = np.linspace(-1.0, 1.0, m); vc = np.exp(-np.abs(xs - xs[0])); r, b = vc[1:].copy(), rng.standard_normal(m)
    rd, bd = torch.from_numpy(r).cuda(), torch.from_numpy(b).cuda()
    x32 = cg.levinson(rd.float(), bd.float())
    T = cg.SymmetricToeplitz(torch.from_numpy(vc).cuda())
    x64 = cg.levinson(rd, bd)
    res32 = relerr((T @ x32.double()).cpu().numpy(), b)
    err32 = relerr(x32.double().cpu().numpy(), x64.cpu().numpy())
    # numpy float32 Levinson (serial order) for comparison
    if m <= 4097:
        r32, b32 = r.astype(np.float32), b.astype(np.float32)
        xo = o.levinson(r32, b32)
        print(m, "gpu32 res %.2e err %.2e | oracle(f32 inputs) dtype %s err %.2e" % (res32, err32, xo.dtype, relerr(xo.astype(np.float64), x64.cpu().numpy())), flush=True)
    else:
        print(m, "gpu32 res %.2e err %.2e" % (res32, err32), flush=True)
    y32 = cg.durbin(rd.float()).double().cpu().numpy(); y64 = cg.durbin(rd).cpu().numpy()
    print("   durbin32 err %.2e" % relerr(y32, y64))
    rj = torch.from_numpy(r / 1.5).cuda()                            # jittered diagonal (1.5 on a unit kernel): well conditioned at every size
    xj32, xj64 = cg.levinson(rj.float(), bd.float()), cg.levinson(rj, bd)
    yj32, yj64 = cg.durbin(rj.float()), cg.durbin(rj)
    print("   jittered: levinson32 err %.2e, durbin32 err %.2e" % (relerr(xj32.double().cpu().numpy(), xj64.cpu().numpy()), relerr(yj32.double().cpu().numpy(), yj64.cpu().numpy())))
