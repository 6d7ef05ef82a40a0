"""Round 4, C4: the fp64 expanded-form gradient MVM with the column records in VGPRs (grad_bcast_kernel: v_fmac_f64_dpp row_newbcast,
counted vector loads) against the scalar-stream kernel (grad_mvm_kernel<..., EXPD>), interleaved on one box.  Option "grad_bcast":
0 = scalar stream, 1 / 4 = broadcast kernel with one / four waves per workgroup.  Output kept as profiles/r04_c4_bcast_ab.txt."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
cases = [(16384, 32, cg.EQ(), "EQ (C4)"), (16384, 8, cg.EQ(), "EQ"), (16384, 16, cg.EQ(), "EQ"), (16384, 24, cg.EQ(), "EQ"), (16384, 48, cg.EQ(), "EQ"),
         (4096, 32, cg.EQ(), "EQ"), (16384, 32, cg.MaternP(2), "MaternP(2)"), (16384, 32, cg.RQ(1.5), "RQ(1.5)"), (16384, 32, cg.Cauchy(), "Cauchy"),
         (16384, 12, cg.MaternP(2), "MaternP(2)")]
if len(sys.argv) > 1: cases = cases[:int(sys.argv[1])]
for n, d, k, name in cases:
    rng = np.random.default_rng(0xC0F + 3)
    X = torch.from_numpy(rng.standard_normal((n, d)) * (1.0 if d <= 32 else 0.8)).cuda(); a = torch.from_numpy(rng.standard_normal(n * d)).cuda()
    K = cg.gramian(cg.GradientKernel(k), X)
    ys = {}
    res = {0: [], 1: [], 4: []}
    for rep in range(5):
        for v in (0, 1, 4):
            cg.set_option("grad_bcast", v)
            y = torch.empty_like(a)
            for _ in range(2): K.mul_(y, a)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): K.mul_(y, a)
            e1.record(); e1.synchronize()
            res[v].append(e0.elapsed_time(e1) / 10)
            used = cg.get_info("last_grad_bcast")
            assert used == v, (used, v)
            ys[v] = y
    dif = {v: float((ys[v] - ys[0]).norm() / ys[0].norm()) for v in (1, 4)}
    t = {v: float(np.median(r)) for v, r in res.items()}
    print(f"n={n} d={d} {name:12s}: scalar stream {t[0]:.3f} ms | broadcast, 1 wave/WG {t[1]:.3f} ms (x{t[0]/t[1]:.2f}, diff {dif[1]:.1e}) | 4 waves/WG {t[4]:.3f} ms (x{t[0]/t[4]:.2f}, diff {dif[4]:.1e})"
          f"   expanded {cg.get_info('last_grad_expand')}", flush=True)
cg.set_option("grad_bcast", -1)
