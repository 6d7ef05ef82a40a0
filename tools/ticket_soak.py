"""Soak of the last-arrival (ticketed) reductions (csrc/pack.hpp: last_arrival) after the round-5 fix (every storing wave drains vmcnt in front of the
ticket's barrier): the matrix-core EQ kernel's in-kernel split-J sum (automatic up to n = 4096) and the low-rank first pass's slab sum
(r <= 128), thousands of launches under UNEVEN load — a second stream keeps a changing share of the chip busy with GEMMs of varying size, so that
workgroups of one launch arrive spread over time and over XCDs — every output word compared with the result of the separate reduce launch
(bit-identical by construction: same fixed order).  Prints mismatching launches per case; any non-zero count is a failure."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))
import covgram as cg
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
side = torch.cuda.Stream()
A = [torch.randn(s, s, device="cuda") for s in (512, 1024, 2048, 3072)]
def background(i):
    with torch.cuda.stream(side):
        m = A[i % 4]
        for _ in range(1 + i % 3): m @ m
total_bad = 0
rng = np.random.default_rng(1)
for n, d in ((1000, 3), (2048, 3), (3000, 5), (4096, 3), (4096, 8), (16384, 3)):
    X = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    cg.set_option("mfma_sym", 0)
    G = cg.gramian(cg.EQ(), X); ref = torch.empty_like(a); y = torch.empty_like(a)
    cg.set_option("inkernel_reduce", 0); G.mul_(ref, a); torch.cuda.synchronize()
    cg.set_option("inkernel_reduce", 1)
    bad = 0
    for i in range(iters):
        background(i)
        y.fill_(float("nan"))
        G.mul_(y, a)
        if i % 8 == 7 or i == iters - 1:
            torch.cuda.synchronize()
        if not torch.equal(y, ref): bad += 1
    used = cg.get_info("last_inkernel_reduce")
    print(f"EQ fp32 n={n} d={d}: in-kernel reduce used={used}, {iters} launches under load, mismatching launches: {bad}", flush=True)
    total_bad += bad
cg.set_option("inkernel_reduce", -1); cg.set_option("mfma_sym", -1)
for dt in (torch.float32, torch.float64):
    for nl, r in ((200000, 8), (1 << 18, 32), (150001, 128)):
        xs = torch.randn(nl, dtype=dt, device="cuda")
        G = cg.gramian(cg.FiniteBasis([lambda t, i=i: torch.cos(0.37 * i * t) for i in range(r)]), xs)
        a = torch.randn(nl, dtype=dt, device="cuda"); ref = torch.empty_like(a); y = torch.empty_like(a)
        G.mul_(ref, a); torch.cuda.synchronize()          # the ticketed path itself, quiet chip: the reference for the loaded runs
        bad = 0
        for i in range(iters // 3):
            background(i)
            y.fill_(float("nan"))
            G.mul_(y, a)
            if not torch.equal(y, ref): bad += 1
        print(f"low rank {str(dt)[6:]} n={nl} r={r}: {iters // 3} launches under load, mismatching launches: {bad}", flush=True)
        total_bad += bad
print("TOTAL mismatches:", total_bad)
sys.exit(1 if total_bad else 0)
