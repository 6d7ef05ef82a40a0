"""bench.py — the contract benchmark: Gramian MVMs/s for the dense EQ kernel, n = 131072, d = 3, fp32
(BASELINE.json metric / configs[1]) on N GPUs of one node.

    python3 bench.py [--gpus N] [--steps K] [--warmup W]          (run it as `python3 bench.py`, also after `rocprofv3 ... --`:
                                                                     the file has no shebang, so nothing re-execs)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU; it starts the second command line as a CHILD
(one rank per GPU over RCCL), forwards the child's one JSON line and exits with its return code.  Every rank asserts
WORLD_SIZE == --gpus, and the line carries `rccl_ranks`.  `--dry-launch` runs the same launch with backend gloo and no GPU work
(a CPU-tier check that N ranks really start and meet in a collective).

One "step" = one full MVM b = G a with the points, a and b resident in HBM.  The contract run follows SURVEY.md §8d: y == x,
symmetry NOT exploited — all n*m entries are evaluated (general matrix-core kernel); for N > 1 the rows of G are sharded over
the ranks and ONE RCCL all-gather completes b on every rank (covgram.dist); n stays 131072, so scaling is "strong".
gramian(EQ, x) is symmetric, and by default the library evaluates its upper triangle once (on N GPUs: the cyclic 256-row panels
p % N == rank + ONE all-reduce): that path is timed after the contract run, same protocol, and reported as "symmetric_variant".

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline      the dominant kernel (the library reports which one ran) priced against the FP32 vector peak it is actually bound by
                (the path is VALU/transcendental-bound; the HBM figures BASELINE.json's metric name asks
                for are reported alongside as hbm_*), duration measured live with HIP events on the launch stream;
  cpu_baseline  the C restatement of src/gramian.jl:78-87 (oracle/, "port") timed on this host's cores on a bounded
                row slice (rank 0, N = 1 only) — BEFORE the first GPU call, so that the compiler it may spawn is never a
                child of a process that has initialised the GPU;
  configs       (N = 1) the other BASELINE.json configs at their stated sizes — C1, one rank's share of C3 (row shard and symmetric
                partial), C4, C5 — each with ms per MVM, rel-err against the oracle and its own roofline.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))

N_POINTS = 131072
DIM = 3
SEED = 0xC0F + 1           # seed = 0xC0F + config index
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBPS = 8000.0


def cpu_baseline(X: np.ndarray, a: np.ndarray, budget_s: float = 12.0):
    """Time the CPU restatement (oracle/covgram_oracle.c, -O3 -march=native -fopenmp -ffast-math) on a row slice
    over ALL columns and scale linearly (rows are independent).  The oracle is the thing TIMED here, as the
    reported baseline — it is never on the product path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes
    import c_oracle
    # the OpenMP workers must SLEEP after the baseline's parallel regions: spinning on all host cores (libgomp's default wait
    # policy) they delay this process's kernel launches afterwards (seen as 0.15 ms gaps per step in the symmetric variant's loop)
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    os.environ.setdefault("GOMP_SPINCOUNT", "0")
    out_dir = None
    # under a profiler (rocprofv3 preloads its library into every child) nothing is spawned: the prebuilt object is loaded in-process
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    try:   # rebuild for this host's ISA; fall back to the prebuilt portable object
        if profiled:
            raise RuntimeError("profiler preload detected")
        out_dir = tempfile.mkdtemp(prefix="covgram_oracle_")
        c_oracle.build(march="native", out=out_dir)
        lib = ctypes.CDLL(os.path.join(out_dir, "libcovgram_cpubaseline.so"))
        flags = "-O3 -march=native -fopenmp -ffast-math"
    except Exception:
        lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "build", "libcovgram_cpubaseline.so"))
        flags = "-O3 -march=x86-64-v3 -fopenmp -ffast-math (prebuilt)"
    threads = c_oracle.num_threads(lib)
    n = X.shape[0]
    rows = 512
    t0 = time.perf_counter(); c_oracle.eq_rows(X, X, a, 0, rows, lib); dt = time.perf_counter() - t0   # calibration (+ warm-up)
    rows = int(min(n, max(rows, rows * budget_s / max(dt, 1e-6))))
    rows = max(256, (rows // 256) * 256)
    best = None
    for _ in range(2):
        t0 = time.perf_counter(); c_oracle.eq_rows(X, X, a, 0, rows, lib); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    pairs_per_s = rows * n / best
    try:
        model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    topo = cpu_topology()
    return {
        "value": pairs_per_s / (float(n) * n), "unit": "MVM/s", "cores": threads, "kind": "port",
        "threads": threads, "physical_cores": topo["physical_cores"], "sockets": topo["sockets"], "logical_cpus": topo["logical_cpus"],
        "cores_note": "`cores` = OpenMP threads the run used (the contract's field); physical_cores / sockets / logical_cpus from /proc/cpuinfo: "
                      "with SMT on, threads = 2 x physical_cores",
        "sample": f"{rows} of {n} rows x all {n} columns, {best:.2f} s, scaled by n/rows (rows are independent); "
                  f"C restatement of src/gramian.jl:78-87 (oracle/covgram_oracle.c, {flags}); CPU: {model}",
        "pairs_per_s": pairs_per_s,
    }


def cpu_topology():
    """Sockets, physical cores and logical CPUs of this host from /proc/cpuinfo (unique (physical id, core id) pairs)."""
    cores, socks, logical = set(), set(), 0
    try:
        phys = core = None
        for l in open("/proc/cpuinfo"):
            if l.startswith("processor"):
                logical += 1
            elif l.startswith("physical id"):
                phys = l.split(":")[1].strip(); socks.add(phys)
            elif l.startswith("core id"):
                core = l.split(":")[1].strip(); cores.add((phys, core))
    except Exception:
        pass
    return {"physical_cores": len(cores) or None, "sockets": len(socks) or None, "logical_cpus": logical or (os.cpu_count() or None)}


def _event_loop(fn, steps):
    """K steps with one event pair per step on the current stream (the library launches on it); returns the per-step ms list
    after a synchronise.  The events sit between the steps, so the wall-clock mean of the same loop is unchanged by them."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return [e0.elapsed_time(e1) for e0, e1 in ev]


def _stats(ms):
    v = sorted(ms)
    return {"ms_median": v[len(v) // 2], "ms_min": v[0], "ms_max": v[-1], "steps": len(v)}


def _timed(fn, warm=5, reps=20, after_warm=None, warm_s=0.25):
    # warm up for at least warm_s seconds as well: these configs follow host-side oracle work, and a GPU that sat idle for
    # seconds runs its first hundred milliseconds at low clocks (C5 read 0.095 ms here against 0.087 back to back)
    t0 = time.perf_counter()
    k = 0
    while k < warm or time.perf_counter() - t0 < warm_s:
        fn(); k += 1
        if k % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    if after_warm is not None:
        after_warm()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def _rel(b, ref):
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(b - ref) / np.linalg.norm(ref))


def _rowwise(b, ref, absref):
    """max_i |b_i - ref_i| / (sum_j |G_ij| |a_j|): the row-wise (component-wise) error the norm-wise figure can hide — the matrix-core paths' risk is
    per row (a far point's exponent), tests/ hold every fp32 path to 1e-5 on BOTH."""
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(b - ref) / absref))


def other_configs(cg, dev):
    """BASELINE.json configs[0], [2], [3], [4] at their stated sizes on this one GPU: ms per MVM (host wall over back-to-back calls,
    inputs resident), rel-err against the oracle (checker, outside the timed loops) and the roofline SURVEY.md §8d assigns."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import covgram_oracle as o
    import c_oracle
    out = {}
    # C1: MaternP(2), d=3, n=4096, fp64 (the reference's CPU-runnable case, here through the device path)
    n, d = 4096, 3
    rng = np.random.default_rng(0xC0F + 0)
    Xh = rng.standard_normal((n, d)); ah = rng.standard_normal(n)
    G = cg.gramian(cg.MaternP(2), torch.from_numpy(Xh).to(dev)); a = torch.from_numpy(ah).to(dev); y = torch.empty_like(a)
    ms = _timed(lambda: G.mul_(y, a))
    fl = float(n) * n * (3 * d + 3)
    out["C1"] = {"what": "MaternP(2) dense Gramian mul!, d=3 n=4096 fp64", "ms": ms, "mvm_per_s": 1e3 / ms,
                 "rel_err_vs_fp64_oracle": _rel(y.cpu().numpy(), o.mul(None, o.Kernel(o.MATERNP, p=2), Xh, Xh, ah)),
                 "roofline": {"bound": "valu_fp64", "achieved": fl / (ms * 1e-3) * 1e-12, "peak": 78.6, "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) * 1e-12 / 78.6,
                              "note": "1.7e7 pairs = 4096 waves of 64 rows x 64 columns, half of the chip's wave slots for ONE round: the kernel is ~21 of the 29 us (its floor at the large-n rate of this profile, 1.07e12 pairs/s since the table exponential of round 3, is 16 us), pack + reduction launches the rest"}}
    # The reference README's own dense case (lazy dense mul!, MaternP(2), d=3, n=16384, Float64: 0.584813 s on its unstated
    # CPU, README.md:26-38), not one of BASELINE.json's configs: all n^2 entries (dense_sym = 0) and the library's default for
    # gramian(k, x) in fp64, the symmetric direct-difference kernel (upper triangle once, exact differences).  Reporting only.
    try:
        n, d = 16384, 3
        rng = np.random.default_rng(0xC0F + 5)
        Xh = rng.standard_normal((n, d)); ah = rng.standard_normal(n)
        G = cg.gramian(cg.MaternP(2), torch.from_numpy(Xh).to(dev)); a = torch.from_numpy(ah).to(dev); y = torch.empty_like(a)
        cg.set_option("dense_sym", 0)
        ms_all = _timed(lambda: G.mul_(y, a))
        cg.set_option("dense_sym", -1)
        ms_sym = _timed(lambda: G.mul_(y, a))
        used = cg.get_info("last_dense_sym") == 1
        rows = np.sort(np.random.default_rng(4).choice(n, 512, replace=False))
        ref = c_oracle.mvm(o.Kernel(o.MATERNP, p=2), Xh[rows], Xh, ah)
        out["README_MaternP2_n16384_f64"] = {
            "what": "the reference README's dense case: MaternP(2) Gramian mul!, d=3 n=16384 fp64, gramian(k, x)", "ms": ms_sym, "ms_all_entries": ms_all,
            "used_symmetric_direct_kernel": bool(used), "rel_err_vs_fp64_oracle": _rel(y.cpu().numpy()[rows], ref), "checked_rows": 512,
            "reference_published_s": 0.584813, "reference_source": "README.md:26-38 (CPU unstated)",
            "roofline": {"bound": "valu_fp64", "achieved": float(n) * n * (3 * d + 3) / (ms_all * 1e-3) * 1e-12, "peak": 78.6, "unit": "TFLOP/s",
                         "frac": float(n) * n * (3 * d + 3) / (ms_all * 1e-3) * 1e-12 / 78.6,
                         "note": "frac prices the ALL-entries time with the reference's 3d+3 flops per pair; the fp64 MaternP pair body is 38 instructions (sqrt 11, exp2 by table 14 + one load), DESIGN.md 3.1"}}
    except Exception as e:   # reporting only: never fail the contract line for it
        cg.set_option("dense_sym", -1)
        out["README_MaternP2_n16384_f64"] = {"what": "failed", "error": str(e)[:200]}
    # C3: EQ, d=8, n=524288, fp32 — what ONE of the 8 ranks computes (the 8-GPU run itself is the driver's)
    n, d, world = 524288, 8, 8
    rng = np.random.default_rng(0xC0F + 2)
    Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).to(dev); a = torch.from_numpy(ah).to(dev)
    per = n // world
    G = cg.gramian(cg.EQ(), X[:per], X); y = torch.empty(per, dtype=torch.float32, device=dev)
    cg.set_option("time_kernels", 1)
    ms = _timed(lambda: G.mul_(y, a), warm=3, reps=10, after_warm=cg.kernel_time)
    kms, kl = cg.kernel_time(); cg.set_option("time_kernels", 0)
    rows = np.sort(np.random.default_rng(2).choice(per, 256, replace=False))
    ref = c_oracle.mvm(o.Kernel(o.EQ), Xh[rows].astype(np.float64), Xh.astype(np.float64), ah.astype(np.float64))
    c3_abs = c_oracle.mvm(o.Kernel(o.EQ), Xh[rows].astype(np.float64), Xh.astype(np.float64), np.abs(ah).astype(np.float64))
    fl = float(per) * n * (3 * d + 3)
    kavg = kms / max(kl, 1)
    c3_k2 = (d + 3) // 4 if cg.get_info("last_mfma_f16") == 1 else (d + 1) // 2     # MFMAs per 32 x 32 tile: fp16 two-way split (round 4) / bf16 three-way split
    c3_cycles64 = 8.0 + (2.25 if c3_k2 <= 2 else 4.0) + 8.0 * c3_k2 / 16.0      # K2 <= 2: packed fmas (profiles/r04_pkfma_ab.txt)
    out["C3_shard"] = {"what": "EQ dense Gramian mul!, d=8 n=524288 fp32: one rank's row shard of the 8-GPU config (65536 rows x 524288 columns, all entries)",
                       "ms": ms, "kernel_avg_ms": kavg, "pairs_per_s": float(per) * n / (ms * 1e-3), "rel_err_vs_fp64_oracle": _rel(y.cpu().numpy()[rows], ref),
                       "rowwise_err_vs_fp64_oracle": _rowwise(y.cpu().numpy()[rows], ref, c3_abs),
                       "checked_rows": len(rows),
                       "roofline": {"bound": "valu_issue", "achieved": fl / (kavg * 1e-3) * 1e-12, "peak": 1024 * 2.4e9 * 64 / c3_cycles64 * (3 * d + 3) * 1e-12,
                                    "unit": "TFLOP/s", "frac": (float(per) * n / (kavg * 1e-3)) / (1024 * 2.4e9 * 64 / c3_cycles64),
                                    "mfmas_per_tile": c3_k2, "split": "fp16 two-way (3 products per coordinate)" if c3_k2 == (d + 3) // 4 else "bf16 three-way (8 products per coordinate)",
                                    "reference_flops_frac": fl / (kavg * 1e-3) * 1e-12 / FP32_VECTOR_PEAK_TFLOPS,
                                    "note": "peak = the VALU issue ceiling of this kernel's instruction stream (v_exp_f32 8 + " + ("half a v_pk_fma_f32 2.25" if c3_k2 <= 2 else "v_fma_f32 4") + " + MFMA hold " + f"{8.0 * c3_k2 / 16.0:g}" + " cycles per 64 pairs per SIMD at 2.4 GHz, "
                                            "MI355X_MICROARCH.md) in the reference's 3d+3 flops per pair, so frac = achieved / peak = evaluated pairs per second over that ceiling; reference_flops_frac = the reference's 3d+3 = 27 flops per pair against the FP32 "
                                            "VECTOR peak (SURVEY.md \u00a78d(i)) — 24 of them run on the matrix pipe here, so that ratio can exceed 1 and is no utilisation"}}
    Gf = cg.gramian(cg.EQ(), X); part = torch.empty(n, dtype=torch.float32, device=dev)
    if Gf.sym_partial_supported(world):
        cg.set_option("time_kernels", 1)
        ms = _timed(lambda: Gf.sym_partial_(part, a, 3, world), warm=3, reps=10, after_warm=cg.kernel_time)
        kms, kl = cg.kernel_time(); cg.set_option("time_kernels", 0)
        kavg = kms / max(kl, 1)
        ev = float(n) * (n + 32) / 2 / world
        sp_k2 = (d + 3) // 4 if cg.get_info("last_mfma_f16") == 1 else (d + 1) // 2
        sp_cycles64 = 8.0 + 2 * (2.25 if sp_k2 <= 2 else 4.0) + 8.0 * sp_k2 / 16.0                   # v_exp_f32 + two v_fma_f32 (row and column sums) + the MFMA hold
        out["C3_sym_partial"] = {"what": "the same config in the symmetric form: rank 3 of 8's cyclic panels of the upper triangle (covgram_mvm_sym_partial); "
                                         "an all-reduce of the 8 partials completes b", "ms": ms, "kernel_avg_ms": kavg, "evaluated_pairs_per_s": ev / (ms * 1e-3),
                                 "roofline": {"bound": "valu_issue", "achieved": ev * (3 * d + 5) / (kavg * 1e-3) * 1e-12, "peak": 1024 * 2.4e9 * 64 / sp_cycles64 * (3 * d + 5) * 1e-12, "unit": "TFLOP/s",
                                              "frac": (ev / (kavg * 1e-3)) / (1024 * 2.4e9 * 64 / sp_cycles64), "mfmas_per_tile": sp_k2,
                                              "reference_flops_frac": ev * (3 * d + 5) / (kavg * 1e-3) * 1e-12 / FP32_VECTOR_PEAK_TFLOPS}}
    del G, Gf, X, a, y, part
    # Which path a caller of gramian(EQ(l), x) gets at the contract size by lengthscale (the matrix-core path needs
    # g^2 R^2 <= 126): path, MVM/s, rel-err on 256 oracle rows; and the GP model 1.5 MaternP(2; l = 0.7) + 0.5 EQ(l = 2) on the same cloud.
    # The library's DEFAULT choices throughout (symmetric kernels where they apply); reporting only.
    try:
        n, d = N_POINTS, DIM
        rng = np.random.default_rng(SEED)
        Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
        X = torch.from_numpy(Xh).to(dev); a = torch.from_numpy(ah).to(dev); y = torch.empty_like(a)
        rows = np.sort(np.random.default_rng(7).choice(n, 256, replace=False))
        Xr = Xh[rows].astype(np.float64); Xd = Xh.astype(np.float64); ad = ah.astype(np.float64)
        names = {1: "lane-per-row direct differences", 2: "matrix cores", 3: "wide rows", 4: "factored dot product"}
        gm = []
        for l in (0.3, 0.5, 0.7, 1.0, 2.0):
            G = cg.gramian(cg.Lengthscale(cg.EQ(), l), X)
            ms = _timed(lambda: G.mul_(y, a), warm=3, reps=8, warm_s=0.05)
            path = cg.get_info("last_dense_path")
            sym = bool(cg.get_info("last_mfma_sym") == 1 or cg.get_info("last_dense_sym") == 1)
            ref = c_oracle.mvm(o.Kernel(o.EQ, lengthscale=l), Xr, Xd, ad)
            absr = c_oracle.mvm(o.Kernel(o.EQ, lengthscale=l), Xr, Xd, np.abs(ad))
            split = "" if path != 2 else (", fp16 two-way split" if cg.get_info("last_mfma_f16") == 1 else ", bf16 three-way split")
            gm.append({"lengthscale": l, "path": names.get(path, str(path)) + split + (", upper triangle once" if sym else ", all entries"), "ms": ms, "mvm_per_s": 1e3 / ms,
                       "rel_err_vs_fp64_oracle": _rel(y.cpu().numpy()[rows], ref), "rowwise_err_vs_fp64_oracle": _rowwise(y.cpu().numpy()[rows], ref, absr)})
        out["gate_map_EQ_C2_size"] = {"what": "gramian(Lengthscale(EQ, l), x) * a, d=3 n=131072 fp32, x ~ N(0, I): the library's default path by lengthscale "
                                              "(the matrix-core path is gated on g^2 R^2 <= 126 about the cloud's centre, its fp16 split on <= 72; csrc/dense_mfma.hip)", "rows_checked": 256, "by_lengthscale": gm}
        kc = 1.5 * cg.Lengthscale(cg.MaternP(2), 0.7) + 0.5 * cg.Lengthscale(cg.EQ(), 2.0)
        G = cg.gramian(kc, X)
        ms = _timed(lambda: G.mul_(y, a), warm=3, reps=8, warm_s=0.05)
        ref = 1.5 * c_oracle.mvm(o.Kernel(o.MATERNP, p=2, lengthscale=0.7), Xr, Xd, ad) + 0.5 * c_oracle.mvm(o.Kernel(o.EQ, lengthscale=2.0), Xr, Xd, ad)
        absr = 1.5 * c_oracle.mvm(o.Kernel(o.MATERNP, p=2, lengthscale=0.7), Xr, Xd, np.abs(ad)) + 0.5 * c_oracle.mvm(o.Kernel(o.EQ, lengthscale=2.0), Xr, Xd, np.abs(ad))
        fused = cg.get_info("last_sum_fused") == 1
        out["F2_composite"] = {"what": "1.5 MaternP(2; l=0.7) + 0.5 EQ(l=2) dense Gramian mul!, d=3 n=131072 fp32, gramian(k, x): " +
                                       ("ONE pass of the symmetric matrix-core Sum kernel" if fused else "one symmetric matrix-core MVM per term (the library's rule for two terms: "
                                        "every term's transcendentals remain in a one-pass kernel, profiles/r05_sum_fused_ab.txt)"),
                               "ms": ms, "mvm_per_s": 1e3 / ms, "rel_err_vs_fp64_oracle": _rel(y.cpu().numpy()[rows], ref),
                               "rowwise_err_vs_fp64_oracle": _rowwise(y.cpu().numpy()[rows], ref, absr), "rows_checked": 256,
                               "precision_note": "fp32 MaternP is evaluated in fp32 here; the reference stores MaternP's coefficients as Float64 and so evaluates the profile in "
                                                 "Float64 on Float32 inputs, rounding on store (src/stationary.jl:126-128): a deviation inside the 1e-5 tolerance"}
        # the same Sum with a third term, where the one-pass kernels pay (one MFMA pass, one slab, one reduce for three profiles)
        k3 = cg.Lengthscale(cg.EQ(), 1.4) + 0.7 * cg.Lengthscale(cg.RQ(0.8), 0.9) + 0.2 * cg.MaternP(1)
        G3 = cg.gramian(k3, X)
        ms3 = _timed(lambda: G3.mul_(y, a), warm=3, reps=8, warm_s=0.05)
        f3 = cg.get_info("last_sum_fused") == 1
        ref3 = (c_oracle.mvm(o.Kernel(o.EQ, lengthscale=1.4), Xr, Xd, ad) + 0.7 * c_oracle.mvm(o.Kernel(o.RQ, param=0.8, lengthscale=0.9), Xr, Xd, ad)
                + 0.2 * c_oracle.mvm(o.Kernel(o.MATERNP, p=1), Xr, Xd, ad))
        cg.set_option("sum_fused", 0)
        ms3t = _timed(lambda: G3.mul_(y, a), warm=3, reps=8, warm_s=0.05)
        cg.set_option("sum_fused", -1)
        G3.mul_(y, a)
        out["F2_sum_of_three"] = {"what": "EQ(l=1.4) + 0.7 RQ(0.8; l=0.9) + 0.2 MaternP(1) dense Gramian mul!, d=3 n=131072 fp32, gramian(k, x)", "ms": ms3, "one_pass": bool(f3),
                                  "ms_one_mvm_per_term": ms3t, "rel_err_vs_fp64_oracle": _rel(y.cpu().numpy()[rows], ref3), "rows_checked": 256}
        del G3
        del G, X, a, y
    except Exception as e:
        out["gate_map_EQ_C2_size"] = {"what": "failed", "error": str(e)[:200]}
    # C4: GradientKernel(EQ), d=32, n=16384, fp64
    n, d = 16384, 32
    rng = np.random.default_rng(0xC0F + 3)
    Xh = rng.standard_normal((n, d)); ah = rng.standard_normal(n * d)
    K = cg.gramian(cg.GradientKernel(cg.EQ()), torch.from_numpy(Xh).to(dev)); a = torch.from_numpy(ah).to(dev); y = torch.empty_like(a)
    cg.set_option("time_kernels", 1)
    ms = _timed(lambda: K.mul_(y, a), warm=3, reps=10, after_warm=cg.kernel_time)
    kms, kl = cg.kernel_time(); cg.set_option("time_kernels", 0)
    kavg = kms / max(kl, 1) if kl else ms
    rows = np.sort(np.random.default_rng(3).choice(n, 128, replace=False))
    ref = c_oracle.grad_mvm(o.Kernel(o.EQ), Xh[rows], Xh, ah)
    fl = float(n) * n * (10 * d + 12)
    out["C4"] = {"what": "GradientKernel(EQ) mul!, d=32 n=16384 fp64", "ms": ms, "kernel_avg_ms": kavg, "mvm_per_s": 1e3 / ms,
                 "rel_err_vs_fp64_oracle": _rel(y.cpu().numpy().reshape(n, d)[rows].reshape(-1), ref), "checked_rows": len(rows),
                 "roofline": {"bound": "valu_fp64", "achieved": fl / (kavg * 1e-3) * 1e-12, "peak": 78.6, "unit": "TFLOP/s", "frac": fl / (kavg * 1e-3) * 1e-12 / 78.6,
                              "algorithmic_flops": fl, "kernel": "covgram::grad_bcast_kernel<EQ, 32, 4 waves>" if cg.get_info("last_grad_bcast") else "covgram::grad_mvm_kernel<double, EQ, 32, expanded>",
                              "note": "round 4: column records in VGPRs (counted vector loads), operands by v_fmac_f64_dpp row_newbcast; 158 VALU instructions per wave and column, VALU-issue bound (profiles/r04_c4_bcast_pmc.txt)"}}
    del K, a, y
    # C5: Exponential on range(-1, 1, 2^22), fp64: Toeplitz MVM through the circulant embedding N = 2^23
    n = 1 << 22
    T = cg.gramian(cg.Exp(), cg.srange(-1, 1, n))
    ah = np.random.default_rng(0xC0F + 4).standard_normal(n)
    a = torch.from_numpy(ah).to(dev); y = torch.empty_like(a)
    ms = _timed(lambda: T.mul_(y, a), warm=5, reps=20)
    ref = o.toeplitz_mul(None, T.vc.cpu().numpy(), None, ah)
    by = 112.0 * n
    out["C5"] = {"what": "Exponential on range(-1,1,2^22): SymmetricToeplitz mul!, fp64, cached spectrum", "ms": ms, "mvm_per_s": 1e3 / ms,
                 "rel_err_vs_numpy_fft": _rel(y.cpu().numpy(), ref),
                 "roofline": {"bound": "hbm", "achieved": by / (ms * 1e-3) * 1e-9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": by / (ms * 1e-3) * 1e-9 / HBM_PEAK_GBPS,
                              "algorithmic_bytes": by, "note": "112 n bytes = the ideal single-pass traffic of SURVEY.md §8d; ms is the whole call (all its kernels)"}}
    return out


def _free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(args, argv) -> int:
    """`bench.py --gpus N` run by hand (no WORLD_SIZE): start N ranks of this file under torch.distributed.run as a CHILD process
    — this process has made no GPU call (torch.cuda.device_count() does not initialise the runtime on this image) and is never
    replaced —, forward the ONE JSON line rank 0 prints and return the child's exit code."""
    if not args.dry_launch and os.environ.get("COVGRAM_BENCH_REHEARSAL") != "1":
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible\n")
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in r.stdout.splitlines():
        if ln not in lines:
            sys.stderr.write(ln + "\n")
    if r.returncode == 0 and len(lines) != 1:
        sys.stderr.write(f"bench.py: expected one JSON line from the {args.gpus} ranks, got {len(lines)}\n")
        return 3
    if lines:
        sys.stdout.write(lines[-1] + "\n"); sys.stdout.flush()
    return r.returncode


def dry_launch(args, world, rank) -> None:
    """--dry-launch: the ranks meet over gloo, no GPU work — every rank contributes (rank, pid) to one all-gather and rank 0 prints a
    line in the contract's shape with value null.  What it proves: N distinct processes were started by `--gpus N` and share a group."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = torch.tensor([rank, os.getpid()], dtype=torch.int64)
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    ranks = [int(t[0]) for t in allr]; pids = [int(t[1]) for t in allr]
    assert ranks == list(range(world)) and len(set(pids)) == world, (ranks, pids)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "Gramian MVMs/sec, dense EQ kernel, n=131072, d=3, fp32 (dry launch: no GPU work)", "value": None, "unit": "MVM/s",
                          "n_gpus": world, "rccl_ranks": 0, "gloo_ranks": world, "backend": "gloo", "dry_launch": True, "ranks": ranks, "pids": pids,
                          "steps": 0, "warmup": 0}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C1 / C3 / C4 / C5 block (N = 1)")
    ap.add_argument("--dry-launch", action="store_true", help="start the --gpus N ranks over gloo, one all-gather, no GPU work (CPU-tier launch check)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_launch):
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write(f"bench.py: WORLD_SIZE={os.environ.get('WORLD_SIZE')} but --gpus {args.gpus}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)\n")
        sys.exit(2)
    if args.dry_launch:
        dry_launch(args, int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")))
        return

    # RCCL prints its version banner and warnings on fd 1; keep the contract's ONE JSON line clean by routing every other
    # write to stdout (C libraries included) to stderr and printing the result on the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus

    rng = np.random.default_rng(SEED)                       # identical on every rank: replicated inputs
    Xh = rng.standard_normal((N_POINTS, DIM)).astype(np.float32)
    ah = rng.standard_normal(N_POINTS).astype(np.float32)

    # the CPU baseline first: no GPU call has been made yet (torch.cuda is not initialised until set_device below)
    cpu_line = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        try:
            cpu_line = cpu_baseline(Xh, ah)
        except Exception as e:   # the baseline is reporting only; never fail the GPU measurement for it
            cpu_line = {"value": None, "unit": "MVM/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}

    # REHEARSAL of the N > 1 code path on a box with ONE GPU (tests/test_gpu_dist.py): COVGRAM_BENCH_REHEARSAL=1 puts every rank on GPU 0
    # and runs the collectives over gloo (covgram.dist stages the MVM's one collective through host memory then) — every line of the
    # multi-rank path below executes, only RCCL itself does not; the line says so (`rehearsal`, `rccl_ranks` 0) and is no measurement.
    rehearsal = world > 1 and os.environ.get("COVGRAM_BENCH_REHEARSAL") == "1"
    if world > 1 or os.environ.get("COVGRAM_FORCE_COLLECTIVE") == "1":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        if rehearsal:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            if torch.cuda.device_count() <= local_rank:
                sys.stderr.write(f"bench.py: rank {rank} wants GPU {local_rank}, {torch.cuda.device_count()} visible\n")
                sys.exit(2)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
        assert dist.get_world_size() == args.gpus
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    sdev = torch.device("cpu") if rehearsal else dev          # where the few statistics scalars are reduced (gloo: host tensors)

    import covgram as cg

    X = torch.from_numpy(Xh).to(dev)
    a = torch.from_numpy(ah).to(dev)

    # The contract workload: y == x but the symmetry is NOT exploited — every one of the n*m entries is
    # evaluated, as the reference does (src/gramian.jl:78-87): the general matrix-core kernel on one GPU, row shards + one
    # all-gather on N.  The library's symmetric kernels (upper triangle once; cyclic panels + one all-reduce on N GPUs) are what
    # a caller of gramian(EQ, x) gets by default; they are timed after the contract run and reported in "symmetric_variant".
    cg.set_option("mfma_sym", 0)
    G = cg.ShardedGramian(cg.EQ(), X, symmetric=False)      # rank's row shard of gramian(EQ(), x) + the all-gather
    b = torch.empty(N_POINTS, dtype=torch.float32, device=dev)

    def step():
        G.matmul(a, out=b)

    # setup, not part of the W + K protocol: the first calls size the library's workspaces, pack the cached fragments, upload the
    # symmetric kernel's workgroup list and (N > 1) open the RCCL channels; a short burst also brings the GPU out of its idle clocks
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    cg.set_option("time_kernels", 1)
    collective = world > 1 or G.force_collective
    G.timing = False                                         # the timed steps take the product route: ONE library call per MVM (kernel + collective on the ctx stream)
    step_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e0, e1 in step_ev:                                    # EXACTLY K steps; one event pair per step on the launch stream
        e0.record(); step(); e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = [e0.elapsed_time(e1) for e0, e1 in step_ev]
    kernel_ms, launches = cg.kernel_time()
    cg.set_option("time_kernels", 0)
    # where a step's time goes — AFTER the timed region, diagnostic: events around the local kernel and around the one collective, for the
    # route the timed steps took (C ABI: the library's own ncclAllGather on its stream) and for torch.distributed's (its own stream + two event hops)
    loc_ms, col_ms, nsplit, col_by_route = 0.0, 0.0, 0, None
    if collective:
        G.timing = True
        for _ in range(3):
            step()
        G.timing_ms()
        for _ in range(10):
            step()
        loc_ms, col_ms, nsplit = G.timing_ms()
        G.timing = False
        col_by_route = {"c_abi" if getattr(G, "_abi", False) else "torch_distributed": col_ms / max(nsplit, 1)}
        if getattr(G, "_abi", False):
            os.environ["COVGRAM_ABI_COLLECTIVE"] = "0"
            try:
                Gt = cg.ShardedGramian(cg.EQ(), X, symmetric=False)
                bt = torch.empty_like(b)
                Gt.timing = True
                for _ in range(3):
                    Gt.matmul(a, out=bt)
                Gt.timing_ms()
                for _ in range(10):
                    Gt.matmul(a, out=bt)
                _, ct, nt = Gt.timing_ms()
                col_by_route["torch_distributed"] = ct / max(nt, 1)
                del Gt, bt
            finally:
                os.environ.pop("COVGRAM_ABI_COLLECTIVE", None)
    dense_path = cg.get_info("last_dense_path")
    sym_path = cg.get_info("last_mfma_sym") == 1
    f16_split = dense_path == 2 and cg.get_info("last_mfma_f16") == 1      # the matrix-core EQ kernels' fp16 two-way split of the coordinates (round 4)
    mfma_inst = cg.get_info("last_mfma_instance")                          # template arguments of the kernel the timed steps launched

    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms / max(launches, 1)], dtype=torch.float64, device=sdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_avg_ms = float(t[0]), float(t[1])
        # every rank's own split of a step, gathered to rank 0: where the time of an N-GPU step goes
        mine = torch.tensor([kernel_ms / max(launches, 1), loc_ms / max(nsplit, 1), col_ms / max(nsplit, 1), float(G.hi - G.lo),
                             float(np.median(step_ms)), float(min(step_ms))], dtype=torch.float64, device=sdev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "kernel_ms": float(v[0]), "local_ms": float(v[1]), "collective_ms": float(v[2]), "shard_rows": int(v[3]),
                     "step_ms_median": float(v[4]), "step_ms_min": float(v[5])} for r, v in enumerate(allr)]
    else:
        kern_avg_ms = kernel_ms / max(launches, 1)
        if collective:
            per_rank = [{"rank": 0, "kernel_ms": kern_avg_ms, "local_ms": loc_ms / max(nsplit, 1), "collective_ms": col_ms / max(nsplit, 1),
                         "shard_rows": int(G.hi - G.lo), "step_ms_median": float(np.median(step_ms)), "step_ms_min": float(min(step_ms))}]

    # sustained shader clock under THIS kernel: a few launches of its clock-stamping diagnostic build (same loop, s_memtime /
    # s_memrealtime around the column loop of every workgroup; MI355X_MICROARCH.md DVFS item 6), untimed, right after the run
    clock_ghz = None
    if dense_path == 2 and not sym_path:
        cg.set_option("mfma_stamp", 1)
        ks = []
        for _ in range(5):
            step()
            ks.append(cg.get_info("last_clock_khz"))
        cg.set_option("mfma_stamp", 0)
        ks = [k for k in ks if k > 0]
        if ks:
            clock_ghz = float(np.median(ks)) * 1e-6
    if world > 1:
        t = torch.tensor([clock_ghz or 0.0], dtype=torch.float64, device=sdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        clock_ghz = float(t[0]) or None

    # the contract run's result, kept for the parity spot check — which runs on the host for seconds and therefore AFTER the
    # symmetric variant's loop below (a GPU left idle that long re-enters its timed loop at low clocks: the variant read 430-730
    # MVM/s instead of 840 when the check sat between the two loops)
    rel_err = None
    got = None
    if rank == 0:
        rows = np.random.default_rng(1).choice(N_POINTS, 1024, replace=False)
        got = b.cpu().numpy()[rows].astype(np.float64)

    # ---- the same MVM on the library's default path for gramian(EQ, x): symmetric kernels (same protocol, after the contract run)
    cg.set_option("mfma_sym", -1)
    Gs = cg.ShardedGramian(cg.EQ(), X)
    bs = torch.empty_like(b)
    for _ in range(20 + args.warmup):
        Gs.matmul(a, out=bs)
    torch.cuda.synchronize()
    cg.set_option("time_kernels", 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    s_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in s_ev:
        e0.record(); Gs.matmul(a, out=bs); e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    s_elapsed = time.perf_counter() - t0
    s_step = _stats([e0.elapsed_time(e1) for e0, e1 in s_ev])
    s_kernel_ms, s_launches = cg.kernel_time()
    cg.set_option("time_kernels", 0)
    s_used = cg.get_info("last_mfma_sym") == 1
    if world > 1:
        t = torch.tensor([s_elapsed, s_kernel_ms / max(s_launches, 1)], dtype=torch.float64, device=sdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        s_elapsed, s_kern_avg_ms = float(t[0]), float(t[1])
    else:
        s_kern_avg_ms = s_kernel_ms / max(s_launches, 1)
    # ---- the reference's own arithmetic: the lane-per-row direct-difference kernel (dense_variant = 1: r^2 from the d differences in
    # fp32, src/util.jl:40-47, no matrix cores), all n*m entries, same protocol with fewer steps (it is ~1.7x slower)
    dd = None
    try:
        cg.set_option("mfma_sym", 0); cg.set_option("dense_variant", 1); cg.set_option("dense_sym", 0)     # all n*m entries (no symmetric direct kernel)
        Gd = cg.ShardedGramian(cg.EQ(), X, symmetric=False)
        bd = torch.empty_like(b)
        d_steps = max(5, min(args.steps, 20))
        for _ in range(5):
            Gd.matmul(a, out=bd)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d_ms = _event_loop(lambda: Gd.matmul(a, out=bd), d_steps)
        if world > 1:
            dist.barrier()
        d_elapsed = time.perf_counter() - t0
        d_path = cg.get_info("last_dense_path")
        if world > 1:
            t = torch.tensor([d_elapsed], dtype=torch.float64, device=sdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d_elapsed = float(t[0])
        dd = {"what": "the same contract workload on the reference's arithmetic: direct differences per pair in fp32 on the VALU (option dense_variant = 1, "
                      "covgram::dense_mvm_kernel, one lane per output row), all n*m entries",
              "used_direct_difference_kernel": d_path == 1, "value": d_steps / d_elapsed, "unit": "MVM/s", "ms_per_step": d_elapsed / d_steps * 1e3,
              "steps": d_steps, **{k: v for k, v in _stats(d_ms).items() if k != "steps"}}
        got_d = bd.cpu().numpy() if rank == 0 else None
    except Exception as e:   # reporting only
        dd = {"what": "failed", "error": str(e)[:200]}
        got_d = None
    finally:
        cg.set_option("dense_variant", 0); cg.set_option("mfma_sym", -1); cg.set_option("dense_sym", -1)

    # ---- the contract step including the transfers of a (host -> device) and b (device -> host), pinned host buffers (SURVEY.md \u00a78d:
    # "a second number includes H2D/D2H of a, b"); never `value`
    incl = None
    if world == 1:
        try:
            cg.set_option("mfma_sym", 0)
            a_h = torch.from_numpy(ah).pin_memory(); b_h = torch.empty(N_POINTS, dtype=torch.float32).pin_memory()
            def step_io():
                a.copy_(a_h, non_blocking=True); G.matmul(a, out=b); b_h.copy_(b, non_blocking=True)
            for _ in range(5):
                step_io()
            torch.cuda.synchronize()
            k_io = max(5, min(args.steps, 20))
            t0 = time.perf_counter()
            for _ in range(k_io):
                step_io()
            torch.cuda.synchronize()
            io_ms = (time.perf_counter() - t0) / k_io * 1e3
            incl = {"what": "one contract step with a copied host -> device before and b device -> host after it (pinned buffers, same stream)",
                    "ms_per_step": io_ms, "value": 1e3 / io_ms, "unit": "MVM/s", "steps": k_io, "bytes_per_step": 2 * 4 * N_POINTS}
        except Exception as e:
            incl = {"what": "failed", "error": str(e)[:200]}
        finally:
            cg.set_option("mfma_sym", -1)

    # ---- N > 1: what ONE GPU needs for the whole MVM (rank 0, after the timed regions): ideal_ms = t1 / N
    t1_ms = None
    if world > 1 and rank == 0:
        cg.set_option("mfma_sym", 0)
        G1 = cg.gramian(cg.EQ(), X); b1 = torch.empty_like(b)
        for _ in range(3):
            G1.mul_(b1, a)
        torch.cuda.synchronize()
        t1_ms = float(np.median(_event_loop(lambda: G1.mul_(b1, a), 10)))
        cg.set_option("mfma_sym", -1)
    if world > 1:
        dist.barrier()

    symmetric_variant = None
    if rank == 0:
        # parity spot check outside the timed regions: 1024 random rows against the fp64 oracle (tests/ hold the full suite)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import covgram_oracle as o
        ref = o.mul(None, o.Kernel(o.EQ), Xh[rows], Xh, ah, dtype=np.float32)
        absref = o.mul(None, o.Kernel(o.EQ), Xh[rows], Xh, np.abs(ah), dtype=np.float32)      # sum_j G_ij |a_j| (G > 0): the row-wise error's scale
        rel_err = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        row_err = _rowwise(got, ref, absref)
        gots = bs.cpu().numpy()[rows].astype(np.float64)
        if dd is not None and got_d is not None:
            dd["rel_err_vs_fp64_oracle"] = float(np.linalg.norm(got_d[rows].astype(np.float64) - ref) / np.linalg.norm(ref))
            dd["rowwise_err_vs_fp64_oracle"] = _rowwise(got_d[rows].astype(np.float64), ref, absref)
        symmetric_variant = {
            "what": "the same mul!(b, gramian(EQ, x), a) on the library's default path: the upper triangle of the symmetric Gramian is "
                    "evaluated once (row and column sums of the same tiles)" + ("" if world == 1 else f"; rank r of {world} takes the cyclic "
                    "256-row panels p % N == r and one RCCL all-reduce completes b") + " — NOT the contract workload, which evaluates all n*m entries",
            "used_symmetric_kernel": bool(s_used),
            "value": args.steps / s_elapsed, "unit": "MVM/s", "ms_per_step": s_elapsed / args.steps * 1e3,
            "kernel_avg_ms": s_kern_avg_ms, "ms_median": s_step["ms_median"], "ms_min": s_step["ms_min"],
            "rel_err_vs_fp64_oracle": float(np.linalg.norm(gots - ref) / np.linalg.norm(ref)),
            "rowwise_err_vs_fp64_oracle": _rowwise(gots, ref, absref),
            "evaluated_pairs_per_launch": float(N_POINTS) * (N_POINTS + 32) / 2 / world,
        }

    if rank == 0:
        n, m, d = N_POINTS, N_POINTS, DIM
        n_local = (n + world - 1) // world
        ms_per_step = elapsed / args.steps * 1e3
        mvms = args.steps / elapsed
        flops_launch = float(n_local) * m * (3 * d + 3)                 # 3d+3 flops per pair (+1 exp)
        bytes_launch = 4.0 * (n_local * d + m * (d + 1) + n_local)      # compulsory: x rows, packed (y, a) stream, b
        kern_s = kern_avg_ms * 1e-3
        achieved_tflops = flops_launch / kern_s * 1e-12
        # roofline.traffic: HBM bytes per launch from the latest recorded rocprofv3 PMC pass — ONLY when that pass profiled the very kernel
        # instance this run launched (the library reports its template arguments; tools/pmc_json.py records the profiled kernel's name);
        # otherwise null with the reason (the figure used to be a file read that no kernel change could invalidate)
        traffic, traffic_note = None, None
        ran_name = None
        if dense_path == 2 and not sym_path and mfma_inst > 0:
            q = mfma_inst
            ran_name = "covgram::dense_mfma_eq_kernel<%d, %d, %d, %d, %d, %d>" % (q // 100000, q // 10000 % 10, q // 1000 % 10, q // 100 % 10, q // 10 % 10, q % 10)
        pmc = None
        for rnd in ("r05", "r04", "r03", "r02", "r01"):                            # the latest recorded PMC pass of this kernel family
            cand = os.path.join(ROOT, "profiles", f"{rnd}_" + (("dense_mfma_sym_pmc.json" if sym_path else "dense_mfma_pmc.json") if dense_path == 2
                                                               else "dense_pmc.json"))
            if os.path.exists(cand):
                pmc = cand
                break
        if pmc:
            try:
                rec = json.load(open(pmc))
                prof_name = rec.get("kernel_name") or rec.get("kernel", "")
                if ran_name is None:
                    traffic_note = f"{os.path.basename(pmc)} exists, but this run's kernel is not the general matrix-core EQ kernel whose instance the library reports"
                elif prof_name.replace(" ", "").startswith(ran_name.replace(" ", "")):
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_note = f"{os.path.basename(pmc)}: a PMC pass of {ran_name}, the instance this run launched"
                else:
                    traffic_note = f"{os.path.basename(pmc)} profiled {prof_name.split('(')[0].strip()}, this run launched {ran_name}: no traffic figure"
            except Exception as e:
                traffic, traffic_note = None, f"unreadable PMC record: {e}"
        else:
            traffic_note = "no PMC record under profiles/"
        # which kernel ran (the library picks the matrix-core EQ path when its norm bound holds)
        # Issue pricing (MI355X_MICROARCH.md, row 'vector-instruction ISSUE cost'): per wave-instruction and SIMD v_exp_f32 8 cycles,
        # v_fma_f32 4, and each v_mfma_f32_32x32x16_bf16 holds the SIMD's vector issue for 8 of its 32 cycles; costs add.
        evaluated_pairs = float(n_local) * m
        k2 = (d + 3) // 4 if f16_split else (d + 1) // 2             # MFMAs per 32x32 tile (bf16 three-way split: two coordinates each; fp16 two-way split: four)
        mfma_hold = 8.0 * k2 / 16.0                                  # per 64 pairs: k2 MFMAs serve 1024 pairs
        if dense_path == 2 and sym_path:
            # gramian(k, x) on one GPU: tiles on / above the diagonal are evaluated once and feed row AND column sums
            evaluated_pairs = float(n) * (n + 32) / 2 / world       # per rank: cyclic panels of the triangle
            # this algorithm's own work: per evaluated pair the reference's 3d+3 flops + the second weighted sum
            flops_launch = evaluated_pairs * (3 * d + 3 + 2)
            achieved_tflops = flops_launch / kern_s * 1e-12
            kname = ("covgram::dense_mfma_sym_kernel<EQ, K2=" + str(k2) + "> (upper triangle only: " + ("fp16 two-way split v_mfma_f32_32x32x16_f16" if f16_split else "bf16x3-split v_mfma_f32_32x32x16_bf16") + " + 1 v_exp_f32 + "
                     "2 v_fma_f32 per EVALUATED pair, each evaluated pair serves the entries (i,j) and (j,i); 8 waves share each column "
                     "tile through LDS)")
            cycles64 = 8.0 + 2 * (2.25 if k2 <= 2 else 4.0) + mfma_hold      # K2 <= 2: the two weighted sums are v_pk_fma_f32, 4.5 cycles per 128 fmas (profiles/r04_pkfma_ab.txt)
            note = ("FP32 VALU + transcendental issue bound. The Gramian of one point set is symmetric: every 32x32 tile on or above the "
                    "diagonal is evaluated once (distance on the bf16 matrix pipe, three-way split) and used for the row sums and — "
                    "strictly above the diagonal — for the column sums, so one MVM costs n(n+32)/2 exponentials instead of n^2 "
                    "(the reference evaluates all n^2, src/gramian.jl:78-87). 'achieved' counts THIS algorithm's flops: evaluated pairs x "
                    "(3d+3 of SURVEY.md \u00a78d + 2 for the second weighted sum) over the measured kernel time; 'reference_algorithm_tflops' "
                    "is the reference's n^2 x (3d+3) over the same time (what the caller gets). 'hbm' does not bound this kernel.")
        elif dense_path == 2:
            kname = (f"covgram::dense_mfma_eq_kernel<K2={k2}, RT=2, WPB=8, LDS, fp16 two-way split> (v_mfma_f32_32x32x16_f16 + 1 v_exp_f32 + "
                     "1 fma per pair as v_pk_fma_f32 on register pairs; 8 waves share each column tile through LDS)" if f16_split else
                     "covgram::dense_mfma_eq_kernel<K2=2, RT=2, WPB=8, LDS> (bf16x3-split v_mfma_f32_32x32x16_bf16 + 1 v_exp_f32 + "
                     "1 v_fma_f32 per pair; 8 waves share each column tile through LDS)")
            cycles64 = 8.0 + (2.25 if k2 <= 2 else 4.0) + mfma_hold           # K2 <= 2: the weighted sum is v_pk_fma_f32, 4.5 cycles per 128 fmas (profiles/r04_pkfma_ab.txt)
            note = ("FP32 VALU + transcendental issue bound: the distance runs on the matrix pipe (" + ("fp16 two-way split x~ = h1 + h2, products h1 h1, "
                    "h1 h2, h2 h1: one MFMA per four coordinates; what it drops is the size of one fp32 rounding of the dot product" if f16_split else
                    "bf16 three-way split, fp32-exact products") + "), the VALU does 1 v_exp_f32 + 1 v_fma_f32 per pair. 'achieved' is the reference's algorithmic 3d+3 flops per "
                    "pair over the measured kernel time; 'peak' is the fixed FP32 vector peak, so 'frac' is the round-to-round comparable throughput ratio; "
                    "'issue_roofline_frac' is the utilisation of the binding pipe: the kernel's VALU issue ceiling (v_exp_f32 8 + " + ("half a v_pk_fma_f32 2.25" if k2 <= 2 else "v_fma_f32 4") + " + "
                    "MFMA hold " + f"{mfma_hold:g}" + " cycle per 64 pairs per SIMD at the 2.4 GHz peak clock); "
                    "'issue_roofline_frac_at_sustained_clock' prices the same ceiling at the clock measured in this run with the kernel's "
                    "stamping build; 'reference_flops_frac' is the flop ratio against the FP32 vector peak of SURVEY.md \u00a78d(i) (most of those "
                    "flops run on the matrix pipe: a throughput ratio, not a utilisation). 'hbm' does not bound this kernel (O(n) bytes, O(n^2) work).")
        else:
            kname = "covgram::dense_mvm_kernel<float, EQ, D=3, NRHS=1, R=1>"
            cycles64 = 23.2                                           # measured packed body, profiles/r01_microbench_valu_rates.txt
            note = ("FP32 VALU + transcendental issue bound (12 flop + 1 v_exp_f32 per pair, O(n) bytes for O(n^2) work); "
                    "'mfma'/'hbm' do not bound this kernel (SURVEY.md §8d, DESIGN.md §4).")
        ceiling = 1024 * 2.4e9 * 64 / cycles64                        # evaluated pairs/s at the 2.4 GHz peak clock
        ceiling_sustained = 1024 * clock_ghz * 1e9 * 64 / cycles64 if clock_ghz else None
        flops_per_pair = flops_launch / evaluated_pairs
        on_matrix_cores = dense_path == 2
        # roofline.frac is the SAME definition in every round's BENCH file (SURVEY.md \u00a78d(i)): the reference's algorithmic 3d+3 flops per
        # pair over the measured kernel time against the FIXED FP32 vector peak of the chip.  The utilisation of the pipe that actually
        # binds the matrix-core kernels — the VALU issue stream (exp + fma + MFMA hold) — is a different question with a kernel-specific
        # ceiling; it lives under its own keys (issue_roofline_frac, issue_peak_tflops) and never replaces frac.
        peak_tflops = FP32_VECTOR_PEAK_TFLOPS
        issue_peak_tflops = ceiling * flops_per_pair * 1e-12
        step_stats = _stats(step_ms)
        line = {
            "metric": "Gramian MVMs/sec, dense EQ kernel, n=131072, d=3, fp32 (+ achieved HBM GB/s in roofline.hbm_*)",
            "value": mvms, "unit": "MVM/s", "n_gpus": world, "rccl_ranks": world if (dist.is_initialized() and not rehearsal) else 0, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_median": step_stats["ms_median"], "ms_min": step_stats["ms_min"], "ms_max": step_stats["ms_max"],
            "timing_note": "value / ms_per_step: wall clock over the K steps between barrier + synchronize (the contract); ms_median / ms_min: one HIP event pair "
                           "per step on the launch stream, same K steps",
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "EQ dense Gramian mul!, d=3 n=131072 fp32 (BASELINE.json configs[1]); x ~ N(0, I_3), a ~ N(0,1), "
                                   "alpha=1, beta=0, y == x" + ("; the upper triangle is evaluated once" if sym_path else "; all n*m entries evaluated") + "; points/a/b resident in HBM",
                       "n": n, "d": d, "kernel": "EQ", "pairs_per_mvm": float(n) * m,
                       "parallelism": "1 GPU" if world == 1 else (
                           f"upper triangle by cyclic 256-row panels x{world} + 1 RCCL all-reduce of b per MVM" if sym_path
                           else f"row-shard x{world} + 1 RCCL all-gather of b per MVM")},
            "pairs_per_s": mvms * float(n) * m,
            "rel_err_vs_fp64_oracle": rel_err, "rowwise_err_vs_fp64_oracle": row_err,
            "err_note": "1024 random rows against the fp64 oracle after the timed regions: rel = |b - ref|_2 / |ref|_2, rowwise = max_i |b_i - ref_i| / sum_j |G_ij a_j|",
            "symmetric_variant": symmetric_variant,
            "direct_difference_variant": dd,
            "incl_h2d_d2h": incl,
            "roofline": {
                "bound": "valu", "binding_pipe": "valu_issue" if on_matrix_cores else "valu", "kernel": kname,
                "achieved": achieved_tflops, "peak": peak_tflops, "unit": "TFLOP/s",
                "frac": achieved_tflops / peak_tflops,
                "frac_definition": "v1 (rounds 1, 2, 4+): algorithmic flops per launch / kernel_avg_ms / the fixed FP32 vector peak; round 3's file "
                                   "carried the issue-ceiling ratio here, which is issue_roofline_frac in every round",
                "peak_note": "FP32 vector peak (MI355X_MICROARCH.md)",
                "issue_peak_tflops": issue_peak_tflops,
                "issue_peak_note": ("the VALU issue ceiling of this kernel's stream — " + f"{cycles64:g}" + " issue cycles per 64 pairs per SIMD (v_exp_f32 8, v_fma_f32 4, "
                                    "MFMA hold; MI355X_MICROARCH.md) on 1024 SIMDs at 2.4 GHz — times the algorithm's flops per pair; issue_roofline_frac = achieved / that"),
                "reference_flops_frac": float(n_local) * m * (3 * d + 3) / kern_s * 1e-12 / FP32_VECTOR_PEAK_TFLOPS,
                "vector_peak_tflops": FP32_VECTOR_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": "recorded rocprofv3 --pmc pass (profiles/, FETCH_SIZE doubled per the guide), not measured in this run; taken only when that pass "
                                  "profiled the kernel instance this run launched", "traffic_check": traffic_note, "kernel_instance": ran_name,
                "kernel_avg_ms": kern_avg_ms, "launches": int(launches),
                "algorithmic_flops_per_launch": flops_launch,
                "note": note,
                "reference_algorithm_tflops": float(n_local) * m * (3 * d + 3) / kern_s * 1e-12,
                "issue_roofline_frac": (evaluated_pairs / kern_s) / ceiling,
                "issue_cycles_per_64_pairs_per_simd": cycles64,
                "sustained_clock_ghz": clock_ghz,
                "issue_roofline_frac_at_sustained_clock": (evaluated_pairs / kern_s) / ceiling_sustained if ceiling_sustained else None,
                "evaluated_pairs_per_launch": evaluated_pairs,
                "hbm_algorithmic_bytes_per_launch": bytes_launch,
                "hbm_achieved_GBps": bytes_launch / kern_s * 1e-9,
                "hbm_frac": bytes_launch / kern_s * 1e-9 / HBM_PEAK_GBPS,
            },
        }
        if rehearsal:
            line["rehearsal"] = "COVGRAM_BENCH_REHEARSAL=1: all ranks on GPU 0, collectives over gloo through host memory — a code-path rehearsal, NOT a measurement"
        if per_rank is not None:
            line["per_rank"] = per_rank
            line["per_rank_note"] = ("kernel_ms: the rank's dominant kernel by HIP events on the launch stream over the K timed steps; local_ms / collective_ms: 10 steps "
                                     "AFTER the timed region with events around the shard's MVM and around the ONE collective (the timed steps make one library call per MVM)")
        if col_by_route is not None:
            line["collective_route"] = ("C ABI: covgram_mvm_sharded = shard kernel + ncclAllGather enqueued on the ctx stream (include/covgram.h)" if getattr(G, "_abi", False)
                                        else "torch.distributed all_gather_into_tensor")
            line["collective_ms_by_route"] = col_by_route
        if t1_ms is not None:
            line["single_gpu_ms"] = t1_ms
            line["ideal_ms"] = t1_ms / world
            line["ideal_note"] = "single_gpu_ms: the whole n x n MVM (same kernel, all entries) on rank 0's GPU alone, median of 10 event-timed steps after the run; ideal_ms = that / N"
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        if world == 1 and not args.no_configs:
            try:
                line["configs"] = other_configs(cg, dev)
            except Exception as e:   # reporting only: the contract line above stands on its own
                line["configs"] = {"error": repr(e)}
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
