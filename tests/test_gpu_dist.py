"""Two REAL processes on the device path: world_size 2 over gloo, both ranks on GPU 0, the DEFAULT local
factory (the HIP kernels through the C ABI), device tensors staged through host memory for the one collective (covgram.dist) —
row shards + all-gather and symmetric partials + all-reduce, against the fp64 oracle.  Plus bench.py's RCCL path at world = 1
(COVGRAM_FORCE_COLLECTIVE=1): init_process_group("nccl"), the all-gather / all-reduce really issued on one rank.
The 8-GPU run itself is the driver's; this proves the multi-process logic with real kernels, not a CPU stand-in."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import covgram as cg
    import covgram_oracle as o
    dev = torch.device("cuda", 0)                       # both ranks share the one GPU of the box
    res = {}
    try:
        rng = np.random.default_rng(2024)               # same seed on every rank: replicated inputs
        # (1) fp32 EQ, d = 3: row shards + all-gather (general matrix-core kernel), ragged n (shards of 15001 / 15000 rows)
        n, d = 30001, 3
        Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
        X = torch.from_numpy(Xh).to(dev); a = torch.from_numpy(ah).to(dev)
        rows = np.random.default_rng(1).choice(n, 256, replace=False)
        ref = o.mul(None, o.Kernel(o.EQ), Xh[rows], Xh, ah, dtype=np.float32)
        G = cg.ShardedGramian(cg.EQ(), X, symmetric=False)
        b = G @ a
        res["gather"] = (G.lo, G.hi, float(np.linalg.norm(b.cpu().numpy()[rows] - ref) / np.linalg.norm(ref)), cg.get_info("last_dense_path"))
        # (2) the symmetric form: cyclic panels of the upper triangle per rank + ONE all-reduce
        Gs = cg.ShardedGramian(cg.EQ(), X, symmetric=True)
        assert Gs.sym_partial is not None
        bs = Gs @ a
        res["reduce"] = (float(np.linalg.norm(bs.cpu().numpy()[rows] - ref) / np.linalg.norm(ref)), cg.get_info("last_mfma_sym"))
        # (2b) the same in fp64 (the reference's default element type): cyclic 64-row blocks on the direct-difference symmetric kernel
        n6 = 6001
        X6h = rng.standard_normal((n6, d)); a6h = rng.standard_normal(n6)
        G6 = cg.ShardedGramian(cg.MaternP(2), torch.from_numpy(X6h).to(dev), symmetric=True)
        assert G6.sym_partial is not None
        b6 = G6 @ torch.from_numpy(a6h).to(dev)
        ref6 = o.mul(None, o.Kernel(o.MATERNP, p=2), X6h, X6h, a6h)
        res["reduce64"] = (float(np.linalg.norm(b6.cpu().numpy() - ref6) / np.linalg.norm(ref6)), cg.get_info("last_dense_sym"))
        # (3) fp64 MaternP(2), two point sets, matrix right-hand side (direct-difference kernel), and the gradient blocks
        m, p = 2111, 3
        Yh = rng.standard_normal((m, d)); Ah = rng.standard_normal((m, p)); X64 = rng.standard_normal((3001, d))
        Gm = cg.ShardedGramian(cg.MaternP(2), torch.from_numpy(X64).to(dev), torch.from_numpy(Yh).to(dev))
        B = Gm @ torch.from_numpy(Ah).to(dev)
        refm = o.mul(None, o.Kernel(o.MATERNP, p=2), X64, Yh, Ah)
        res["matrix"] = float(np.linalg.norm(B.cpu().numpy() - refm) / np.linalg.norm(refm))
        Xg = rng.standard_normal((333, 5)); ag = rng.standard_normal(333 * 5)
        Gg = cg.ShardedGramian(cg.GradientKernel(cg.EQ()), torch.from_numpy(Xg).to(dev))
        bg = Gg @ torch.from_numpy(ag).to(dev)
        refg = o.grad_mul(None, o.Kernel(o.EQ), Xg, Xg, ag)
        res["grad"] = float(np.linalg.norm(bg.cpu().numpy() - refg) / np.linalg.norm(refg))
        # (4) CG on the sharded operator: every rank must reach the same solution
        S = cg.ShardedGramian(cg.MaternP(1), torch.from_numpy(X64[:800]).to(dev))
        Aop = cg.LazyMatrixSum(S, 0.5 * torch.ones(800, dtype=torch.float64, device=dev))
        xs, info = cg.cg(Aop, torch.from_numpy(Ah[:800, 0].copy()).to(dev), reltol=1e-10, maxiter=400)
        Mh = o.matrix(o.Kernel(o.MATERNP, p=1), X64[:800]) + 0.5 * np.eye(800)
        res["cg"] = xs.cpu().numpy()
        res["cg_err"] = float(np.linalg.norm(res["cg"] - np.linalg.solve(Mh, Ah[:800, 0])) / np.linalg.norm(res["cg"]))
        res["cg_converged"] = bool(info["converged"])
        res["ok"] = True
    except Exception as e:      # report instead of hanging the other rank in a collective
        import traceback
        res["ok"] = False; res["error"] = traceback.format_exc()
    q.put((rank, res))
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        pass


def test_two_processes_share_a_gpu_over_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")                       # fresh children: a process that touched the GPU is never re-exec'ed
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(2):
        rank, res = q.get(timeout=600)
        out[rank] = res
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in (0, 1):
        assert out[r]["ok"], out[r].get("error")
    assert (out[0]["gather"][0], out[0]["gather"][1]) == (0, 15001) and (out[1]["gather"][0], out[1]["gather"][1]) == (15001, 30001)
    for r in (0, 1):
        assert out[r]["gather"][2] <= 1e-5 and out[r]["gather"][3] == 2, out[r]["gather"]       # the matrix-core kernel ran in each rank
        assert out[r]["reduce"][0] <= 1e-5 and out[r]["reduce"][1] == 1, out[r]["reduce"]         # the symmetric kernel ran in each rank
        assert out[r]["reduce64"][0] <= 1e-12 and out[r]["reduce64"][1] == 1, out[r]["reduce64"]  # ... and its fp64 counterpart
        assert out[r]["matrix"] <= 1e-12 and out[r]["grad"] <= 1e-12, (out[r]["matrix"], out[r]["grad"])
    assert out[0]["cg_converged"] and out[0]["cg_err"] <= 1e-8, out[0]["cg_err"]
    assert np.array_equal(out[0]["cg"], out[1]["cg"])                                             # replicated vectors stay bit-identical


def test_bench_runs_its_rccl_collectives_at_world_one():
    """bench.py with COVGRAM_FORCE_COLLECTIVE=1: backend "nccl" (= RCCL), world = 1, the all-gather of the contract run and the
    all-reduce of the symmetric variant are issued for real; the line must parse and both results must match the oracle."""
    env = dict(os.environ, COVGRAM_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-configs"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["unit"] == "MVM/s" and line["value"] > 50
    assert line["rel_err_vs_fp64_oracle"] <= 1e-5
    assert line["symmetric_variant"]["rel_err_vs_fp64_oracle"] <= 1e-5
    assert "row-shard" in line["config"]["parallelism"] or "1 GPU" in line["config"]["parallelism"]


def test_bench_gpus_one_with_no_environment_at_all():
    """`python3 bench.py --gpus 1` the way the driver types it — no WORLD_SIZE / RANK / MASTER_* in the environment: one process, no
    process group (rccl_ranks 0), the contract line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "COVGRAM_FORCE_COLLECTIVE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-configs"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 0 and line["value"] > 50 and line["rel_err_vs_fp64_oracle"] <= 1e-5
    assert line["roofline"]["peak"] == 157.3 and 0 < line["roofline"]["frac"] < 1 and line["roofline"]["issue_roofline_frac"] < 1


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_gpus_two_launches_two_rccl_ranks():
    """`python3 bench.py --gpus 2` with no environment: the launcher starts two ranks over RCCL; the line says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and len(line["per_rank"]) == 2 and line["rel_err_vs_fp64_oracle"] <= 1e-5


def test_bench_multi_rank_path_rehearsed_on_one_gpu():
    """Every line of bench.py's N > 1 path — row shards + the ONE all-gather per MVM, the symmetric variant's partials + all-reduce, the per-rank
    split gathered to rank 0, single_gpu_ms / ideal_ms — executed on the one GPU this box has: `--gpus 2` with COVGRAM_BENCH_REHEARSAL=1 (both ranks on GPU 0,
    collectives over gloo through host memory).  What is NOT covered is RCCL itself (tests/rccl_worker.py at world 1, two ranks where two GPUs exist).  The
    line must say that it is a rehearsal, carry two per-rank entries whose shards cover the rows, and both results must match the oracle."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["COVGRAM_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 0 and "rehearsal" in line
    assert line["rel_err_vs_fp64_oracle"] <= 1e-5 and line["symmetric_variant"]["rel_err_vs_fp64_oracle"] <= 1e-5
    assert [p["shard_rows"] for p in line["per_rank"]] == [65536, 65536] and all(p["kernel_ms"] > 0 and p["collective_ms"] > 0 for p in line["per_rank"])
    assert line["single_gpu_ms"] > 0 and abs(line["ideal_ms"] - line["single_gpu_ms"] / 2) < 1e-9
    assert "row-shard x2" in line["config"]["parallelism"]


def _run_rccl_worker(nproc):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "rccl_worker.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)       # fresh children of this process, nothing re-exec'ed
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_WORKER ")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0][len("RCCL_WORKER "):])


def _check_rccl(res, world):
    assert res["world"] == world and res["backend"] == "nccl"
    # the C-ABI communicator carried every collective of this run (include/covgram.h: covgram_comm_*, covgram_mvm_sharded, covgram_mvm_sym_allreduce)
    assert res["abi_route"] and res["comm_info"] == [0, world], (res["abi_route"], res["comm_info"])
    assert res["abi_mul"]["rel"] <= 1e-5 and res["abi_mul"]["rel_alpha_beta"] <= 1e-5, res["abi_mul"]
    assert res["abi_same_as_split"] and res["torch_route"] == {"abi": False, "same": True}, (res["abi_same_as_split"], res["torch_route"])
    assert res["reduce"]["abi"] and res["reduce"]["rel_alpha_beta"] <= 1e-5, res["reduce"]
    assert res["gather"]["rel"] <= 1e-5 and res["gather"]["steps"] == 1 and res["gather"]["collective_ms"] > 0, res["gather"]
    assert res["reduce"]["rel"] <= 1e-5 and res["reduce"]["used_partials"], res["reduce"]
    assert res["reduce64"]["rel"] <= 1e-12 and res["reduce64"]["used_partials"], res["reduce64"]
    assert res["matrix"] <= 1e-12 and res["grad"] <= 1e-12, (res["matrix"], res["grad"])
    assert res["replicated"]


def test_rccl_worker_at_world_one():
    """The worker script of the two-rank RCCL test below, on the one GPU every box has: backend "nccl", the all-gather and the all-reduce
    issued for real (world = 1), every result against the oracle — so the script itself is known good where two GPUs are missing."""
    _check_rccl(_run_rccl_worker(1), 1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: two ranks over RCCL / xGMI")
def test_rccl_two_ranks_row_shards_and_symmetric_partials():
    """Two ranks, one per GPU, backend "nccl" (= RCCL): row shards + ONE all-gather, cyclic-panel partials + ONE all-reduce, fp64 matrix
    right-hand sides and gradient blocks against the oracle; b identical on both ranks."""
    res = _run_rccl_worker(2)
    _check_rccl(res, 2)
    assert res["gather"]["shard"] == [0, 20001]


def test_c_abi_communicator_raw_calls_at_world_one(cg, oracle):
    """include/covgram.h's communicator through ctypes alone, the way the Julia shim's ccalls use it: unique id, covgram_comm_create on a ctx of
    its own (RCCL for real, world = 1), covgram_comm_info, the two raw collectives, covgram_mvm_sharded and covgram_mvm_sym_allreduce against the
    oracle, the error paths (no communicator; a second create), covgram_comm_destroy."""
    import ctypes as C
    o = oracle
    ffi = cg._ffi
    lib = ffi.lib()
    ctx = ffi._P()
    ffi.check(lib.covgram_ctx_create(C.byref(ctx), 0, ffi._P(torch.cuda.current_stream().cuda_stream)))
    try:
        n, d = 5000, 3
        rng = np.random.default_rng(11)
        Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
        X = torch.from_numpy(Xh).cuda(); a = torch.from_numpy(ah).cuda(); y = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
        px = ffi._P()
        ffi.check(lib.covgram_points_create(ctx, C.byref(px), ffi._P(X.data_ptr()), n, d, ffi.F32, ffi.DEVICE))
        k = ffi.covgram_kernel(ffi.EQ, ffi.ISOTROPIC, 0, 1, 0.0, 1.0, 1.0)
        rc = lib.covgram_mvm_sharded(ctx, ffi.kref(k), px, px, ffi._P(a.data_ptr()), ffi._P(y.data_ptr()), 1.0, 0.0)
        assert rc == ffi.EINVAL and b"no communicator" in lib.covgram_last_error()
        ident = (C.c_ubyte * ffi.COMM_ID_BYTES)()
        assert lib.covgram_comm_unique_id(C.cast(ident, C.c_void_p), 64) == ffi.EINVAL          # buffer too small
        ffi.check(lib.covgram_comm_unique_id(C.cast(ident, C.c_void_p), ffi.COMM_ID_BYTES))
        ffi.check(lib.covgram_comm_create(ctx, C.cast(ident, C.c_void_p), 0, 1))
        assert lib.covgram_comm_create(ctx, C.cast(ident, C.c_void_p), 0, 1) == ffi.EINVAL       # one communicator per ctx
        rk, wd = C.c_int32(-1), C.c_int32(-1)
        ffi.check(lib.covgram_comm_info(ctx, C.byref(rk), C.byref(wd)))
        assert (rk.value, wd.value) == (0, 1)
        ref = o.mul(None, o.Kernel(o.EQ), Xh.astype(np.float64), Xh.astype(np.float64), ah.astype(np.float64))
        ffi.check(lib.covgram_mvm_sharded(ctx, ffi.kref(k), px, px, ffi._P(a.data_ptr()), ffi._P(y.data_ptr()), 1.0, 0.0))
        ffi.check(lib.covgram_sync(ctx))
        b = y.cpu().numpy().astype(np.float64)
        assert np.isfinite(b).all() and np.linalg.norm(b - ref) / np.linalg.norm(ref) <= 1e-5
        y2 = torch.from_numpy(ah[::-1].copy()).cuda()
        ffi.check(lib.covgram_mvm_sharded(ctx, ffi.kref(k), px, px, ffi._P(a.data_ptr()), ffi._P(y2.data_ptr()), -0.7, 1.3))
        ffi.check(lib.covgram_sync(ctx))
        want = -0.7 * ref + 1.3 * ah[::-1].astype(np.float64)
        assert np.linalg.norm(y2.cpu().numpy() - want) / np.linalg.norm(want) <= 1e-5
        # the symmetric form: forced below its automatic size
        ffi.check(lib.covgram_ctx_set_option(ctx, b"mfma_sym", 1))
        y3 = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
        ffi.check(lib.covgram_mvm_sym_allreduce(ctx, ffi.kref(k), px, ffi._P(a.data_ptr()), ffi._P(y3.data_ptr()), 1.0, 0.0))
        ffi.check(lib.covgram_sync(ctx))
        assert np.linalg.norm(y3.cpu().numpy() - ref) / np.linalg.norm(ref) <= 1e-5
        # raw collectives: in-place all-gather of one piece, all-reduce of a vector (world = 1: identities, but RCCL runs them)
        v = torch.arange(1000, dtype=torch.float64, device="cuda"); v0 = v.clone()
        ffi.check(lib.covgram_comm_all_gather(ctx, ffi._P(v.data_ptr()), ffi._P(v.data_ptr()), 1000, ffi.F64))
        ffi.check(lib.covgram_comm_all_reduce_sum(ctx, ffi._P(v.data_ptr()), 1000, ffi.F64))
        ffi.check(lib.covgram_sync(ctx))
        assert torch.equal(v, v0)
        ffi.check(lib.covgram_points_destroy(px))
        ffi.check(lib.covgram_comm_destroy(ctx))
        ffi.check(lib.covgram_comm_info(ctx, C.byref(rk), C.byref(wd)))
        assert wd.value == 0
    finally:
        lib.covgram_ctx_destroy(ctx)
