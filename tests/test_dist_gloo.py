"""Multi-process CPU test of the row-sharded MVM: world_size 2 over gloo.  The local operator is
injected (the oracle, on CPU) so that the sharding + single all-gather logic of covgram.dist runs without a GPU."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, m, d, nrhs, q):
    sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import covgram as cg
    import covgram_oracle as o
    rng = np.random.default_rng(123)             # same seed on every rank: replicated inputs
    X = torch.from_numpy(rng.standard_normal((n, d))); Y = torch.from_numpy(rng.standard_normal((m, d)))
    a = torch.from_numpy(rng.standard_normal((m, nrhs) if nrhs > 1 else m))
    ko = o.Kernel(o.MATERNP, p=2)

    def factory(k, x_rows, y_full):              # CPU stand-in for the device Gramian
        return lambda vec: torch.from_numpy(o.mul(None, ko, x_rows.numpy(), y_full.numpy(), vec.numpy()))

    G = cg.ShardedGramian(cg.MaternP(2), X, Y, local_factory=factory)
    b = G @ a
    ref = o.mul(None, ko, X.numpy(), Y.numpy(), a.numpy())
    err = float(np.linalg.norm(b.numpy() - ref) / np.linalg.norm(ref))
    q.put((rank, G.lo, G.hi, tuple(b.shape), err))
    dist.barrier()
    dist.destroy_process_group()


def _grad_worker(rank, world, port, n, m, d, nrhs, q):
    """Row-sharded GradientKernel Gramian: flat point-major block vectors, rank g owns the blocks of its rows."""
    sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import covgram as cg
    import covgram_oracle as o
    rng = np.random.default_rng(321)
    X = torch.from_numpy(rng.standard_normal((n, d))); Y = torch.from_numpy(rng.standard_normal((m, d)))
    a = torch.from_numpy(rng.standard_normal(m * d))
    ko = o.Kernel(o.EQ)

    def factory(k, x_rows, y_full):
        return lambda vec: torch.from_numpy(o.grad_mul(None, ko, x_rows.numpy(), y_full.numpy(), vec.numpy()))

    G = cg.ShardedGramian(cg.GradientKernel(cg.EQ()), X, Y, local_factory=factory)
    b = G @ a
    ref = o.grad_mul(None, ko, X.numpy(), Y.numpy(), a.numpy())
    err = float(np.linalg.norm(b.numpy() - ref) / np.linalg.norm(ref))
    q.put((rank, G.lo, G.hi, tuple(b.shape), err))
    dist.barrier()
    dist.destroy_process_group()


def _sym_worker(rank, world, port, n, m, d, nrhs, q):
    """Symmetric form: every rank holds all of x, takes the cyclic panels p = rank (mod world) of the upper triangle (here
    panels of 4 rows, as a CPU stand-in for covgram_mvm_sym_partial) and ONE all-reduce completes b; a matrix right-hand side
    falls back to row shards + all-gather on the same object."""
    sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import covgram as cg
    import covgram_oracle as o
    rng = np.random.default_rng(77)
    X = torch.from_numpy(rng.standard_normal((n, d)))
    a = torch.from_numpy(rng.standard_normal(n)); A3 = torch.from_numpy(rng.standard_normal((n, 3)))
    ko = o.Kernel(o.EQ)
    calls = []

    def sym_factory(k, x):
        Gm = o.matrix(ko, x.numpy(), x.numpy())
        i = np.arange(x.shape[0])
        owner = (np.minimum(i[:, None], i[None, :]) // 4)                      # panel of the pair's upper-triangle entry
        def partial(out, vec, r, w):
            calls.append((r, w))
            out.copy_(torch.from_numpy((Gm * (owner % w == r)) @ vec.numpy()))
            return out
        return partial

    def factory(k, x_rows, y_full):
        return lambda vec: torch.from_numpy(o.mul(None, ko, x_rows.numpy(), y_full.numpy(), vec.numpy()))

    G = cg.ShardedGramian(cg.EQ(), X, local_factory=factory, sym_partial_factory=sym_factory)
    b = G @ a
    out = torch.full((n,), float("nan"), dtype=torch.float64); G.mul_(out, a)
    B3 = G @ A3                                                                 # several right-hand sides: row shards
    ref = o.mul(None, ko, X.numpy(), X.numpy(), a.numpy()); ref3 = o.mul(None, ko, X.numpy(), X.numpy(), A3.numpy())
    err = max(float(np.linalg.norm(b.numpy() - ref) / np.linalg.norm(ref)), float(np.linalg.norm(out.numpy() - ref) / np.linalg.norm(ref)),
              float(np.linalg.norm(B3.numpy() - ref3) / np.linalg.norm(ref3)))
    Gr = cg.ShardedGramian(cg.EQ(), X, local_factory=factory, sym_partial_factory=sym_factory, symmetric=False)
    err = max(err, float(np.linalg.norm((Gr @ a).numpy() - ref) / np.linalg.norm(ref)))
    q.put((rank, len(calls), calls[0] if calls else None, tuple(b.shape), err))
    dist.barrier()
    dist.destroy_process_group()


def _run(n, m, d, nrhs, world=2, worker=None):
    worker = worker or _worker
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, n, m, d, nrhs, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_row_sharded_mvm_world2_vector():
    res = _run(n=101, m=67, d=3, nrhs=1)
    assert [r[1:3] for r in res] == [(0, 51), (51, 101)]
    for r in res:
        assert r[3] == (101,) and r[4] < 1e-14


def test_row_sharded_mvm_world2_matrix_and_ragged():
    res = _run(n=3, m=40, d=2, nrhs=3)           # n < 2*ceil: second shard is short
    assert [r[1:3] for r in res] == [(0, 2), (2, 3)]
    for r in res:
        assert r[3] == (3, 3) and r[4] < 1e-14


def test_row_sharded_gradient_gramian_world2():
    res = _run(n=11, m=7, d=3, nrhs=1, worker=_grad_worker)      # ragged: shards of 6 and 5 points = 18 and 15 entries
    assert [r[1:3] for r in res] == [(0, 6), (6, 11)]
    for r in res:
        assert r[3] == (33,) and r[4] < 1e-13


def test_symmetric_cyclic_panels_allreduce_world2():
    res = _run(n=37, m=37, d=2, nrhs=1, worker=_sym_worker)
    for rank, (r, ncalls, first, shape, err) in enumerate(res):
        assert r == rank and ncalls == 2 and first == (rank, 2) and shape == (37,) and err < 1e-13


def _clean_env():
    """No torchrun variables: the way the driver (or a user) types `python3 bench.py --gpus N`."""
    return {k: v for k, v in os.environ.items()
            if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK", "TORCHELASTIC_RUN_ID")}


def test_bench_gpus_flag_launches_the_ranks_itself():
    """`python3 bench.py --gpus 2` with no WORLD_SIZE must start TWO ranks (round 3 parsed --gpus and never read it).
    --dry-launch runs exactly that launch path with backend gloo and no GPU work: two distinct processes meet in one all-gather and rank 0's
    single line comes back through the launcher."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=_clean_env(),
                       capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["gloo_ranks"] == 2 and line["ranks"] == [0, 1] and len(set(line["pids"])) == 2
    assert os.getpid() not in line["pids"]                      # the launcher itself is not a rank (it never touches a GPU)


def test_bench_refuses_a_world_that_is_not_gpus():
    """One process told it is one rank of one (WORLD_SIZE=1) while --gpus says 2: exit code 2, no JSON line — never a silent 1-GPU run
    labelled as N.  Likewise --gpus 2 on a box with fewer than two GPUs (this container has none)."""
    import subprocess
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1 but --gpus 2" in r.stderr and not r.stdout.strip()
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_clean_env(),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and "GPU(s) are visible" in r.stderr and not r.stdout.strip()
