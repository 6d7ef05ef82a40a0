"""One rank of the RCCL check (tests/test_gpu_dist.py): launched by `python -m torch.distributed.run`, one process per GPU, backend "nccl"
(= RCCL on ROCm).  Row shards + ONE all-gather, symmetric partials + ONE all-reduce, an fp64 matrix right-hand side and the gradient
blocks, each against rows of the fp64 oracle; rank 0 prints one JSON line.  Not collected by pytest (no test_ prefix)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    world = int(os.environ["WORLD_SIZE"]); rank = int(os.environ["RANK"]); local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("COVGRAM_FORCE_COLLECTIVE", "1")       # world = 1 still issues the collectives
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import covgram as cg
    import covgram_oracle as o
    res = {"world": world, "backend": dist.get_backend()}
    rng = np.random.default_rng(77)                          # same seed on every rank: replicated inputs
    n, d = 40001, 3
    Xh = rng.standard_normal((n, d)).astype(np.float32); ah = rng.standard_normal(n).astype(np.float32)
    X = torch.from_numpy(Xh).to(dev); a = torch.from_numpy(ah).to(dev)
    rows = np.random.default_rng(1).choice(n, 256, replace=False)
    ref = o.mul(None, o.Kernel(o.EQ), Xh[rows], Xh, ah, dtype=np.float32)
    rel = lambda b, r: float(np.linalg.norm(np.asarray(b, dtype=np.float64) - r) / np.linalg.norm(r))
    G = cg.ShardedGramian(cg.EQ(), X, symmetric=False)
    # the route under RCCL is the C ABI's (round 5): covgram_mvm_sharded = the shard's kernels + ncclAllGather on the library's stream
    res["abi_route"] = bool(G._abi)
    rk, wd = C.c_int32(-1), C.c_int32(-1)
    cg._ffi.check(cg._ffi.lib().covgram_comm_info(cg.get_ctx(dev).handle, C.byref(rk), C.byref(wd)))
    res["comm_info"] = [rk.value, wd.value]
    b1 = G @ a                                                   # ONE library call: kernel + collective
    y2 = torch.from_numpy(ah[::-1].copy()).to(dev)
    G.mul_(y2, a, -0.7, 1.3)                                     # alpha / beta through covgram_mvm_sharded
    res["abi_mul"] = {"rel": rel(b1.cpu().numpy()[rows], ref), "rel_alpha_beta": rel(y2.cpu().numpy()[rows], -0.7 * ref + 1.3 * ah[::-1][rows].astype(np.float64))}
    G.timing = True
    b = G @ a                                                    # timing on: the shard's MVM, then covgram_comm_all_gather, bracketed by events
    loc, col, steps = G.timing_ms()
    res["gather"] = {"rel": rel(b.cpu().numpy()[rows], ref), "shard": [G.lo, G.hi], "local_ms": loc, "collective_ms": col, "steps": steps}
    res["abi_same_as_split"] = bool(torch.equal(b, b1))
    # ... and torch.distributed's own collective (COVGRAM_ABI_COLLECTIVE=0) gives the same bits
    os.environ["COVGRAM_ABI_COLLECTIVE"] = "0"
    Gt = cg.ShardedGramian(cg.EQ(), X, symmetric=False)
    res["torch_route"] = {"abi": bool(Gt._abi), "same": bool(torch.equal(Gt @ a, b1))}
    os.environ.pop("COVGRAM_ABI_COLLECTIVE")
    Gs = cg.ShardedGramian(cg.EQ(), X, symmetric=True)
    res["reduce"] = {"rel": rel((Gs @ a).cpu().numpy()[rows], ref), "used_partials": Gs.sym_partial is not None, "abi": bool(Gs._abi)}
    y3 = torch.from_numpy(ah[::-1].copy()).to(dev)
    Gs.mul_(y3, a, 2.0, -0.5)                                    # covgram_mvm_sym_allreduce with alpha / beta
    res["reduce"]["rel_alpha_beta"] = rel(y3.cpu().numpy()[rows], 2.0 * ref - 0.5 * ah[::-1][rows].astype(np.float64))
    n6 = 6001
    X6h = rng.standard_normal((n6, d)); a6h = rng.standard_normal(n6)
    G6 = cg.ShardedGramian(cg.MaternP(2), torch.from_numpy(X6h).to(dev), symmetric=True)
    res["reduce64"] = {"rel": rel((G6 @ torch.from_numpy(a6h).to(dev)).cpu().numpy(), o.mul(None, o.Kernel(o.MATERNP, p=2), X6h, X6h, a6h)),
                       "used_partials": G6.sym_partial is not None}
    m, p = 2111, 3
    Yh = rng.standard_normal((m, d)); Ah = rng.standard_normal((m, p)); X64 = rng.standard_normal((3001, d))
    Gm = cg.ShardedGramian(cg.MaternP(2), torch.from_numpy(X64).to(dev), torch.from_numpy(Yh).to(dev))
    res["matrix"] = rel((Gm @ torch.from_numpy(Ah).to(dev)).cpu().numpy(), o.mul(None, o.Kernel(o.MATERNP, p=2), X64, Yh, Ah))
    Xg = rng.standard_normal((333, 5)); ag = rng.standard_normal(333 * 5)
    Gg = cg.ShardedGramian(cg.GradientKernel(cg.EQ()), torch.from_numpy(Xg).to(dev))
    res["grad"] = rel((Gg @ torch.from_numpy(ag).to(dev)).cpu().numpy(), o.grad_mul(None, o.Kernel(o.EQ), Xg, Xg, ag))
    # every rank holds the same complete b: compare a checksum across ranks
    chk = torch.tensor([float(b.double().sum())], dtype=torch.float64, device=dev)
    lo = chk.clone(); hi = chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    res["replicated"] = bool(lo.item() == hi.item())
    dist.barrier()
    if rank == 0:
        print("RCCL_WORKER " + json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
