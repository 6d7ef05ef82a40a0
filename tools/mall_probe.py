"""Does a re-read of a buffer that fits the 256 MB Infinity Cache come back faster than HBM?  Repeated full reads (torch sum over fp32) of buffers of
32 MiB .. 1 GiB: effective GB/s of the steady state.  (Basis for the low-rank MVM's second pass over U, csrc/lowrank.hip.)"""
import torch
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for mib in (16, 32, 64, 96, 128, 160, 192, 256, 384, 512, 1024):
    x = torch.ones(mib * 262144, dtype=torch.float32, device="cuda")
    for _ in range(5): x.sum()
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): x.sum()
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"{mib:5d} MiB: {t * 1e6:7.1f} us per read, {mib * 1048576 / t * 1e-9:7.0f} GB/s", flush=True)
