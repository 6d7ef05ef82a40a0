"""The contract kernel (C2: EQ, d = 3, n = 131072, fp32, all entries): column split (option jsplit; 0 = automatic), interleaved; us per MVM."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "covariancefunctions.jl_amd"))
import covgram as cg
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
n = 131072
rng = np.random.default_rng(0xC0F + 1)
X = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).cuda(); a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
cg.set_option("mfma_sym", 0)
G = cg.gramian(cg.EQ(), X); y = torch.empty(n, dtype=torch.float32, device="cuda")
res = {}
for rep in range(3):
    for js in (0, 4, 6, 8, 12, 16, 24, 32):
        cg.set_option("jsplit", js)
        for _ in range(4): G.mul_(y, a)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): G.mul_(y, a)
        e1.record(); e1.synchronize()
        if rep: res.setdefault(js, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        if js == 0 and rep == 0: auto = cg.get_info("last_jsplit")
cg.set_option("jsplit", 0); cg.set_option("mfma_sym", -1)
print(f"C2 general (automatic split {auto}): " + "  ".join(f"jsplit={k or 'auto'}: {min(v):.1f}" for k, v in res.items()), flush=True)
