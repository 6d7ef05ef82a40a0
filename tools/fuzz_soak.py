"""Soak: tests/test_gpu_fuzz.py's dense / symmetric sweeps with other seeds (COVGRAM_FUZZ_OFFSET added to every seed); run as  pytest-free script on a GPU box."""
import os, sys, importlib, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import covgram as cg, covgram_oracle as oracle
import test_gpu_fuzz as t
off = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
orig = np.random.default_rng
fails = 0
for s in range(count):
    for fn in (t.test_random_dense_cases, t.test_random_symmetric_cases):
        np.random.default_rng = lambda seed=None, _o=orig, _k=off + s: _o((seed or 0) + 7919 * _k)
        try:
            fn.__wrapped__(cg, oracle, s) if hasattr(fn, "__wrapped__") else fn(cg, oracle, s)
        except AssertionError as e:
            fails += 1; print("FAIL", fn.__name__, off + s, str(e)[:300], flush=True)
        finally:
            np.random.default_rng = orig
print("soak done: seeds", off, "...", off + count - 1, "failures", fails)
