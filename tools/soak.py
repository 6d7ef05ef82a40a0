"""One-off soak: run the GPU fuzz tests over many more seeds than the committed parametrisation (usage: soak.py [first] [count])."""
import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import covgram as cg, covgram_oracle as oracle
fz = importlib.import_module("test_gpu_fuzz")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100; count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for seed in range(first, first + count):
    for fn in (fz.test_random_dense_cases, fz.test_random_symmetric_cases, fz.test_random_gradient_cases):
        try:
            fn.__wrapped__(cg, oracle, seed) if hasattr(fn, "__wrapped__") else fn(cg, oracle, seed)
        except AssertionError as e:
            bad += 1; print("FAIL", fn.__name__, seed, str(e)[:300], flush=True)
print("soak done:", count, "seeds x 3 tests,", bad, "failures")
