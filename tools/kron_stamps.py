"""In-kernel timeline of the fused Kronecker pass (KRON_DIAG build, COVGRAM_KRON_STAMPS=<file>): per wave s_memtime stamps
0 = prologue loads issued, 1 = first barrier passed, then per step G: 2+2G = work of the step done (before the barrier), 3+2G = barrier passed;
30 = output stored.  Prints medians over workgroups, in shader cycles relative to stamp 0, for one wave of each half.
usage: COVGRAM_LIB=.../lib_diag/libcovgram.so COVGRAM_KRON_STAMPS=/tmp/st.bin python tools/kron_stamps.py [kron64|kron32]"""
import os, sys, numpy as np, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.environ["COVGRAM_KRON_STAMPS"]
which = sys.argv[1] if len(sys.argv) > 1 else "kron64"
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_run.py"), which, "5"], check=True, stdout=subprocess.DEVNULL)
st = np.fromfile(path, dtype=np.int64).reshape(-1, 8, 32)
st = st[st[:, 0, 0] != 0]
rel = st - st[:, :1, :1].min(axis=(1, 2), keepdims=True)      # relative to the workgroup's earliest stamp 0
print(f"{st.shape[0]} workgroups; median shader cycles since the workgroup's first stamp (wave 0 = half 0, wave 4 = half 1)")
print(" idx   wave0   wave4   | step time (barrier to barrier, wave 0)")
prev = None
for i in range(32):
    if i in (28, 29) or not (st[:, 0, i] != 0).any(): continue
    m0, m4 = np.median(rel[:, 0, i]), np.median(rel[:, 4, i])
    extra = ""
    if i >= 1 and i % 2 == 1:
        if prev is not None: extra = f"   {m0 - prev:8.0f}"
        prev = m0
    print(f"{i:4d} {m0:8.0f} {m4:8.0f}{extra}")
span = (st[:, :, 30].max(axis=1) - st[:, :, 0].min(axis=1))
print(f"workgroup lifetime stamp 0 -> 30: median {np.median(span):.0f} cycles, max {span.max():.0f}")
# absolute picture: s_memrealtime (100 MHz, one counter for the chip) at stamp 0 (slot 28) and at the end (slot 29)
r0 = st[:, :, 28].min(axis=1); r1 = st[:, :, 29].max(axis=1)
print(f"real time (10 ns ticks): workgroup starts spread over {(r0.max() - r0.min()) * 0.01:.2f} us; first start -> last end {(r1.max() - r0.min()) * 0.01:.2f} us; "
      f"median workgroup {np.median(r1 - r0) * 0.01:.2f} us")
