#!/usr/bin/env python3
"""tools/dbg/double_abi_lengths.py [reps] -- cross_correlation(double*) wall time per call (PCIe and host narrowing included) at the
reference's six interval lengths, pageable caller buffers, frames exactly float32 (GPU box; same-box A/B of host_narrow.cpp builds)."""
import ctypes, os, sys, time, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
import __graft_entry__ as g
g.load()
L = ctypes.CDLL(os.path.join(g.PKG_DIR, "libaudiosync.so"))
dp = ctypes.POINTER(ctypes.c_double)
L.cross_correlation.restype = ctypes.c_int
L.cross_correlation.argtypes = [dp, dp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_long), dp]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(3)
out = {}
for n in (144000, 288000, 480000, 720000, 960000, 1440000):
    src = rng.uniform(-1, 1, 2 * n).astype(np.float32).astype(np.float64)
    smp = (0.5 * src[1234:1234 + n] + 0.1 * rng.uniform(-1, 1, n)).astype(np.float32).astype(np.float64)
    lag, coef = ctypes.c_long(0), ctypes.c_double(0)
    call = lambda: L.cross_correlation(src.ctypes.data_as(dp), smp.ctypes.data_as(dp), n, ctypes.byref(lag), ctypes.byref(coef))
    for _ in range(3): assert call() == 0 and lag.value == 1234
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
    out[n] = round(statistics.median(ts), 4)
print(out)
