"""diagnostic: the peak-search state of the FAST path (batched entry point) for sin(i) inputs (run on the GPU box)"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as g, oracle
asx = g.load()
L = asx.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 144000
i = np.arange(2 * n, dtype=np.float64)
src = np.sin(i).astype(np.float32); smp = np.sin(i[:n]).astype(np.float32)
o_ret, o_lag, o_coef, o_r, margin = oracle.cross_correlation(src, smp, want_results=True)
key = np.abs(o_r); key[0] = o_r[0]; kstar = int(np.argmax(key))
plan = asx.Plan(n, 1, 0)
m1, m2, T = plan.split
lag, coef, ret = plan.xcorr_batch_f32(src[None], smp[None])
b2 = ctypes.c_float(); cn = ctypes.c_uint32(); rn = ctypes.c_uint32(); pm = ctypes.c_uint64()
cap = plan.peak_capacity
vals = np.zeros(cap); idxs = np.zeros(cap, dtype=np.uint32)
L.asx_plan_debug_peak.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
rc = L.asx_plan_debug_peak(plan._h, 0, ctypes.byref(b2), ctypes.byref(cn), ctypes.byref(rn), ctypes.byref(pm), vals.ctypes.data, idxs.ctypes.data, cap)
print("lag dev", int(lag[0]), "oracle", o_lag, "kstar", kstar, "tile of kstar", (kstar % m2) // T, "margin-1", margin - 1)
print("bound2", b2.value, "cand_n", cn.value, "refine_n", rn.value, "cap", cap, "pairmax", hex(pm.value), "idx", 0xFFFFFFFF - (pm.value & 0xFFFFFFFF))
nr = rn.value
sel = idxs[:nr]
print("kstar in refine list:", kstar in set(sel.tolist()))
tiles = sorted(set(((sel % m2) // T).tolist()))
print("tiles with refine entries:", tiles)
# which lags SHOULD be candidates (float64 view: within 2B of the max, generous)
B2 = b2.value / plan.fft_len
near = np.nonzero(key >= key.max() - B2)[0]
print("lags within 2B of the float64 max:", near.size, "tiles:", sorted(set(((near % m2) // T).tolist())))
