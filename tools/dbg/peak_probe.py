"""diagnostic: the peak-search state for sin(i) inputs (run on the GPU box)"""
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
d_src = torch.from_numpy(src).cuda(); d_smp = torch.from_numpy(smp).cuda()
d_r = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
d_lag = torch.zeros(1, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(1, dtype=torch.float64, device="cuda"); d_ret = torch.zeros(1, dtype=torch.int32, device="cuda")
plan.debug_r_dev(d_src.data_ptr(), d_smp.data_ptr(), d_r.data_ptr(), d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr()); plan.sync()
r32 = d_r.cpu().numpy().astype(np.float64)
F = plan.fft_len
b2 = ctypes.c_float(); cn = ctypes.c_uint32(); rn = ctypes.c_uint32(); pm = ctypes.c_uint64()
cap = plan.peak_capacity
vals = np.zeros(cap); idxs = np.zeros(cap, dtype=np.uint32)
L.asx_plan_debug_peak.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
rc = L.asx_plan_debug_peak(plan._h, 0, ctypes.byref(b2), ctypes.byref(cn), ctypes.byref(rn), ctypes.byref(pm), vals.ctypes.data, idxs.ctypes.data, cap)
k32 = np.abs(r32); k32[0] = r32[0]
print("rc", rc, "lag dev", int(d_lag[0]), "oracle", o_lag, "kstar", kstar, "margin-1", margin - 1)
print("bound2", b2.value, "cand_n", cn.value, "refine_n", rn.value, "cap", cap, "pairmax key idx", hex(pm.value))
print("max32", k32.max(), "argmax32", int(k32.argmax()), "key32[kstar]", k32[kstar], "diff", k32.max() - k32[kstar])
nx = np.linalg.norm(src.astype(np.float64)); ny = np.linalg.norm(smp.astype(np.float64))
print("expected bound2", 2 * 4 * 2.0**-24 * np.log2(F) * nx * ny, " (x F: ", 2 * 4 * 2.0**-24 * np.log2(F) * nx * ny * F, ")")
print("max err r32 vs oracle (scaled by F):", np.abs(r32 / F - o_r / (2 * n)).max() * F)
nr = rn.value
if nr:
    sel = idxs[:nr]; print("kstar in refine list:", kstar in set(sel.tolist()))
    exact = o_r[sel] / (2 * n)
    print("max |dots - oracle|/|oracle|:", np.abs(vals[:nr] - exact).max() / np.abs(exact).max())
    j = int(np.argmax(np.abs(vals[:nr]))); print("dots argmax idx", int(sel[j]))
