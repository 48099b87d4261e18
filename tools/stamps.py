#!/usr/bin/env python3
"""diagnostic: per-phase cycle shares of one transform kernel (needs a -DASX_STAMPS build).
usage: stamps.py [N] [split|-] [rows|fwd|inv]"""
import ctypes, os, sys
import numpy as np
which = sys.argv[3] if len(sys.argv) > 3 else "rows"
os.environ["ASX_STAMPS"] = {"rows": "1", "fwd": "fwd", "inv": "inv"}[which]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
asx = g.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1440000
split = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
batch = 16
d_src = torch.empty(batch * 2 * n, dtype=torch.float32, device="cuda")
d_smp = torch.empty(batch * n, dtype=torch.float32, device="cuda")
d_lag = torch.zeros(batch, dtype=torch.int64, device="cuda")
d_coef = torch.zeros(batch, dtype=torch.float64, device="cuda")
d_ret = torch.zeros(batch, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
asx.synth_pairs_dev(1, 0, batch, n, 1, d_src.data_ptr(), d_smp.data_ptr(), 0, st)
plan = asx.Plan(n, batch, 0, split=split)
for _ in range(3):
    plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), batch, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), st)
torch.cuda.synchronize()
L = asx.lib()
L.asx_plan_debug_stamps.restype = ctypes.c_long
L.asx_plan_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
cap = 8 * 16 * 8192
buf = np.zeros(cap, dtype=np.uint64)
got = L.asx_plan_debug_stamps(plan._h, buf.ctypes.data, cap)
s = buf[:got].reshape(-1, 8).astype(np.int64)
if which == "rows":
    names, last = ["setup+load", "fwd fft", "combine", "inv fft", "store"], 5
elif which == "fwd":
    names, last = ["load tile", "fft", "store tile"], 3
else:
    names, last = ["load tile", "fft", "peak scan"], 3
s = s[s[:, last] > 0]
d = np.diff(s[:, :last + 1], axis=1)
tot = (s[:, last] - s[:, 0])
print("kernel", which, "blocks", len(s), "split", plan.split, "threads", plan.threads)
print("median block cycles", int(np.median(tot)))
for i, nm in enumerate(names):
    print("%-12s median %7d  share %.1f%%" % (nm, np.median(d[:, i]), 100 * np.median(d[:, i]) / np.median(tot)))
