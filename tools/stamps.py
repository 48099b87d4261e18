#!/usr/bin/env python3
"""diagnostic: per-phase cycle shares of k_rows (needs a -DASX_STAMPS build; ASX_STAMPS=1 env)."""
import ctypes, os, sys
import numpy as np
os.environ["ASX_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
asx = g.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1440000
split = sys.argv[2] if len(sys.argv) > 2 else None
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
cap = 8 * 16 * 4096
buf = np.zeros(cap, dtype=np.uint64)
got = L.asx_plan_debug_stamps(plan._h, buf.ctypes.data, cap)
s = buf[:got].reshape(-1, 8).astype(np.int64)
s = s[s[:, 5] > 0]
d = np.diff(s[:, :6], axis=1)
names = ["setup+load", "fwd fft", "combine", "inv fft", "store"]
tot = (s[:, 5] - s[:, 0])
print("blocks", len(s), "split", plan.split, "threads", plan.threads)
print("median block cycles", int(np.median(tot)))
for i, nm in enumerate(names):
    print("%-12s median %7d  share %.1f%%" % (nm, np.median(d[:, i]), 100 * np.median(d[:, i]) / np.median(tot)))
print("kernel span cycles", int(s[:, 5].max() - s[:, 0].min()))
