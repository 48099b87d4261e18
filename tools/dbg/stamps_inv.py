import ctypes, os, sys
import numpy as np
os.environ["ASX_STAMPS"] = "inv"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as g
asx = g.load()
n, batch = 1440000, 64
d_src = torch.empty(batch * 2 * n, dtype=torch.float32, device="cuda"); d_smp = torch.empty(batch * n, dtype=torch.float32, device="cuda")
d_lag = torch.zeros(batch, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(batch, dtype=torch.float64, device="cuda"); d_ret = torch.zeros(batch, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
asx.synth_pairs_dev(1, 0, batch, n, 1, d_src.data_ptr(), d_smp.data_ptr(), 0, st)
plan = asx.Plan(n, batch, 0)
for _ in range(3):
    plan.xcorr_batch_dev(d_src.data_ptr(), d_smp.data_ptr(), batch, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), st)
torch.cuda.synchronize()
L = asx.lib()
L.asx_plan_debug_stamps.restype = ctypes.c_long
L.asx_plan_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
cap = 8 * 64 * 8192
buf = np.zeros(cap, dtype=np.uint64)
got = L.asx_plan_debug_stamps(plan._h, buf.ctypes.data, cap)
s = buf[:got].reshape(-1, 8).astype(np.int64)
s = s[(s[:, 3] > 0) & (s[:, 4] > 0) & (s[:, 5] > 0)]
print("blocks", len(s))
# order of stamps in time: 0 start, 1 (fed begin), 2 head done, 4 after last stage + per-thread scan, 5 after reduce/fold, 3 end
seq = [0, 1, 2, 4, 5, 3]
names = ["setup", "fed stage + stage 1 (head)", "last stage + per-thread scan", "wave max, barrier, fold, atomicMax", "threshold + candidates"]
tot = np.median(s[:, 3] - s[:, 0])
print("median block cycles", int(tot))
for i, nm in enumerate(names):
    d = s[:, seq[i + 1]] - s[:, seq[i]]
    print("%-40s median %7d  share %.1f%%" % (nm, np.median(d), 100 * np.median(d) / tot))
