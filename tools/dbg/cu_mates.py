"""tools/dbg/cu_mates.py [fwd|invhw] -- which blocks of a column kernel's launch share a CU, and when they start (needs tools/experiments/column_kernels_hw_stamps.patch applied and a -DASX_STAMPS build:
tools/mkfull.sh r_stamps "-DASX_STAMPS").  Prints, for the first generation of blocks, how the launch-order index of CU mates relates."""
import ctypes, os, sys, collections
import numpy as np
which = sys.argv[1] if len(sys.argv) > 1 else "invhw"
os.environ["ASX_STAMPS"] = which
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as g
asx = g.load()
n, batch = 1440000, 124
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
cap = 8 * 80000
buf = np.zeros(cap, dtype=np.uint64)
got = L.asx_plan_debug_stamps(plan._h, buf.ctypes.data, cap)
s = buf[:got].reshape(-1, 8).astype(np.int64)
hw, t0 = s[:, 6], s[:, 7]
ok = t0 > 0
idx = np.nonzero(ok)[0]
print("blocks with a stamp:", len(idx))
cu = ((hw >> 16) & 0xF) * 4096 + (hw & 0xFF00)          # xcc, se, sh, cu
tmin = t0[ok].min()
by = collections.defaultdict(list)
for i in idx: by[int(cu[i])].append((int(t0[i] - tmin), int(i)))
print("CUs seen:", len(by))
first = []
for c, v in by.items():
    v.sort()
    first.append(v[:4])
# the first two blocks of every CU: launch indices, parity, start offset
same_par = sum(1 for v in first if len(v) > 1 and (v[0][1] & 1) == (v[1][1] & 1))
d256 = sum(1 for v in first if len(v) > 1 and abs(v[0][1] - v[1][1]) == 256)
d8 = sum(1 for v in first if len(v) > 1 and abs(v[0][1] - v[1][1]) == 8)
both_lt512 = sum(1 for v in first if len(v) > 1 and v[0][1] < 512 and v[1][1] < 512)
print("first two blocks of a CU: same parity on %d of %d CUs; indices 256 apart on %d, 8 apart on %d; both < 512 on %d" % (same_par, len(first), d256, d8, both_lt512))
gap = [v[1][0] - v[0][0] for v in first if len(v) > 1]
print("start gap between them [us]: median %.2f, 10%% %.2f, 90%% %.2f" % (np.median(gap) / 100, np.percentile(gap, 10) / 100, np.percentile(gap, 90) / 100))
for v in first[:12]: print([(t / 100.0, i) for t, i in v])
