"""tools/dbg/stamps_r.py [rows|inv] [N] -- per-phase cycle shares of the real-column kernels (needs a -DASX_STAMPS build of
csrc/rlayout.hip: tools/mkr.sh r_stamps "-DASX_STAMPS", copied over libaudiosync_hip.so)"""
import ctypes, os, sys
import numpy as np
which = sys.argv[1] if len(sys.argv) > 1 else "rows"
os.environ["ASX_STAMPS"] = {"rows": "1", "inv": "inv"}[which]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as g
asx = g.load()
n, batch = (int(sys.argv[2]) if len(sys.argv) > 2 else 1440000), 64
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
if which == "rows":
    seq = [0, 1, 2, 3, 4]
    names = ["loads + twiddles + fill", "forward stage 0", "wave-local chain (fwd 1, 2, product, inv 2, 1)", "inverse stage 0 (+ last radix-2) + stores"]
    s = s[(s[:, 4] > 0) & (s[:, 0] > 0)]
else:
    seq = [0, 1, 2, 4, 5, 3]
    names = ["loads + tangling + first butterflies", "barrier + stage 1", "last stage + per-thread scan", "wave max, barrier, fold, atomicMax", "threshold + candidates"]
    s = s[(s[:, 3] > 0) & (s[:, 4] > 0) & (s[:, 5] > 0)]
print("kernel", which, "blocks", len(s), "split", plan.split, "layout", plan.layout)
tot = np.median(s[:, seq[-1]] - s[:, seq[0]])
print("median block cycles (from the first stamp)", int(tot))
for i, nm in enumerate(names):
    d = s[:, seq[i + 1]] - s[:, seq[i]]
    print("%-50s median %7d  share %.1f%%" % (nm, np.median(d), 100 * np.median(d) / tot))

if which == "rows":
    pro = s[:, 0] - s[:, 6]
    wall = (s[:, 5] - s[:, 7])                      # 100 MHz ticks, entry to exit
    cyc = s[:, 4] - s[:, 6]
    print("prologue (entry -> first stamp): median %d cycles; entry -> exit: %d cycles = %.2f us wall -> %.2f GHz" %
          (np.median(pro), np.median(cyc), np.median(wall) / 100.0, np.median(cyc) / (np.median(wall) * 10.0)))
    # how densely blocks follow each other: sort by entry wall time; throughput = blocks / (span) against residency
    w0 = np.sort(s[:, 7]); span = (w0[-1] - w0[0]) / 100.0
    print("blocks %d in %.1f us wall: %.3f us per block per chip; with R resident blocks per CU x 256 CUs a block slot turns over every R x 256 x that" % (len(s), span, span / len(s)))
