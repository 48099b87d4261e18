"""companion of alloc_probe.py: FIXED inputs, the PLAN (its workspaces) re-created again and again -- does k_fwd_cols_r flip
with the workspaces' physical pages too?"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as g
asx = g.load()
n, batch = 1440000, 124
torch.cuda.set_device(0)
st = torch.cuda.Stream()
d_lag = torch.zeros(batch, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(batch, dtype=torch.float64, device="cuda"); d_ret = torch.zeros(batch, dtype=torch.int32, device="cuda")
a = torch.empty(batch * 2 * n, dtype=torch.float32, device="cuda"); b = torch.empty(batch * n, dtype=torch.float32, device="cuda")
asx.synth_pairs_dev(1, 0, batch, n, 1, a.data_ptr(), b.data_ptr(), 0, st.cuda_stream)
def run(plan):
    ps, pm = a.data_ptr(), b.data_ptr()
    for _ in range(30):
        plan.xcorr_batch_dev(ps, pm, batch, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), st.cuda_stream)
    plan.set_profiling(20)
    for _ in range(20):
        plan.xcorr_batch_dev(ps, pm, batch, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    rows = [plan.last_timings_ms(k) for k in range(20)]
    plan.set_profiling(0)
    return tuple(statistics.median(r[k] for r in rows) for k in ("fwd_cols", "rows", "inv_cols", "pearson", "total"))
for rep in range(8):
    with asx.Plan(n, batch, 0) as plan:
        print("plan %d: fwd %.3f rows %.3f inv %.3f pearson %.3f total %.3f" % ((rep,) + run(plan)), flush=True)
