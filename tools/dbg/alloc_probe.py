"""does the allocator of the INPUT buffers change k_fwd_cols_r's time?  N processes-worth of samples inside one process:
allocate inputs (torch caching allocator / raw hipMalloc via the library / one big block for both), time 40 steps, free."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as g
asx = g.load()
L = asx.lib()
n, batch = 1440000, 124
torch.cuda.set_device(0)
st = torch.cuda.Stream()
d_lag = torch.zeros(batch, dtype=torch.int64, device="cuda"); d_coef = torch.zeros(batch, dtype=torch.float64, device="cuda"); d_ret = torch.zeros(batch, dtype=torch.int32, device="cuda")
plan = asx.Plan(n, batch, 0)
def run(ps, pm):
    asx.synth_pairs_dev(1, 0, batch, n, 1, ps, pm, 0, st.cuda_stream)
    for _ in range(30):
        plan.xcorr_batch_dev(ps, pm, batch, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), st.cuda_stream)
    plan.set_profiling(20)
    for _ in range(20):
        plan.xcorr_batch_dev(ps, pm, batch, d_lag.data_ptr(), d_coef.data_ptr(), d_ret.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    rows = [plan.last_timings_ms(b) for b in range(20)]
    plan.set_profiling(0)
    return statistics.median(r["fwd_cols"] for r in rows), statistics.median(r["pearson"] for r in rows), statistics.median(r["total"] for r in rows)
for rep in range(4):
    a = torch.empty(batch * 2 * n, dtype=torch.float32, device="cuda"); b = torch.empty(batch * n, dtype=torch.float32, device="cuda")
    print("torch.empty x2      fwd %.3f pearson %.3f total %.3f" % run(a.data_ptr(), b.data_ptr()), flush=True)
    del a, b; torch.cuda.empty_cache()
    pa = L.asx_device_malloc(batch * 2 * n * 4, 0); pb = L.asx_device_malloc(batch * n * 4, 0)
    print("hipMalloc x2        fwd %.3f pearson %.3f total %.3f" % run(pa, pb), flush=True)
    L.asx_device_free(ctypes.c_void_p(pa)); L.asx_device_free(ctypes.c_void_p(pb))
    pc = L.asx_device_malloc(batch * 3 * n * 4 + (2 << 20), 0)
    print("hipMalloc one block fwd %.3f pearson %.3f total %.3f" % run(pc, pc + batch * 2 * n * 4), flush=True)
    print("  ... sample 2 MiB later fwd %.3f pearson %.3f total %.3f" % run(pc, pc + batch * 2 * n * 4 + (2 << 20)), flush=True)
    L.asx_device_free(ctypes.c_void_p(pc))
