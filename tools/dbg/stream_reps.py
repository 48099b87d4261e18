"""diagnostic: per-repetition interval times of the growing-window run (bench.py --mode streaming averages them)"""
import sys, os, time
import numpy as np
ROOT = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import __graft_entry__ as graft
import oracle
asx = graft.load()
torch.cuda.set_device(0)
sr = 48000; n_max = 30 * sr
src32, smp32, _ = oracle.synth_pair(20260101, 0, n_max, 1)
true_lag = 12345
rng = np.random.default_rng(1)
smp32 = (0.5 * src32[true_lag: true_lag + n_max] + 0.25 * rng.uniform(-1, 1, n_max)).astype(np.float32)
pin_src, pin_smp = asx.PinnedArray(2 * n_max), asx.PinnedArray(n_max)
pin_src.array[:] = src32.astype(np.float64); pin_smp.array[:] = smp32.astype(np.float64)
src, smp = pin_src.array, pin_smp.array
seconds = (3, 6, 10, 15, 20, 30)
st = asx.Stream(n_max, 0)
for s_ in seconds:
    st.append(src[st.lengths()[0]: 2 * s_ * sr], smp[st.lengths()[1]: s_ * sr]); st.xcorr(s_ * sr)
for rep in range(8):
    st.reset(); row = []
    for s_ in seconds:
        n = s_ * sr
        t0 = time.perf_counter()
        a, b = st.lengths()
        st.append(src[a: 2 * n], smp[b: n])
        t1 = time.perf_counter()
        ret, lag, coef = st.xcorr(n)
        t2 = time.perf_counter()
        row.append("%.3f+%.3f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
        assert ret == 0 and lag == true_lag
    print(rep, " ".join(row))
