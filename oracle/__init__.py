"""ctypes access to oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

The oracle is the CPU checker for the HIP path (see oracle/xcorr_oracle.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under old-audiosync_amd/ does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    """compile liboracle.so with the committed Makefile (gcc only)."""
    srcs = [os.path.join(_HERE, f) for f in ("fft64.c", "xcorr_oracle.c", "cpu_baseline.c", "fft64.h", "xcorr_oracle.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        fp = ctypes.POINTER(ctypes.c_float)
        L.oracle_max_abs_index.restype = ctypes.c_size_t
        L.oracle_max_abs_index.argtypes = [dp, ctypes.c_size_t]
        L.oracle_pearson_coefficient.restype = ctypes.c_double
        L.oracle_pearson_coefficient.argtypes = [dp, dp, dp, dp]
        L.oracle_cross_correlation.restype = ctypes.c_int
        L.oracle_cross_correlation.argtypes = [dp, dp, ctypes.c_size_t,
                                               ctypes.POINTER(ctypes.c_long), dp]
        L.oracle_cross_correlation_ex.restype = ctypes.c_int
        L.oracle_cross_correlation_ex.argtypes = [dp, dp, ctypes.c_size_t,
                                                  ctypes.POINTER(ctypes.c_long), dp, dp, dp]
        L.oracle_cross_correlation_f32.restype = ctypes.c_int
        L.oracle_cross_correlation_f32.argtypes = [fp, fp, ctypes.c_size_t,
                                                   ctypes.POINTER(ctypes.c_long), dp]
        L.oracle_synth_pair.restype = None
        L.oracle_synth_pair.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_size_t,
                                        ctypes.c_int, fp, fp, ctypes.POINTER(ctypes.c_int64)]
        L.oracle_cross_correlation_faithful.restype = ctypes.c_int
        L.oracle_cross_correlation_faithful.argtypes = [dp, dp, ctypes.c_size_t,
                                                        ctypes.POINTER(ctypes.c_long), dp]
        L.oracle_baseline_backend.restype = ctypes.c_char_p
        L.oracle_worker_create.restype = ctypes.c_void_p
        L.oracle_worker_create.argtypes = [ctypes.c_size_t]
        L.oracle_worker_run.restype = ctypes.c_int
        L.oracle_worker_run.argtypes = [ctypes.c_void_p, dp, dp, ctypes.POINTER(ctypes.c_long), dp]
        L.oracle_worker_destroy.restype = None
        L.oracle_worker_destroy.argtypes = [ctypes.c_void_p]
        L.offt_rfft.restype = ctypes.c_int
        L.offt_rfft.argtypes = [ctypes.c_size_t, dp, ctypes.c_void_p]
        L.offt_irfft.restype = ctypes.c_int
        L.offt_irfft.argtypes = [ctypes.c_size_t, ctypes.c_void_p, dp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def max_abs_index(arr):
    a = np.ascontiguousarray(arr, dtype=np.float64)
    return int(lib().oracle_max_abs_index(_d(a), a.size))


def pearson_coefficient(source_seg, sample_seg):
    """both segments as arrays of equal length (the [start,end) pointer pairs of the C API)."""
    a = np.ascontiguousarray(source_seg, dtype=np.float64)
    b = np.ascontiguousarray(sample_seg, dtype=np.float64)
    assert a.size == b.size
    pa, pb = _d(a), _d(b)
    ea = ctypes.cast(ctypes.addressof(pa.contents) + 8 * a.size, ctypes.POINTER(ctypes.c_double))
    eb = ctypes.cast(ctypes.addressof(pb.contents) + 8 * b.size, ctypes.POINTER(ctypes.c_double))
    return float(lib().oracle_pearson_coefficient(pa, ea, pb, eb))


def cross_correlation(source, sample, want_results=False):
    """-> (ret, lag, coefficient[, r, peak_margin]); inputs are widened to float64."""
    s = np.ascontiguousarray(source, dtype=np.float64)
    t = np.ascontiguousarray(sample, dtype=np.float64)
    n = t.size
    assert s.size == 2 * n
    lag = ctypes.c_long(0)
    coef = ctypes.c_double(0.0)
    if not want_results:
        ret = lib().oracle_cross_correlation(_d(s), _d(t), n, ctypes.byref(lag), ctypes.byref(coef))
        return ret, lag.value, coef.value
    r = np.empty(2 * n, dtype=np.float64)
    margin = ctypes.c_double(0.0)
    ret = lib().oracle_cross_correlation_ex(_d(s), _d(t), n, ctypes.byref(lag), ctypes.byref(coef),
                                            _d(r), ctypes.byref(margin))
    return ret, lag.value, coef.value, r, margin.value


def baseline_backend():
    """"fftw3" when libfftw3.so.3 could be dlopen()ed on this node, else "port" (oracle/fft64.c)"""
    return lib().oracle_baseline_backend().decode()


def cross_correlation_faithful(source, sample):
    """one call with the reference's cost model (two threads, plan per call, allocations per call)"""
    s = np.ascontiguousarray(source, dtype=np.float64)
    t = np.ascontiguousarray(sample, dtype=np.float64)
    assert s.size == 2 * t.size
    lag = ctypes.c_long(0)
    coef = ctypes.c_double(0.0)
    ret = lib().oracle_cross_correlation_faithful(_d(s), _d(t), t.size, ctypes.byref(lag), ctypes.byref(coef))
    return ret, lag.value, coef.value


class Worker:
    """an independent per-core worker (bench.py's node-throughput leg): one thread, plans and buffers kept between calls"""

    def __init__(self, sample_len):
        self.n = int(sample_len)
        self._w = lib().oracle_worker_create(self.n)
        if not self._w:
            raise MemoryError("oracle_worker_create(%d)" % self.n)

    def run(self, source, sample):
        s = np.ascontiguousarray(source, dtype=np.float64)
        t = np.ascontiguousarray(sample, dtype=np.float64)
        assert t.size == self.n and s.size == 2 * self.n
        lag = ctypes.c_long(0)
        coef = ctypes.c_double(0.0)
        ret = lib().oracle_worker_run(self._w, _d(s), _d(t), ctypes.byref(lag), ctypes.byref(coef))
        return ret, lag.value, coef.value

    def close(self):
        if self._w:
            lib().oracle_worker_destroy(self._w)
            self._w = None

    def __del__(self):
        self.close()


def synth_pair(seed, pair, n, noise_shift=1):
    """-> (source float32[2n], sample float32[n], true_lag)"""
    src = np.empty(2 * n, dtype=np.float32)
    smp = np.empty(n, dtype=np.float32)
    lag = ctypes.c_int64(0)
    lib().oracle_synth_pair(seed, pair, n, noise_shift, _f(src), _f(smp), ctypes.byref(lag))
    return src, smp, lag.value


def rfft(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(x.size // 2 + 1, dtype=np.complex128)
    rc = lib().offt_rfft(x.size, _d(x), out.ctypes.data)
    assert rc == 0
    return out


def irfft_unnormalised(X, L):
    X = np.ascontiguousarray(X, dtype=np.complex128)
    assert X.size == L // 2 + 1
    out = np.empty(L, dtype=np.float64)
    rc = lib().offt_irfft(L, X.ctypes.data, _d(out))
    assert rc == 0
    return out
