"""ctypes view of libaudiosync_hip.so (the C-ABI in include/audiosync/xcorr_hip.h)."""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaudiosync_hip.so")
_lib = None

c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_intp = ctypes.POINTER(ctypes.c_int)

# every symbol include/audiosync/xcorr_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "asx_device_count", "asx_last_error", "asx_abi_version", "asx_plan_create", "asx_plan_create_ex", "asx_plan_destroy",
    "asx_plan_sample_len", "asx_plan_fft_len", "asx_plan_split", "asx_plan_threads", "asx_plan_group",
    "asx_plan_workspace_bytes", "asx_xcorr_f64", "asx_xcorr_batch_f32", "asx_xcorr_batch_f32_dev",
    "asx_xcorr_debug_r_dev", "asx_pearson_f64", "asx_results_to_ms_dev", "asx_stream_create", "asx_stream_destroy",
    "asx_stream_append_f64", "asx_stream_lengths", "asx_stream_reset", "asx_stream_xcorr", "asx_synth_pairs_dev", "asx_plan_set_profiling",
    "asx_plan_last_timings_ms", "asx_device_malloc", "asx_device_free", "asx_memcpy_h2d",
    "asx_memcpy_d2h", "asx_stream_sync", "asx_plan_peak_overflows", "asx_plan_peak_repairs", "asx_plan_set_exact", "asx_plan_peak_capacity",
    "asx_current_device", "asx_plan_timings_ms", "asx_xcorr_batch_multi", "asx_plan_layout", "asx_plan_narrowed_calls",
    "asx_plan_set_pearson", "asx_plan_pearson_modes", "asx_plan_placement", "asx_host_malloc", "asx_host_free", "asx_shard_range", "asx_result_bytes", "asx_comm_create", "asx_comm_destroy", "asx_xcorr_batch_multi_dev",
]


class AsxError(RuntimeError):
    pass


def _one_hip_runtime():
    """PyTorch's ROCm wheel ships its own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP
    runtimes in one process do not work, so when torch is installed it is imported FIRST: the
    loader then binds this library's libamdhip64.so.7 dependency to the copy torch already
    loaded, and torch tensors / streams can be handed to the C-ABI as raw pointers.
    ASX_NO_TORCH=1 skips this (pure C / ctypes users of the system runtime)."""
    if os.environ.get("ASX_NO_TORCH") == "1" or "torch" in sys.modules:
        return
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def lib():
    """dlopen the HIP layer; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AsxError(LIB_PATH + " is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _one_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    vp = ctypes.c_void_p
    L.asx_device_count.restype = ctypes.c_int
    L.asx_last_error.restype = ctypes.c_char_p
    L.asx_abi_version.restype = ctypes.c_int
    L.asx_plan_create.restype = vp
    L.asx_plan_create.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
    L.asx_plan_create_ex.restype = vp
    L.asx_plan_create_ex.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p]
    L.asx_plan_destroy.restype = None
    L.asx_plan_destroy.argtypes = [vp]
    L.asx_plan_peak_overflows.restype = ctypes.c_int
    L.asx_plan_peak_overflows.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.asx_plan_set_exact.restype = ctypes.c_int
    L.asx_plan_set_exact.argtypes = [vp, ctypes.c_int]
    L.asx_plan_placement.restype = ctypes.c_int
    L.asx_plan_placement.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    L.asx_plan_set_pearson.restype = ctypes.c_int
    L.asx_plan_set_pearson.argtypes = [vp, ctypes.c_int]
    L.asx_plan_pearson_modes.restype = ctypes.c_int
    L.asx_plan_pearson_modes.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.asx_plan_peak_repairs.restype = ctypes.c_int
    L.asx_plan_peak_repairs.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.asx_plan_narrowed_calls.restype = ctypes.c_int
    L.asx_plan_narrowed_calls.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    for name in ("asx_plan_sample_len", "asx_plan_fft_len", "asx_plan_group", "asx_plan_workspace_bytes",
                 "asx_plan_peak_capacity"):
        getattr(L, name).restype = ctypes.c_size_t
        getattr(L, name).argtypes = [vp]
    L.asx_plan_threads.restype = ctypes.c_int
    L.asx_plan_threads.argtypes = [vp, c_intp, c_intp]
    L.asx_plan_split.restype = ctypes.c_int
    L.asx_plan_split.argtypes = [vp, c_intp, c_intp, c_intp]
    L.asx_plan_layout.restype = ctypes.c_int
    L.asx_plan_layout.argtypes = [vp]
    L.asx_xcorr_f64.restype = ctypes.c_int
    L.asx_xcorr_f64.argtypes = [vp, c_f64p, c_f64p, ctypes.POINTER(ctypes.c_long), c_f64p]
    L.asx_xcorr_batch_f32.restype = ctypes.c_int
    L.asx_xcorr_batch_f32.argtypes = [vp, c_f32p, c_f32p, ctypes.c_size_t, c_i64p, c_f64p, c_i32p]
    L.asx_xcorr_batch_multi.restype = ctypes.c_int
    L.asx_xcorr_batch_multi.argtypes = [ctypes.POINTER(vp), ctypes.c_int, c_f32p, c_f32p, ctypes.c_size_t, c_i64p, c_f64p, c_i32p]
    L.asx_xcorr_batch_f32_dev.restype = ctypes.c_int
    L.asx_xcorr_batch_f32_dev.argtypes = [vp, vp, vp, ctypes.c_size_t, vp, vp, vp, vp]
    L.asx_shard_range.restype = ctypes.c_int
    L.asx_shard_range.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                                  ctypes.POINTER(ctypes.c_size_t)]
    L.asx_result_bytes.restype = ctypes.c_size_t
    L.asx_result_bytes.argtypes = [ctypes.c_size_t]
    L.asx_comm_create.restype = vp
    L.asx_comm_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    L.asx_comm_destroy.restype = None
    L.asx_comm_destroy.argtypes = [vp]
    L.asx_xcorr_batch_multi_dev.restype = ctypes.c_int
    L.asx_xcorr_batch_multi_dev.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t),
                                            ctypes.c_size_t, ctypes.POINTER(vp)]
    L.asx_xcorr_debug_r_dev.restype = ctypes.c_int
    L.asx_xcorr_debug_r_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.asx_pearson_f64.restype = ctypes.c_int
    L.asx_pearson_f64.argtypes = [c_f64p, c_f64p, ctypes.c_size_t, ctypes.c_int, c_f64p]
    L.asx_results_to_ms_dev.restype = ctypes.c_int
    L.asx_results_to_ms_dev.argtypes = [vp, vp, vp, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, vp, vp, vp]
    L.asx_stream_create.restype = vp
    L.asx_stream_create.argtypes = [ctypes.c_size_t, ctypes.c_int]
    L.asx_stream_destroy.restype = None
    L.asx_stream_destroy.argtypes = [vp]
    L.asx_stream_append_f64.restype = ctypes.c_int
    L.asx_stream_append_f64.argtypes = [vp, c_f64p, ctypes.c_size_t, c_f64p, ctypes.c_size_t]
    L.asx_stream_lengths.restype = ctypes.c_int
    L.asx_stream_lengths.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    L.asx_stream_reset.restype = ctypes.c_int
    L.asx_stream_reset.argtypes = [vp]
    L.asx_stream_xcorr.restype = ctypes.c_int
    L.asx_stream_xcorr.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_long), c_f64p]
    L.asx_synth_pairs_dev.restype = ctypes.c_int
    L.asx_synth_pairs_dev.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_size_t,
                                      ctypes.c_int, vp, vp, vp, vp]
    L.asx_plan_set_profiling.restype = ctypes.c_int
    L.asx_plan_set_profiling.argtypes = [vp, ctypes.c_int]
    L.asx_plan_last_timings_ms.restype = ctypes.c_int
    L.asx_plan_last_timings_ms.argtypes = [vp, c_f32p]
    L.asx_plan_timings_ms.restype = ctypes.c_int
    L.asx_plan_timings_ms.argtypes = [vp, ctypes.c_int, c_f32p]
    L.asx_device_malloc.restype = vp
    L.asx_device_malloc.argtypes = [ctypes.c_size_t, ctypes.c_int]
    L.asx_device_free.restype = ctypes.c_int
    L.asx_device_free.argtypes = [vp]
    L.asx_host_malloc.restype = vp
    L.asx_host_malloc.argtypes = [ctypes.c_size_t]
    L.asx_host_free.restype = ctypes.c_int
    L.asx_host_free.argtypes = [vp]
    L.asx_memcpy_h2d.restype = ctypes.c_int
    L.asx_memcpy_h2d.argtypes = [vp, vp, ctypes.c_size_t]
    L.asx_memcpy_d2h.restype = ctypes.c_int
    L.asx_memcpy_d2h.argtypes = [vp, vp, ctypes.c_size_t]
    L.asx_stream_sync.restype = ctypes.c_int
    L.asx_stream_sync.argtypes = [vp, vp]
    # planning arithmetic (host only)
    L.asx_planmath_describe.restype = ctypes.c_int
    L.asx_planmath_describe.argtypes = [ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32),
                                        ctypes.POINTER(ctypes.c_uint32), c_intp, c_intp, c_intp, c_intp,
                                        c_intp, c_intp, c_intp]
    L.asx_planmath_table.restype = ctypes.c_int
    L.asx_planmath_table.argtypes = [ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int, c_intp, ctypes.c_size_t]
    L.asx_planmath_twiddles.restype = ctypes.c_int
    L.asx_planmath_twiddles.argtypes = [ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int, c_f32p, ctypes.c_size_t]
    _lib = L
    return L


def _err():
    return lib().asx_last_error().decode("utf-8", "replace")


def abi_version():
    return lib().asx_abi_version()


def device_count():
    return lib().asx_device_count()


def _split_arg(split):
    return split.encode() if split else None


def planmath_describe(sample_len, split=None):
    """host-only: what plan would be built for sample_len (no GPU needed)."""
    F, valid = ctypes.c_uint32(), ctypes.c_uint32()
    m1, m2, t, n1, n2 = (ctypes.c_int() for _ in range(5))
    r1 = (ctypes.c_int * 16)()
    r2 = (ctypes.c_int * 16)()
    rc = lib().asx_planmath_describe(sample_len, _split_arg(split), F, valid, m1, m2, t, n1, r1, n2, r2)
    if rc != 0:
        raise AsxError(_err())
    return {"F": F.value, "src_valid": valid.value, "M1": m1.value, "M2": m2.value, "T": t.value,
            "radix1": list(r1[: n1.value]), "radix2": list(r2[: n2.value])}


def planmath_candidates(sample_len, max_count=16):
    """host-only: the splits the measured mode (split="measure") would time, cheapest first."""
    buf = ctypes.create_string_buffer(64 * max_count + 1)
    L = lib()
    L.asx_planmath_candidates.restype = ctypes.c_int
    L.asx_planmath_candidates.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    n = L.asx_planmath_candidates(sample_len, max_count, buf, len(buf))
    if n < 0:
        raise AsxError(_err())
    return [x for x in buf.value.decode().split("\n") if x]


def planmath_table(sample_len, which, split=None):
    d = planmath_describe(sample_len, split)
    cap = max(d["M1"], d["M2"])
    buf = (ctypes.c_int * cap)()
    n = lib().asx_planmath_table(sample_len, _split_arg(split), which, buf, cap)
    if n < 0:
        raise AsxError(_err())
    return np.array(buf[:n], dtype=np.int64)


def planmath_twiddles(sample_len, which, split=None):
    d = planmath_describe(sample_len, split)
    cap = max(d["M1"], d["M2"], 2048, (d["F"] >> 11) + 2)
    buf = np.zeros(2 * cap, dtype=np.float32)
    n = lib().asx_planmath_twiddles(sample_len, _split_arg(split), which,
                                    buf.ctypes.data_as(c_f32p), cap)
    if n < 0:
        raise AsxError(_err())
    return buf[: 2 * n].view(np.complex64).copy()


def pearson_f64(source_seg, sample_seg, device=-1):
    a = np.ascontiguousarray(source_seg, dtype=np.float64)
    b = np.ascontiguousarray(sample_seg, dtype=np.float64)
    assert a.size == b.size
    out = ctypes.c_double(0.0)
    rc = lib().asx_pearson_f64(a.ctypes.data_as(c_f64p), b.ctypes.data_as(c_f64p), a.size, device,
                               ctypes.byref(out))
    if rc != 0:
        raise AsxError(_err())
    return out.value


def results_to_ms_dev(d_lag, d_coef, d_ret, batch, d_lag_ms, d_accept=0, min_confidence=0.95, sample_rate=48000.0, stream=0):
    rc = lib().asx_results_to_ms_dev(d_lag, d_coef, d_ret, batch, min_confidence, sample_rate, d_lag_ms,
                                     d_accept or None, stream or None)
    if rc != 0:
        raise AsxError(_err())


def synth_pairs_dev(seed, first_pair, count, sample_len, noise_shift, d_src, d_smp, d_lag=0, stream=0):
    rc = lib().asx_synth_pairs_dev(seed, first_pair, count, sample_len, noise_shift, d_src, d_smp,
                                   d_lag or None, stream or None)
    if rc != 0:
        raise AsxError(_err())


def xcorr_batch_multi(plans, source, sample):
    """host float32 arrays [B,2N], [B,N] block-partitioned over `plans` (one per device) -> (lag, coef, ret)"""
    s = np.ascontiguousarray(source, dtype=np.float32)
    t = np.ascontiguousarray(sample, dtype=np.float32)
    n = plans[0].sample_len
    batch = t.size // n
    assert t.size == batch * n and s.size == 2 * n * batch
    lag = np.zeros(batch, dtype=np.int64)
    coef = np.zeros(batch, dtype=np.float64)
    ret = np.zeros(batch, dtype=np.int32)
    handles = (ctypes.c_void_p * len(plans))(*[p._h for p in plans])
    rc = lib().asx_xcorr_batch_multi(handles, len(plans), s.ctypes.data_as(c_f32p), t.ctypes.data_as(c_f32p), batch,
                                     lag.ctypes.data_as(c_i64p), coef.ctypes.data_as(c_f64p), ret.ctypes.data_as(c_i32p))
    if rc != 0:
        raise AsxError(_err())
    return lag, coef, ret


class PinnedArray:
    """a numpy float64 array in page-locked host memory (asx_host_malloc): what audiosync_run() keeps its tracks in"""

    def __init__(self, count, dtype=np.float64):
        self.nbytes = int(count) * np.dtype(dtype).itemsize
        self._p = lib().asx_host_malloc(self.nbytes)
        if not self._p:
            raise AsxError(_err())
        buf = (ctypes.c_char * self.nbytes).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(count))

    def close(self):
        if self._p:
            self.array = None
            lib().asx_host_free(self._p)
            self._p = None


def shard_range(total, nshards, shard):
    """block partition of a batch over shards (asx_shard_range): -> (start, count)"""
    start, count = ctypes.c_size_t(0), ctypes.c_size_t(0)
    if lib().asx_shard_range(total, nshards, shard, ctypes.byref(start), ctypes.byref(count)) != 0:
        raise AsxError(_err())
    return start.value, count.value


def result_bytes(width):
    return int(lib().asx_result_bytes(width))


class Comm:
    """One RCCL communicator over the devices of `plans` (one plan per device), created inside the library
    (ncclCommInitAll); `run` = asx_xcorr_batch_multi_dev: device-resident shards, one all-gather of the result records."""

    def __init__(self, plans):
        self.plans = list(plans)
        handles = (ctypes.c_void_p * len(self.plans))(*[p._h for p in self.plans])
        self._h = lib().asx_comm_create(handles, len(self.plans))
        if not self._h:
            raise AsxError(_err())

    def run(self, d_sources, d_samples, counts, width, d_gathered):
        n = len(self.plans)
        assert len(d_sources) == len(d_samples) == len(counts) == len(d_gathered) == n
        src = (ctypes.c_void_p * n)(*d_sources)
        smp = (ctypes.c_void_p * n)(*d_samples)
        out = (ctypes.c_void_p * n)(*d_gathered)
        cnt = (ctypes.c_size_t * n)(*counts)
        if lib().asx_xcorr_batch_multi_dev(self._h, src, smp, cnt, width, out) != 0:
            raise AsxError(_err())

    def close(self):
        if self._h:
            lib().asx_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Stream:
    """asx_stream: both tracks resident in HBM, new frames appended, one plan per prefix length."""

    def __init__(self, max_sample_len, device=-1):
        self._h = lib().asx_stream_create(int(max_sample_len), int(device))
        if not self._h:
            raise AsxError(_err())
        self._destroy = lib().asx_stream_destroy  # bound now: module globals may be gone at interpreter exit

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    def append(self, source_frames, sample_frames):
        s = np.ascontiguousarray(source_frames, dtype=np.float64)
        t = np.ascontiguousarray(sample_frames, dtype=np.float64)
        rc = lib().asx_stream_append_f64(self._h, s.ctypes.data_as(c_f64p), s.size, t.ctypes.data_as(c_f64p), t.size)
        if rc != 0:
            raise AsxError(_err())

    def lengths(self):
        a, b = ctypes.c_size_t(), ctypes.c_size_t()
        lib().asx_stream_lengths(self._h, a, b)
        return a.value, b.value

    def reset(self):
        lib().asx_stream_reset(self._h)

    def xcorr(self, sample_len):
        lag = ctypes.c_long(0)
        coef = ctypes.c_double(0.0)
        ret = lib().asx_stream_xcorr(self._h, int(sample_len), ctypes.byref(lag), ctypes.byref(coef))
        return ret, lag.value, coef.value


class Plan:
    """asx_plan: fixed sample_len, owns tables + HBM workspaces on one device."""

    def __init__(self, sample_len, max_batch=1, device=-1, split=None):
        self._h = lib().asx_plan_create_ex(int(sample_len), int(max_batch), int(device),
                                           _split_arg(split))
        if not self._h:
            raise AsxError(_err())
        self._destroy = lib().asx_plan_destroy  # bound now: module globals may be gone at interpreter exit
        self.sample_len = int(sample_len)

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def fft_len(self):
        return lib().asx_plan_fft_len(self._h)

    @property
    def group(self):
        return lib().asx_plan_group(self._h)

    @property
    def peak_capacity(self):
        return lib().asx_plan_peak_capacity(self._h)

    def peak_overflows(self):
        """pairs (since plan creation) with more near-tied lags than the plan re-evaluates exactly"""
        c = ctypes.c_uint64(0)
        if lib().asx_plan_peak_overflows(self._h, ctypes.byref(c)) != 0:
            raise AsxError(_err())
        return c.value

    def set_exact(self, on=True):
        """on (the default): every entry point takes the second look at overflowing pairs (the device-resident batch waits
        once per call for its kernels); off: that entry point stays asynchronous and marks such pairs with ret = 1"""
        if lib().asx_plan_set_exact(self._h, 1 if on else 0) != 0:
            raise AsxError(_err())

    def placement(self):
        """("measure" plans) -> ((ms of the forward column kernel on the first / second allocation of its workspaces), kept)"""
        ms = (ctypes.c_double * 2)()
        kept = ctypes.c_int(-1)
        if lib().asx_plan_placement(self._h, ms, ctypes.byref(kept)) != 0:
            raise AsxError(_err())
        return (float(ms[0]), float(ms[1])), int(kept.value)

    def set_pearson(self, spectral=True):
        """spectral (the default on real-column plans): the coefficient from r[peak] and the forward pass's band sums, no second
        pass over the inputs; False: the reference's reduction over both segments (include/audiosync/xcorr_hip.h)"""
        if lib().asx_plan_set_pearson(self._h, 1 if spectral else 0) != 0:
            raise AsxError(_err())

    def pearson_modes(self):
        """pairs so far that took (spectral, spectral + wrap-around correction, direct) under the spectral setting"""
        c = (ctypes.c_uint64 * 3)()
        if lib().asx_plan_pearson_modes(self._h, c) != 0:
            raise AsxError(_err())
        return tuple(int(v) for v in c)

    def narrowed_calls(self):
        """xcorr_f64 calls whose frames were all exactly float32 and crossed PCIe as 4 bytes each"""
        c = ctypes.c_uint64(0)
        if lib().asx_plan_narrowed_calls(self._h, ctypes.byref(c)) != 0:
            raise AsxError(_err())
        return c.value

    def peak_repairs(self):
        """overflowing pairs that were looked at again with lists for all 2N lags"""
        c = ctypes.c_uint64(0)
        if lib().asx_plan_peak_repairs(self._h, ctypes.byref(c)) != 0:
            raise AsxError(_err())
        return c.value

    @property
    def workspace_bytes(self):
        return lib().asx_plan_workspace_bytes(self._h)

    @property
    def split(self):
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib().asx_plan_split(self._h, a, b, c)
        return a.value, b.value, c.value

    @property
    def layout(self):
        """'real-column' (csrc/rlayout.hip) or 'packed' (csrc/xcorr_kernels.hip): which decomposition this plan runs"""
        return "real-column" if lib().asx_plan_layout(self._h) == 1 else "packed"

    @property
    def threads(self):
        a, b = ctypes.c_int(), ctypes.c_int()
        lib().asx_plan_threads(self._h, a, b)
        return a.value, b.value

    def xcorr_f64(self, source, sample):
        """the reference's calling convention: -> (ret, lag, coefficient)"""
        s = np.ascontiguousarray(source, dtype=np.float64)
        t = np.ascontiguousarray(sample, dtype=np.float64)
        assert t.size == self.sample_len and s.size == 2 * self.sample_len
        lag = ctypes.c_long(0)
        coef = ctypes.c_double(0.0)
        ret = lib().asx_xcorr_f64(self._h, s.ctypes.data_as(c_f64p), t.ctypes.data_as(c_f64p),
                                  ctypes.byref(lag), ctypes.byref(coef))
        return ret, lag.value, coef.value

    def xcorr_batch_f32(self, source, sample):
        """host float32 arrays [B,2N], [B,N] -> (lag int64[B], coef float64[B], ret int32[B])"""
        s = np.ascontiguousarray(source, dtype=np.float32)
        t = np.ascontiguousarray(sample, dtype=np.float32)
        n = self.sample_len
        batch = t.size // n
        assert t.size == batch * n and s.size == 2 * n * batch
        lag = np.zeros(batch, dtype=np.int64)
        coef = np.zeros(batch, dtype=np.float64)
        ret = np.zeros(batch, dtype=np.int32)
        rc = lib().asx_xcorr_batch_f32(self._h, s.ctypes.data_as(c_f32p), t.ctypes.data_as(c_f32p), batch,
                                       lag.ctypes.data_as(c_i64p), coef.ctypes.data_as(c_f64p),
                                       ret.ctypes.data_as(c_i32p))
        if rc != 0:
            raise AsxError(_err())
        return lag, coef, ret

    def xcorr_batch_dev(self, d_src, d_smp, batch, d_lag, d_coef, d_ret, stream=0):
        """raw device pointers (ints); asynchronous on `stream` (0 = the plan's own)"""
        rc = lib().asx_xcorr_batch_f32_dev(self._h, d_src, d_smp, batch, d_lag or None, d_coef,
                                           d_ret or None, stream or None)
        if rc != 0:
            raise AsxError(_err())

    def debug_r_dev(self, d_src, d_smp, d_r, d_lag, d_coef, d_ret, stream=0):
        rc = lib().asx_xcorr_debug_r_dev(self._h, d_src, d_smp, d_r, d_lag, d_coef, d_ret, stream or None)
        if rc != 0:
            raise AsxError(_err())

    def set_profiling(self, depth):
        """depth > 0: keep the kernel events of the last `depth` batch calls; 0 / False: off"""
        lib().asx_plan_set_profiling(self._h, int(depth))

    def last_timings_ms(self, calls_back=0):
        out = (ctypes.c_float * 6)()
        rc = lib().asx_plan_timings_ms(self._h, int(calls_back), out)
        if rc != 0:
            raise AsxError(_err())
        return dict(zip(("fwd_cols", "rows", "inv_cols", "finalize", "pearson", "total"), list(out)))

    def sync(self, stream=0):
        rc = lib().asx_stream_sync(self._h, stream or None)
        if rc != 0:
            raise AsxError(_err())
