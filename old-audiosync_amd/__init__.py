"""old-audiosync_amd — MI355X (gfx950) implementation of ONE path of
vidify/old-audiosync: the FFT cross-correlation of src/cross_correlation.c.

The product is native: csrc/ (hand-written HIP kernels behind the C-ABI of
include/audiosync/xcorr_hip.h -> libaudiosync_hip.so) and host/ (plain C, the
reference's own API -> libaudiosync.so, and the CPython module `audiosync`).
This Python package is only a thin ctypes view of that C-ABI for tests, the
benchmark and notebooks.  There is no CPU fallback: loading fails loudly when
the library has not been built, and every call fails when no GPU works.

The directory name contains a hyphen, so import it with `load()` from
__graft_entry__.py (importlib), e.g.  `asx = __graft_entry__.load()`.
"""
from .hipxcorr import (  # noqa: F401
    LIB_PATH, AsxError, Plan, Stream, abi_version, device_count, lib, pearson_f64, planmath_candidates, planmath_describe,
    planmath_table, planmath_twiddles, results_to_ms_dev, synth_pairs_dev, xcorr_batch_multi,
    Comm, PinnedArray, result_bytes, shard_range,
)
