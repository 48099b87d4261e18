"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol
include/audiosync/xcorr_hip.h declares, the host C library exports the reference's
own API, and (without a GPU) the product fails loudly instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

from util import ROOT, asx, graft, have_gpu


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", "audiosync", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(asx_[a-z0-9_]+)\s*\(", text)))


def test_build_and_load():
    mod = graft.build()
    assert mod.abi_version() == 2


def test_every_declared_symbol_is_exported():
    mod = asx()
    L = mod.lib()
    declared = declared_symbols("xcorr_hip.h")
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), name
    from audiosync_amd import hipxcorr
    assert sorted(hipxcorr.ABI_SYMBOLS) == declared


def test_reference_api_symbols_exported_by_host_library():
    L = ctypes.CDLL(os.path.join(graft.PKG_DIR, "libaudiosync.so"))
    # include/audiosync/cross_correlation.h:10-11,24-25 and audiosync.h:51-119 of the reference
    for name in ("cross_correlation", "pearson_coefficient", "audiosync_run", "audiosync_abort",
                 "audiosync_pause", "audiosync_resume", "audiosync_status", "audiosync_setup",
                 "audiosync_get_debug", "audiosync_set_debug", "status_to_string", "global_status",
                 "global_debug", "mutex", "interval_done", "read_continue", "audiosync_set_feed"):
        assert hasattr(L, name), name


def test_no_oracle_or_fallback_in_product_sources():
    # the product must never route through oracle/ or a CPU implementation
    for base, _, files in os.walk(graft.PKG_DIR):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp")):
                text = open(os.path.join(base, f)).read()
                assert "liboracle" not in text and "import oracle" not in text, f
                assert "from oracle" not in text, f
                assert not re.search(r'#\s*include\s*[<"][^>"]*(xcorr_oracle|fft64)', text), f


@pytest.mark.skipif(have_gpu(), reason="only meaningful on a box without a GPU")
def test_fails_loudly_without_gpu():
    mod = asx()
    assert mod.device_count() == 0
    with pytest.raises(mod.AsxError):
        mod.Plan(1000)
    with pytest.raises(mod.AsxError):
        mod.pearson_f64([1.0, 2.0], [2.0, 1.0])
    # the reference API reports failure the way the reference reports allocation failure
    L = ctypes.CDLL(os.path.join(graft.PKG_DIR, "libaudiosync.so"))
    L.cross_correlation.restype = ctypes.c_int
    src = np.zeros(10)
    smp = np.zeros(5)
    lag = ctypes.c_long(77)
    coef = ctypes.c_double(0.5)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = L.cross_correlation(src.ctypes.data_as(dp), smp.ctypes.data_as(dp), ctypes.c_size_t(5),
                             ctypes.byref(lag), ctypes.byref(coef))
    assert rc == -1 and lag.value == 77 and coef.value == 0.5  # outputs untouched
