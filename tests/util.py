"""shared helpers for the test-suite"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def asx():
    """the ctypes package over libaudiosync_hip.so (built on demand)"""
    if not os.path.exists(os.path.join(graft.PKG_DIR, "libaudiosync_hip.so")):
        graft.build()
    return graft.load()


def have_gpu():
    try:
        return asx().device_count() > 0
    except Exception:
        return False
