""""existing callers compile and link unchanged".  The reference's own test programs
(/root/reference/tests/test_cross_correlation.c, test_pearson_coefficient.c) are compiled UNMODIFIED against this repo's
include/ (audiosync/audiosync.h:12-119, audiosync/cross_correlation.h:10-11,24-25) and linked against libaudiosync.so.
Compile + link only: running them needs a GPU (tests/test_gpu_capi.py runs the same twelve known answers from C through
tests/c/kat_runner.c).  Nothing of the reference is copied or shipped: the sources are read where they lie and the programs
land in a temporary directory.  Skipped where /root/reference does not exist (the GPU box)."""
import os
import shutil
import subprocess
import tempfile

import pytest

from util import asx, graft

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALLERS = ["tests/test_cross_correlation.c", "tests/test_pearson_coefficient.c"]


@pytest.mark.parametrize("rel", CALLERS)
def test_reference_test_program_compiles_and_links_against_this_library(rel):
    src = os.path.join(REF, rel)
    if not os.path.exists(src):
        pytest.skip("no reference checkout here")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    asx()  # builds libaudiosync.so / libaudiosync_hip.so if they are not there
    lib_dir = graft.PKG_DIR
    assert os.path.exists(os.path.join(lib_dir, "libaudiosync.so"))
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "caller")
        # the flags of the reference's own build (CMakeLists.txt: C99, -Wall -Wextra) with ITS include directory replaced by ours
        cmd = ["gcc", "-std=gnu99", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
               "-L", lib_dir, "-laudiosync", "-laudiosync_hip", "-lm", "-lpthread", "-Wl,-rpath," + lib_dir]
        p = subprocess.run(cmd, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-3000:]
        assert os.path.exists(exe)
        # the program binds the two entry points of the path by name
        nm = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
        wanted = "cross_correlation" if "cross_correlation" in rel else "pearson_coefficient"
        assert any(line.split()[-1].split("@")[0] == wanted for line in nm.splitlines() if line.strip()), nm
