"""Register / scratch / LDS budgets of the gfx950 kernels, read from the code objects inside the built
libaudiosync_hip.so (no GPU needed).  The transform kernels live on exact occupancy budgets -- DESIGN.md
sections 3 and 5: column kernels two 512-thread blocks per CU (<= 128 VGPRs), k_rows four 256-thread blocks
per CU (<= 128 VGPRs, static LDS small enough that 4 x (38.4 KB + static) fits 160 KB) -- and a spill to scratch
costs tens of percent (the persistent k_rows experiment: 172 bytes of scratch, 1.22 -> 1.44 ms).  This guards them."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

from util import asx, graft

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels_of(so_path):
    data = open(so_path, "rb").read()
    found = {}
    with tempfile.TemporaryDirectory() as tmp:
        for n, m in enumerate(re.finditer(re.escape(MAGIC), data)):
            p = m.start()
            count = struct.unpack_from("<Q", data, p + 24)[0]
            q = p + 32
            for _ in range(count):
                off, size, tsz = struct.unpack_from("<QQQ", data, q)
                q += 24
                triple = data[q:q + tsz].decode()
                q += tsz
                if "gfx950" not in triple or size == 0:
                    continue
                path = os.path.join(tmp, "k%d.co" % n)
                with open(path, "wb") as f:
                    f.write(data[p + off:p + off + size])
                txt = subprocess.run([READELF, "--notes", path], capture_output=True, text=True, check=True).stdout
                for block in txt.split("  - .agpr_count:")[1:]:
                    name = re.search(r"\.name:\s+(\S+)", block).group(1)
                    found[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
                                   for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    return found


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(READELF):
        pytest.skip("no llvm-readelf in this image")
    asx()  # builds the library if it is not there
    return kernels_of(os.path.join(graft.PKG_DIR, "libaudiosync_hip.so"))


def demangled(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()


def test_every_kernel_is_built_for_gfx950_and_none_spills_in_the_production_variants(kernels):
    names = {demangled(k): v for k, v in kernels.items()}
    assert any(n.startswith("void k_rows<12, Sched<1200") for n in names), sorted(names)[:5]
    for n, r in names.items():
        if "Sched<" in n:  # compile-time schedules = the production lengths
            assert r["private_segment_fixed_size"] == 0, (n, r)
            assert r["vgpr_count"] <= 128, (n, r)


def test_k_rows_headline_kernel_keeps_four_blocks_per_cu(kernels):
    names = {demangled(k): v for k, v in kernels.items()}
    (n, r), = [(n, r) for n, r in names.items() if n.startswith("void k_rows<12, Sched<1200, 12, 10, 10>, 256>")]
    assert r["vgpr_count"] <= 128, r                       # four waves per SIMD (125 with the iterative ILP scheduler)
    lds_dynamic = 4 * 1200 * 8                             # four rows of M2 = 1200 complex values
    assert 4 * (lds_dynamic + r["group_segment_fixed_size"]) <= 160 * 1024, r


def test_column_kernels_headline_keep_two_blocks_per_cu(kernels):
    names = {demangled(k): v for k, v in kernels.items()}
    for prefix in ("void k_fwd_cols<12, Sched<1200, 12, 10, 10>, 8, 512>", "void k_inv_cols<12, Sched<1200, 12, 10, 10>, 8, 512>"):
        (n, r), = [(n, r) for n, r in names.items() if n.startswith(prefix)]
        assert r["vgpr_count"] <= 128 and r["private_segment_fixed_size"] == 0, (n, r)
        assert 2 * (1200 * 8 * 8 + r["group_segment_fixed_size"]) <= 160 * 1024, (n, r)


def test_real_column_kernels_keep_their_occupancy_budgets(kernels):
    """csrc/rlayout.hip (the production path): k_rows_r eight 128-thread blocks per CU in its one-row form (19.2 KB of
    dynamic LDS each) and four 256-thread blocks in its two-half form (38.4 KB), both at <= 128 VGPRs = four waves per
    SIMD; the column kernels two or three blocks per CU by their tile (76.8 KB for 600 x 16, 51.2 KB for 400 x 16,
    38.4 KB for 300 x 16); nothing spills."""
    names = {demangled(k): v for k, v in kernels.items()}
    rows = [(n, r) for n, r in names.items() if n.startswith("void k_rows_r<")]
    assert len(rows) >= 3, sorted(names)[:8]
    for n, r in rows:
        assert r["vgpr_count"] <= 128 and r["private_segment_fixed_size"] == 0, (n, r)
        two = n.rstrip(">( ").endswith("true") or ", true>" in n
        m2 = int(re.search(r"Sched<(\d+)", n).group(1)) * (2 if two else 1)
        blocks = 4 if two else 8
        if m2 <= 1200 or two:
            assert blocks * (m2 * 16 + r["group_segment_fixed_size"]) <= 160 * 1024, (n, r)
    cols = [(n, r) for n, r in names.items() if n.startswith("void k_fwd_cols_r<") or n.startswith("void k_inv_cols_r<")]
    assert len(cols) >= 6
    for n, r in cols:
        assert r["vgpr_count"] <= 128 and r["private_segment_fixed_size"] == 0, (n, r)
        m1 = int(re.search(r"Sched<(\d+)", n).group(1))
        assert 2 * (m1 * 16 * 8 + r["group_segment_fixed_size"]) <= 160 * 1024, (n, r)
