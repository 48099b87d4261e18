"""Where the hot kernels issue their table look-ups, read from the disassembly of the built libaudiosync_hip.so (no GPU needed).

Round 5 (EXPERIMENTS.md 19, 20, 23; DESIGN.md section 2 "Order of a thread's loads"): `vmcnt` completes in order and the compiler sinks
a load into the only block that uses it, so a table look-up issued behind a barrier, or behind the HBM loads whose data it meets, puts an
L2 round trip between the arrival of that data and its first use.  Measured: rows -3.6 % (load steps) and -1.4 % (store phase), inverse
columns -3.6 %.  A refactoring can silently undo this; this test notices."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

from util import asx, graft

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def disassembly(so_path):
    """{demangled kernel name: [instruction mnemonics + operands]} of the gfx950 code objects in the library"""
    data = open(so_path, "rb").read()
    out = {}
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
                txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
                name = None
                for line in txt.splitlines():
                    m2 = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
                    if m2:
                        name = m2.group(1)
                        if name.startswith("_Z"):
                            out[name] = []
                        else:
                            name = None
                        continue
                    if name and line.startswith("\t"):
                        out[name].append(line.split("//")[0].strip())
    return out


PINNED_ROCM = "7.2"   # the compiler these orders were measured with; another one may order instructions differently without being wrong


def rocm_version():
    try:
        return open("/opt/rocm/.info/version").read().strip()
    except OSError:
        return ""


@pytest.fixture(scope="module")
def isa():
    if not os.path.exists(OBJDUMP):
        pytest.skip("no llvm-objdump in this image")
    # A SPEED property of the code the pinned compiler emits, not a correctness property: under another ROCm release the
    # checks below say "re-measure", they must not turn the suite red.
    if not rocm_version().startswith(PINNED_ROCM):
        pytest.skip("instruction orders are pinned to ROCm %s (found %r): re-measure profiles/r5_experiments/19-24 there" % (PINNED_ROCM, rocm_version()))
    asx()  # builds the library if it is not there
    d = disassembly(os.path.join(graft.PKG_DIR, "libaudiosync_hip.so"))
    names = subprocess.run(["c++filt"], input="\n".join(d), capture_output=True, text=True).stdout.split("\n")
    return {nm: d[k] for nm, k in zip(names, d)}


def one(isa, prefix):
    hit = [(n, v) for n, v in isa.items() if n.startswith(prefix)]
    assert len(hit) == 1, (prefix, [n for n, _ in hit][:4])
    return hit[0][1]


def positions(ins, pred):
    return [i for i, s in enumerate(ins) if pred(s)]


def test_two_half_row_kernel_issues_no_load_behind_a_barrier(isa):
    """k_rows_r, two-half form (N = 1 440 000, 960 000): every table look-up -- the load steps' w_M2 values, the stage seeds, the store
    phase's twiddles -- is issued in front of the first barrier; behind it the kernel only computes, exchanges through LDS and stores"""
    ins = one(isa, "void k_rows_r<Sched<1200, 12, 10, 10>, 128, true>")
    barriers = positions(ins, lambda s: s.startswith("s_barrier"))
    loads = positions(ins, lambda s: s.startswith("global_load"))
    assert len(barriers) >= 4 and loads, (len(barriers), len(loads))
    late = [i for i in loads if i > barriers[0]]
    assert not late, [ins[i] for i in late][:4]
    # ... and the rows (16-byte non-temporal loads) come behind the table values they meet: at least the three w_M2 loads of the load steps
    rows = positions(ins, lambda s: s.startswith("global_load_dwordx4") and s.endswith(" nt"))
    tables_x4 = positions(ins, lambda s: s.startswith("global_load_dwordx4") and not s.endswith(" nt"))
    assert rows and len(tables_x4) >= 3 and max(tables_x4) < min(rows), (tables_x4, rows[:2])


def test_one_wave_row_kernel_issues_every_look_up_in_front_of_its_rows(isa):
    """k_rows_r, 480-point rows (N = 144 000, 288 000, 480 000): twelve look-ups, then the rows, then waits that count down"""
    ins = one(isa, "void k_rows_r<Sched<480, 10, 8, 6>, 64, false>")
    rows = positions(ins, lambda s: s.startswith("global_load_dwordx4") and s.endswith(" nt"))
    k1_block = positions(ins, lambda s: s.startswith("global_load_dword ") )          # the k1 == 0 block's norm loads (4-byte), in front of everything
    tables = positions(ins, lambda s: s.startswith("global_load_dwordx2"))
    assert rows and tables
    behind = [i for i in tables if i > min(rows)]
    assert not behind, [ins[i] for i in behind][:4]
    assert all(i < min(rows) for i in k1_block)


def test_inverse_column_kernel_600_asks_for_its_twiddle_in_front_of_the_rows(isa):
    """k_inv_cols_r, 600-row tiles: between the first row load and the first barrier there is no 8-byte table load left"""
    ins = one(isa, "void k_inv_cols_r<Sched<600, 10, 10, 6>, 16, 512>")
    rows = positions(ins, lambda s: s.startswith("global_load_dwordx4") and s.endswith(" nt"))
    first_barrier = positions(ins, lambda s: s.startswith("s_barrier"))[0]
    tables = positions(ins, lambda s: s.startswith("global_load_dwordx2"))
    between = [i for i in tables if min(rows) < i < first_barrier]
    assert not between, [ins[i] for i in between][:4]
