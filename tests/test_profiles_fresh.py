"""The committed evidence must describe the committed kernels: every profiles/r6_* file that carries a
stamp (tools/collect.sh writes `head` and `kernel_commit` into each JSON, and a `# ... kernel commit <hash>` line into
each CSV / text file) must have been taken at the last commit that touched old-audiosync_amd/csrc.  No GPU needed; skipped
outside a git checkout (the GPU box gets a snapshot without .git)."""
import glob
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = 6


def git(*args):
    return subprocess.run(["git", "-C", ROOT] + list(args), capture_output=True, text=True).stdout.strip()


def stamps():
    out = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r%d_*" % ROUND))):
        if os.path.isdir(path):
            continue
        name = os.path.basename(path)
        if name.endswith(".json"):
            try:
                d = json.load(open(path))
            except ValueError:
                d = json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
            out[name] = d.get("kernel_commit")
        else:
            m = re.search(r"kernel commit ([0-9a-f]{6,})", open(path).readline())
            out[name] = m.group(1) if m else None
    return out


def test_round_profiles_were_taken_at_the_kernels_of_head():
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout")
    found = stamps()
    if not found:
        pytest.skip("no profiles/r%d_* files yet (tools/collect.sh %d writes them)" % (ROUND, ROUND))
    kernels_at_head = git("log", "-1", "--format=%h", "--", "old-audiosync_amd/csrc")
    assert kernels_at_head
    unstamped = [n for n, c in found.items() if not c]
    assert not unstamped, "profiles without a commit stamp (regenerate with tools/collect.sh): %s" % unstamped
    stale = {n: c for n, c in found.items() if not (kernels_at_head.startswith(c) or c.startswith(kernels_at_head))}
    assert not stale, ("profiles taken at other kernels than HEAD's (%s); run tools/collect.sh %d: %s"
                       % (kernels_at_head, ROUND, stale))
