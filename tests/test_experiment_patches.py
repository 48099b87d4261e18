"""tools/experiments/*.patch are measured-and-rejected experiments kept as patches (EXPERIMENTS.md).  Each must still apply
to the commit tools/experiments/MANIFEST.txt names for it ("nothing checks it"): `git apply --cached --check`
against an index read from that commit -- no checkout, no GPU.  A patch without a manifest line, or a manifest line
without a patch, fails too."""
import glob
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "tools", "experiments")


def manifest():
    out = {}
    for line in open(os.path.join(EXP, "MANIFEST.txt")):
        line = line.split("#")[0].split()
        if len(line) == 2:
            out[line[0]] = line[1]
    return out


def test_every_patch_has_a_commit_and_applies_to_it():
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout")
    m = manifest()
    patches = sorted(os.path.basename(p) for p in glob.glob(os.path.join(EXP, "*.patch")))
    assert patches == sorted(m), (patches, sorted(m))
    with tempfile.TemporaryDirectory() as tmp:
        for name, commit in m.items():
            env = dict(os.environ, GIT_INDEX_FILE=os.path.join(tmp, "idx"))
            r = subprocess.run(["git", "-C", ROOT, "read-tree", commit], env=env, capture_output=True, text=True)
            assert r.returncode == 0, (name, commit, r.stderr)
            r = subprocess.run(["git", "-C", ROOT, "apply", "--cached", "--check", os.path.join(EXP, name)], env=env,
                               capture_output=True, text=True)
            assert r.returncode == 0, (name, commit, r.stderr[-500:])
