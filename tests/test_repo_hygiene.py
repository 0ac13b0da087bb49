"""CPU: what must never be in the history -- built binaries.  tests/cpp/test_ivfpq_codec (a 107 KB ELF) was committed twice
although .gitignore lists it (`git add -A` after a `git rm --cached` puts an ignored-but-present file back only when it is
still tracked; a tracked file is never ignored)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_built_binary_is_tracked():
    try:
        files = subprocess.run(["git", "ls-files", "-z"], cwd=ROOT, capture_output=True, check=True).stdout.split(b"\0")
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("not a git checkout")
    bad = []
    for f in files:
        if not f:
            continue
        path = os.path.join(ROOT, f.decode())
        try:
            with open(path, "rb") as fh:
                head = fh.read(4)
        except OSError:
            continue
        if head == b"\x7fELF" or f.endswith((b".so", b".o", b".a", b".hsaco")):
            bad.append(f.decode())
    assert not bad, "built binaries are tracked: %s" % bad
