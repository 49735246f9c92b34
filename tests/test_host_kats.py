"""Runs the C++ known-answer tests of the host mirror (tests/cpp/host_kats.cpp), one pytest case per reference #[test]."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "voxel-rs_amd" / "lib" / "host_kats"


def _ensure_built():
    if not BIN.exists():
        subprocess.run(["make", "-C", str(ROOT / "voxel-rs_amd"), "kats"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def _cases():
    _ensure_built()
    return subprocess.run([str(BIN), "--list"], check=True, stdout=subprocess.PIPE, text=True).stdout.split()


@pytest.mark.parametrize("case", _cases())
def test_host_kat(case):
    r = subprocess.run([str(BIN), case], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
