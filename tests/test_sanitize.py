"""Everything of the product that runs on host threads, under the sanitizers (CPU build only; the GPU boxes run none): `make -C voxel-rs_amd sanitize`
builds and runs the host mirror's known-answer tests under ASan + UBSan, and tests/cpp/sanitize_stress.cpp -- scene builder, chunk streamer workers, the
traversal image's Workers (whole-world build, 200 incremental updates at 1 / 4 / 16 threads), the device header walking what they built -- under
ASan + UBSan and, separately, TSan. The behaviour it protects: src/systems/worldsvo.rs:90-151 (workers only build chunks; the caller's thread owns the buffer)."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.timeout(900)
def test_host_threads_under_the_sanitizers():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-j3", "-C", str(ROOT / "voxel-rs_amd"), "sanitize"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=850)
    tail = "\n".join(ln for ln in r.stdout.splitlines() if "pragma" not in ln and not ln.lstrip().startswith(("|", "#pragma")))[-4000:]
    assert r.returncode == 0, tail
    # both sanitizer builds of the stress driver ran to their last line, and the known-answer tests passed under ASan + UBSan
    assert r.stdout.count("sanitize_stress: all checks passed") == 2 or "is up to date" in r.stdout, tail
    assert "FAILED" not in r.stdout and "runtime error" not in r.stdout and "ThreadSanitizer" not in r.stdout and "AddressSanitizer" not in r.stdout, tail
