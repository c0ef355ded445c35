"""AddressSanitizer + UBSan over the host half of the library (CPU build; GPU sanitizers are
not available on this pool)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_host_setup_under_asan_ubsan(tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host.sh")], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "asan_host: ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
