"""AddressSanitizer + UBSan over the host-only sources (GPU sanitizers are not available on the pool): the .glaze reader and
writer with their own xz / PNG / JPEG codecs, the OBJ converter and the host SAH builder, on the reference's fixtures, on
truncated and bit-flipped files, and on degenerate builder input (tools/sanitize/)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_sources_under_asan_ubsan(tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path))
    out = subprocess.run([os.path.join(ROOT, "tools", "sanitize", "run.sh")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "host sanitize: ok" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
