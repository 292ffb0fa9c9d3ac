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


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_mutated_files_through_the_decoders_under_asan_ubsan(tmp_path):
    """Random scenes' files with bytes changed INSIDE a chunk and the chunk's XXH64 prefix made right again, so that the damage reaches the
    xz and PNG decoders (tools/sanitize/mutate_glaze.py), a tenth of them truncated as well: the reader may reject or default a chunk, the
    sanitizers must stay silent.  (4 500 files ran clean when this was written; the test takes 400.)"""
    env = dict(os.environ, TMPDIR=str(tmp_path))
    out = subprocess.run([os.path.join(ROOT, "tools", "sanitize", "fuzz_parser.sh"), "400", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "parsed" in out.stdout and "ERROR" not in out.stderr, (out.stdout[-2000:], out.stderr[-4000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_damaged_wavefront_files_through_the_converter_under_asan_ubsan(tmp_path):
    """tests/golden/cube.obj and its .mtl with lines dropped, doubled and shuffled, tokens replaced by junk (non-finite and huge numbers,
    indices that are zero, negative, out of range or malformed), missing material files, truncations (tools/sanitize/mutate_obj.py): the
    converter may refuse, the sanitizers must stay silent.  (3 000 files ran clean when this was written; the test takes 300.)"""
    env = dict(os.environ, TMPDIR=str(tmp_path))
    out = subprocess.run([os.path.join(ROOT, "tools", "sanitize", "fuzz_converter.sh"), "300", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "converted" in out.stdout and "ERROR" not in out.stderr, (out.stdout[-2000:], out.stderr[-4000:])
