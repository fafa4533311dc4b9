"""The product's own multi-threaded HOST code under the sanitizers (aux row "sanitizers"): the host form of the
RDP import (pangea-plus_amd/csrc/rdp_host.hpp -- the name index built by compare-and-swap on all cores, the line / cursor /
field passes on up to 16 threads) is free of HIP, so it builds with plain g++ here: once under AddressSanitizer +
UndefinedBehaviorSanitizer, once under ThreadSanitizer, each comparing the parse (1, 3, 8, 16 threads) with a sequential
reading of the same odd-shaped files (tests/host/rdp_host_test.cpp).  CPU only; the same header is what libpangea_hip.so
compiles (tests/test_gpu_pipeline.py runs it on the GPU box with PGX_RDP_HOST=1)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")


@pytest.mark.parametrize("target", ["asan", "tsan"])
def test_host_rdp_import_under_sanitizers(target):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    p = subprocess.run(["make", "-C", HERE, target], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, out[-3000:]
    assert out.count(" ok") >= 9 and "DIFFERENT" not in out and "Sanitizer" not in out
