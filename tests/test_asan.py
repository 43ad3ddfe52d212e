"""SURVEY.md section 5 (sanitizers): the HOST half of the C-ABI library -- argument validation, table builders, BatchNorm
folding / weight packing, workspace sizing, launch-strategy selection, error-string lifetime -- compiled with
``hipcc --cuda-host-only -fsanitize=address,undefined`` and run against a host-memory double of the HIP runtime
(tests/native/hip_host_double.cpp) by tests/native/abi_asan_driver.cpp.  CPU build only; no GPU sanitizer, no XNACK."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_make_asan_is_clean():
    if shutil.which("make") is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs make and hipcc")
    r = subprocess.run(["make", "-j4", "asan"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "abi_asan_driver: ok" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail and "LeakSanitizer" not in tail, tail
