"""The host graph preparation (graphflow_amd/csrc/smp_prep.cpp: threads, reused vectors, index tables) under the
sanitizers the reference never ran: AddressSanitizer + UBSan, and ThreadSanitizer."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "cpp", "prep_sanitize.cpp"), os.path.join(ROOT, "graphflow_amd", "csrc", "smp_prep.cpp")]


@pytest.mark.parametrize("flags", ["-fsanitize=address,undefined", "-fsanitize=thread"])
def test_host_preparation_is_clean_under_sanitizers(tmp_path, flags):
    exe = str(tmp_path / "prep_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", flags, "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "graphflow_amd", "csrc"), "-o", exe] + SRC
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitizer" in (r.stderr + r.stdout).lower() and "cannot find" in (r.stderr + r.stdout).lower():
        pytest.skip("sanitizer runtime not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="halt_on_error=1")
    env.pop("GF_PREP_THREADS", None)
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-3000:]
    assert "PASSED" in run.stdout
