"""CPU builds of the native host code under AddressSanitizer + UBSan (GPU sanitizers are not available on the
pool; the reference has no sanitizer runs at all): the C restatement of the oracle and the HIP-free
stage planner, each with a self-test driver from tests/native/."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]


def _run(cmd, **kw):
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, **kw)
    assert out.returncode == 0, f"{' '.join(cmd)}\n{out.stdout}\n{out.stderr}"
    return out.stdout


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "ref_selftest")
    _run(["gcc", "-std=gnu11", "-fopenmp", "-fcx-limited-range", *SAN, os.path.join(ROOT, "tests", "native", "ref_selftest.c"),
          os.path.join(ROOT, "oracle", "aqc_ref.c"), "-lm", "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="3")
    assert "ok" in _run([exe], env=env)


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_planner_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "plan_selftest")
    _run(["g++", "-std=c++17", *SAN, os.path.join(ROOT, "tests", "native", "plan_selftest.cpp"),
          os.path.join(ROOT, "aqc_research_amd", "csrc", "aqc_plan.cpp"), "-o", exe])
    out = _run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert " 0 failures" in out
