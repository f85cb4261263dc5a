"""Runs the C++ host-mirror test program (tests/host/host_test.cpp): the reference-named classes
(Buffer, RayBuffer, BVH/SAHBVHBuilder, CudaBVH, CudaBVHTracer, HLBVHBuilder, RayGen, Renderer)
driven the way NTrace drives them."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host", "host_test")


def build():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "ntrace_amd", "csrc")])


def test_host_mirror_cpu():
    build()
    out = subprocess.run([EXE, "cpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host_test cpu: ok" in out.stdout


@pytest.mark.gpu
def test_host_mirror_gpu():
    assert os.path.exists(EXE), "tests/host/host_test was not built"
    out = subprocess.run([EXE, "gpu"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host_test gpu: ok" in out.stdout
