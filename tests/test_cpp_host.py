"""The C++ host mirror (lightmotif_amd/host/lightmotif_hip.hpp) and its test program
tests/cpp/test_dna.cpp, the counterpart of lightmotif/tests/{dna,stripe,encode}.rs."""
import os
import subprocess
from pathlib import Path

import pytest

CPP = Path(__file__).resolve().parent / "cpp"


def build():
    subprocess.run(["make", "-C", str(CPP)], check=True, capture_output=True)
    return CPP / "test_dna"


def test_cpp_mirror_compiles_and_links_against_the_c_abi():
    exe = build()
    assert exe.exists()
    out = subprocess.run(["ldd", str(exe)], capture_output=True, text=True).stdout
    assert "liblightmotif_hip.so" in out and "not found" not in out.split("liblightmotif_hip.so")[1].split("\n")[0]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_cpp_mirror_without_device_raises_unsupported_backend():
    r = subprocess.run([str(build())], capture_output=True, text=True)
    assert r.returncode != 0 and "UnsupportedBackend" in r.stderr


@pytest.mark.gpu
def test_cpp_reference_style_tests_pass_on_gpu():
    r = subprocess.run([str(build())], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "all checks passed" in r.stdout


def test_dispatch_twin_compiles_and_links():
    build()
    exe = CPP / "test_dispatch"
    assert exe.exists()
    out = subprocess.run(["ldd", str(exe)], capture_output=True, text=True).stdout
    assert "liblightmotif_hip.so" in out and "liblm_avx2.so" in out and "not found" not in out


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_dispatch_twin_without_device_cannot_be_constructed():
    build()
    r = subprocess.run([str(CPP / "test_dispatch")], capture_output=True, text=True)
    assert r.returncode == 2 and "UnsupportedBackend" in r.stderr   # never a silent CPU-only `Hip` variant


@pytest.mark.gpu
def test_dispatch_hip_arm_table_routes_by_size_with_identical_bits():
    """every `match self.backend` site of dispatch.rs:58-207 + Scanner through HipDispatch<Cpu>;
    small inputs take the CPU tier, large ones the GPU, and the route never shows in the result."""
    build()
    r = subprocess.run([str(CPP / "test_dispatch")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "test_dispatch: all checks passed" in r.stdout
