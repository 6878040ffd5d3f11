"""CPU check of the third-generation NTT pass (olavm_amd/csrc/ntt3_core.cuh + tform.cuh): the per-thread phases the GPU kernels
run are compiled for the host and executed thread by thread on a range-checked limb type (tests/host_ntt3_check.cpp) -- T-form
arithmetic against the canonical field functions, every LDS access pattern free of bank conflicts, no 32-bit limb overflow, and
2^18-point transforms (forward / inverse, natural / bit-reversed order, coset LDE) equal to a plain radix-2 transform."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ntt3_phases_on_the_host(tmp_path):
    exe = str(tmp_path / "host_ntt3_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-o", exe, os.path.join(ROOT, "tests", "host_ntt3_check.cpp")])
    out = subprocess.run([exe, "18"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "all ok" in out.stdout and "0 bank conflicts" in out.stdout
