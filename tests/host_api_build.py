"""Builds tests/host_api_check.cpp (the C++ host layer's check program) against the in-tree libola_gpu.so."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def build(out_dir):
    exe = os.path.join(str(out_dir), "host_api_check")
    lib = os.path.join(ROOT, "olavm_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(HERE, "host_api_check.cpp"), "-o", exe, "-L" + lib, "-lola_gpu", "-Wl,-rpath," + lib,
           "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe
