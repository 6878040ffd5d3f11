"""The C++ host layer (include/ola_host.hpp) and its check program compile warning-free with a plain host compiler and
link against the C ABI only (no HIP headers, no torch)."""
import os
import subprocess

from tests import host_api_build


def test_host_layer_compiles_and_links_against_the_c_abi(tmp_path):
    from olavm_amd.backend import load_library
    load_library()                                     # the library exists (built by __graft_entry__.build)
    exe = host_api_build.build(tmp_path)
    assert os.path.exists(exe)
    # the program depends on libola_gpu.so and on nothing of ours besides it
    needed = subprocess.check_output(["readelf", "-d", exe], text=True)
    assert "libola_gpu.so" in needed and "liboracle" not in needed and "torch" not in needed
    # without its arguments it explains itself and exits before touching a device
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stdout


def test_rank_barrier_of_the_multi_device_context(tmp_path):
    """olavm_amd/csrc/peer_group.h on the host alone (tests/host_peer_group_check.cpp): 2 / 4 / 8 rank threads meet at the barrier
    thousands of times without anybody running ahead; a rank that gives up releases the others with an error instead of a
    deadlock, also when they arrive later; reset() makes the group usable again."""
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(str(tmp_path), "host_peer_group_check")
    subprocess.check_call(["hipcc", "-O1", "-std=c++17", "-pthread", "-o", exe, os.path.join(here, "host_peer_group_check.cpp")],
                          stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "all ok" in r.stdout, r.stdout + r.stderr


def test_header_is_valid_c_and_the_library_answers_its_revision(tmp_path):
    """include/ola_gpu.h compiled as C99 (-Wall -Wextra -pedantic -Werror) in tests/host_c_abi_check.c, linked against libola_gpu.so
    alone; run without arguments it compares the header's ABI revision and struct sizes with the library's and drives the host-only
    challenger entry points -- no device needed."""
    from olavm_amd.backend import lib_path, load_library
    load_library()
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.dirname(lib_path())
    exe = os.path.join(str(tmp_path), "host_c_abi_check")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", os.path.join(here, "host_c_abi_check.c"), "-o", exe,
                           "-L" + lib, "-lola_gpu", "-Wl,-rpath," + lib])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    import re
    rev = re.search(r"#define OLA_GPU_ABI_VERSION (\d+)", open(os.path.join(os.path.dirname(here), "include", "ola_gpu.h")).read()).group(1)
    assert r.returncode == 0 and "c abi ok: revision %s" % rev in r.stdout, r.stdout + r.stderr


def test_tform_arithmetic_on_the_host(tmp_path):
    """olavm_amd/csrc/tform.cuh -- the limb arithmetic of the transform passes -- is host + device code: tests/host_tform_check.cpp
    runs every primitive against canonical arithmetic on bound-sitting limb vectors (values AND magnitudes), checks that the
    blocks' powers of two are the reference's roots of unity, and the radix-16 / 8 / 4 / 2 blocks against a naive DFT in the
    reference's decimation-in-frequency order, forward and inverse."""
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(str(tmp_path), "host_tform_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-Wno-unknown-pragmas", "-I" + os.path.join(os.path.dirname(here), "olavm_amd", "csrc"),
                           os.path.join(here, "host_tform_check.cpp"), "-o", exe])
    r = subprocess.run([exe, "300000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "tform ok" in r.stdout, r.stdout + r.stderr
