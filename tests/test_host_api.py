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
