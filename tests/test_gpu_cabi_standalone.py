"""The C ABI is a plain shared library: a C++ host with no Python and no torch links libpgnn.so, drives the
structure build, the GIN aggregation and a Linear, and checks them against host loops (examples/c_abi_smoke.cpp)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_links_and_runs(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    libdir = os.path.join(ROOT, "pretrain_gnns_amd")
    exe = str(tmp_path / "c_abi_smoke")
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-w", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_smoke.cpp"), "-L" + libdir, "-lpgnn", "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.run(cmd, check=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "differing from the host scatter_add: 0" in out.stdout
