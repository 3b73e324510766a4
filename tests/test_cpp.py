"""C++ side of the boundary: the API-compatible sdr:: core (CPU), and the sdr::gpu nodes wired into
graphs with that core (GPU) and with the UNMODIFIED reference runtime (GPU, drop-in proof)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
CXX = ["g++", "-O2", "-std=c++17", "-Wall", "-Werror=return-type", "-I" + os.path.join(ROOT, "include")]


def _build(src, out, extra=()):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, out)
    cmd = CXX + [os.path.join(ROOT, "tests", "cpp", src), "-o", exe] + list(extra) + ["-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "warning" not in r.stderr, r.stderr[-3000:]
    return exe


def _gpu_link_flags():
    from oracle import pyoracle
    pyoracle.build()
    return ["-I" + os.path.join(ROOT, "oracle"), "-L" + os.path.join(ROOT, "libsdr_amd"), "-lsdrhip",
            "-L" + os.path.join(ROOT, "oracle", "_build"), "-loracle",
            "-Wl,-rpath," + os.path.join(ROOT, "libsdr_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle", "_build")]


def test_core_cpu():
    exe = _build("test_core.cc", "test_core")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK (0 failures)" in r.stdout, r.stdout + r.stderr


def test_gpu_nodes_compile_and_link():
    """The node headers build against our core and link against the C-ABI library (no device needed)."""
    _build("test_gpu_nodes.cc", "test_gpu_nodes", _gpu_link_flags())


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference tree only exists in the build container")
def test_gpu_nodes_compile_against_reference_core():
    """Drop-in: include/sdr/gpu/nodes.hh compiles unchanged against the reference's own headers and links
    with the reference's own objects (oracle/Makefile, target _ref/dropin_ref)."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.exists(os.path.join(ROOT, "oracle", "_ref", "dropin_ref"))


@pytest.mark.gpu
def test_gpu_nodes_in_graphs():
    exe = _build("test_gpu_nodes.cc", "test_gpu_nodes", _gpu_link_flags())
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK (0 failures)" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_reference_runtime_drives_gpu_nodes():
    exe = os.path.join(ROOT, "oracle", "_ref", "dropin_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/dropin_ref was not built (needs /root/reference at build time)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK (0 failures)" in r.stdout, r.stdout + r.stderr
