import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")

_DT = {"u8": np.uint8, "i8": np.int8, "i16": np.int16, "cs16": np.int16, "i32": np.int32, "f32": np.float32, "cf32": np.float32,
       "f64": np.float64}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "hostptr_only: keep this GPU test out of the red-zoned arena (it exercises the host-pointer path itself)")


class Golden:
    """Golden vectors cut from the compiled reference by oracle/ref_driver.cc (`make -C oracle golden`)."""

    def __init__(self):
        with open(os.path.join(GOLDEN, "manifest.json")) as f:
            self.manifest = json.load(f)

    def meta(self, name):
        return self.manifest[name]

    def load(self, name):
        m = self.manifest[name]
        a = np.fromfile(os.path.join(GOLDEN, m["file"]), dtype=_DT[m["dtype"]])
        assert a.size == m["count"], name
        if m["dtype"] in ("cs16", "cf32"):
            a = a.reshape(-1, 2)
        return a


@pytest.fixture(scope="session")
def golden():
    return Golden()


@pytest.fixture(scope="session")
def orc():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(autouse=True)
def redzone(request):
    """GPU parity and fuzz tests run every node call inside a red-zoned device arena (tests/redzone.py: guard
    bands before / between / after the rows of the input and output buffers, checked after every call). The parity
    module runs twice: through the arena (device-pointer entry points, strided rows) and through the host-pointer entry
    points (the library's own staging) — mark a test `hostptr_only` to keep it out of the arena."""
    mod = request.node.module.__name__ if request.node.module else ""
    gpu = request.node.get_closest_marker("gpu") is not None
    if not gpu or not any(k in mod for k in ("test_gpu_parity", "test_gpu_fuzz")):
        yield None
        return
    mode = getattr(request, "param", "redzone")
    from redzone import RedZone
    RedZone.install(mode == "redzone" and request.node.get_closest_marker("hostptr_only") is None)
    try:
        yield mode
    finally:
        RedZone.install(False)


def pytest_generate_tests(metafunc):
    # the parity module: every test in both modes (the fuzz module: arena only)
    if "redzone" in metafunc.fixturenames and metafunc.module.__name__.endswith("test_gpu_parity"):
        metafunc.parametrize("redzone", ["redzone", "hostptr"], indirect=True)
