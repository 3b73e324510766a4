import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

_DT = {"u8": np.uint8, "i8": np.int8, "i16": np.int16, "cs16": np.int16, "i32": np.int32, "f32": np.float32, "cf32": np.float32,
       "f64": np.float64}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
