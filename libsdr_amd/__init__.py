"""libsdr_amd — MI355X-native libsdr hot path.

The product is `libsdrhip.so` (hand-written HIP for gfx950 behind the C ABI of include/sdrhip.h) and
the header-only C++ nodes of include/sdr/gpu/. This Python package is the ctypes plumbing tests and
bench.py use to reach the C ABI; it contains no compute and no CPU fallback.
"""
from . import abi  # noqa: F401
from .abi import (EPI_NONE, EPI_FM, EPI_AM, EPI_USB, FIR_CS16_EXACT, FIR_CF32, T_CS16, T_CF32,  # noqa: F401
                  FFTCONV_OLA, FFTCONV_OLS, SdrHipError)
from .nodes import *  # noqa: F401,F403

__version__ = "0.1.0"
