"""Host-buffer path of the drop-in nodes (`*_process`: H2D copy, kernel, D2H copy, sync), the PCIe-inclusive
figure DESIGN.md quotes beside bench.py's device-resident `value`. usage: python tools/host_path_bench.py"""
import time

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import libsdr_amd as sa

FS, N = 2.4e6, 65536
ctx = sa.Context(0)
taps, lut, inc = sa.design_iqbb_taps(100e3, 50e3, FS, 127), sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(100e3, FS)
rng = np.random.default_rng(1)
for C in (1, 64, 1024):
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, False, 8, channels=C, max_in=N, epilogue=sa.EPI_FM)
    x = rng.integers(-8000, 8000, (C, N, 2), dtype=np.int16)
    for _ in range(3):
        node.process(x)
    reps = 200 if C == 1 else 20 if C == 64 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        node.process(x)
    dt = (time.perf_counter() - t0) / reps
    print("IQBaseBand<int16>(127,/8)->FM host path: C=%4d  %8.3f ms per buffer  %8.1f MS/s  (real time needs %.1f ms of signal per buffer)"
          % (C, dt * 1e3, C * N / dt / 1e6, N / FS * 1e3))
# the same calls on PINNED caller buffers (sdrhip_host_register): the runtime then copies by plain DMA instead of
# staging pageable memory through its own bounce buffers
import ctypes as C_
L = sa.abi.lib()
for C in (1, 64, 1024):
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, False, 8, channels=C, max_in=N, epilogue=sa.EPI_FM)
    x = rng.integers(-8000, 8000, (C, N, 2), dtype=np.int16)
    no = node.out_count(N) + 1
    out = np.zeros((C, no), np.int16)
    for a in (x, out):
        sa.abi.check(L.sdrhip_host_register(a.ctypes.data_as(C_.c_void_p), a.nbytes))
    got = C_.c_size_t(0)
    call = lambda: sa.abi.check(L.sdrhip_iqbb_i16_process(node._h, x.ctypes.data_as(C_.c_void_p), N, N, out.ctypes.data_as(C_.c_void_p), no, C_.byref(got)))
    for _ in range(3):
        call()
    reps = 200 if C == 1 else 20 if C == 64 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    print("  ... pinned caller buffers:             C=%4d  %8.3f ms per buffer  %8.1f MS/s" % (C, dt * 1e3, C * N / dt / 1e6))
    for a in (x, out):
        L.sdrhip_host_unregister(a.ctypes.data_as(C_.c_void_p))
alpha = sa.design_fir_lowpass(127, 100e3, FS)
fb = sa.FloatBaseBand(ctx, 100e3, FS, alpha, 8, channels=1, max_in=N)
xf = (rng.standard_normal((1, N, 2)) * 0.3).astype(np.float32)
for _ in range(3):
    fb.process(xf)
t0 = time.perf_counter()
for _ in range(200):
    fb.process(xf)
dt = (time.perf_counter() - t0) / 200
print("float baseband (config 2) host path: C=   1  %8.3f ms per buffer  %8.1f MS/s" % (dt * 1e3, N / dt / 1e6))
