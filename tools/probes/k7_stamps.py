"""Diagnostic (tools/build_variant_k7.sh stamps "-DK7_STAMPS"): where a turn of the pipelined 16384-point convolution kernel
(fftconv_fused_kernel PIPE) spends its shader clocks, wave 0 of every workgroup.
usage: python tools/probes/k7_stamps.py [variant=stamps] [mode=ols|ola]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SDRHIP_LIB"] = os.path.join(ROOT, "libsdr_amd", "libsdrhip_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "stamps"))
mode = sys.argv[2] if len(sys.argv) > 2 else "ols"
import torch
import numpy as np
import libsdr_amd as sa
C = 1024
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = sa.Context(0, stream=stream.cuda_stream)
    rng = np.random.default_rng(1)
    if mode == "ols":
        N = 6 * 12288
        h = (rng.standard_normal((4097, 2)) * 0.02).astype(np.float32)
        node = sa.FFTConv(ctx, sa.FFTCONV_OLS, 16384, h, channels=C, max_in=N)
    else:
        N = 65536
        K = sa.design_fftfilt_spectrum(sa.design_fftfilt_kernel(8192, 50e3, 150e3, 2.4e6))
        node = sa.FFTConv(ctx, sa.FFTCONV_OLA, 16384, K, channels=C, max_in=N)
    x = [torch.randn((C, N, 2), dtype=torch.float32, device=dev) for _ in range(2)]
    out = torch.zeros((C, N, 2), dtype=torch.float32, device=dev)
    for i in range(10):
        node.process_dev(x[i % 2].data_ptr(), N, N, out.data_ptr(), N)
    torch.cuda.synchronize()
    Kc = 500
    t0 = time.perf_counter()
    for i in range(Kc):
        node.process_dev(x[i % 2].data_ptr(), N, N, out.data_ptr(), N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / Kc
    L = sa.abi.lib()
    L.sdrhip_debug_k7_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    W = 256 * 16
    buf = (ctypes.c_ulonglong * W)()
    g = L.sdrhip_debug_k7_stamps(node._h, buf, W)
    assert g > 0, g
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 16)[:g].astype(np.float64)
turns = a[:, 8]
life = (a[:, 10] - a[:, 9]) / 100.0
names = ["fwd 1, 2, middle, inv 2", "prefetch issue", "inv 1 (+ its LDS drain)", "barrier", "last pass to dft16", "stores issued", "pass 0: inputs there, swapped", "pass 0 rest + barrier"]
tot = a[:, :8].sum()
print("mode %s: launch %.1f us (with the history roll); %d workgroups, turns %.1f each; wave 0 lives %.1f us (min %.1f max %.1f); clock %.2f GHz"
      % (mode, dt * 1e6, g, turns.mean(), life.mean(), life.min(), life.max(), a[:, :8].sum(axis=1).mean() / life.mean() / 1e3))
for q, n_ in enumerate(names):
    per = a[:, q].sum() / turns.sum()
    print("  %-32s %7.0f clocks per turn  %5.1f %%   (workgroup min %.0f max %.0f)" % (n_, per, 100 * a[:, q].sum() / tot, (a[:, q] / turns).min(), (a[:, q] / turns).max()))
print("  per turn %.0f clocks" % (tot / turns.sum()))
