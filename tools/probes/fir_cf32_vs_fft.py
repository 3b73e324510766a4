"""FIRFilter<complex<float>> without decimation: the time-domain kernel against overlap-save FFT convolution with the same taps
(1024 channels x 65536 samples, device pointers) — where the cross-over lies. usage: python tools/probes/fir_cf32_vs_fft.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import libsdr_amd as sa

FS, C, N = 2.4e6, 1024, 65536
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = sa.Context(0, stream=stream.cuda_stream)
    x = torch.randn((C, N, 2), dtype=torch.float32, device=dev) * 0.3
    y = torch.zeros((C, N, 2), dtype=torch.float32, device=dev)

    def timeit(call, reps=10):
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    for order in (8, 16, 32, 64, 127, 255, 1023, 4097):
        alpha = sa.design_fir_lowpass(order, 100e3, FS)
        os.environ["SDRHIP_FIR_TIME_DOMAIN"] = "1"
        fir = sa.FIR(ctx, sa.FIR_CF32, alpha, channels=C, max_in=N)
        del os.environ["SDRHIP_FIR_TIME_DOMAIN"]
        t_td = timeit(lambda: fir.process_dev(x.data_ptr(), N, N, y.data_ptr(), N), 3 if order > 1000 else 10)
        res = []
        for L in (1024, 2048, 4096, 8192, 16384):
            if L - order + 1 < L // 4:
                continue
            taps = np.stack([alpha[::-1], np.zeros_like(alpha)], 1).astype(np.float32)
            f = sa.FFTConv(ctx, sa.FFTCONV_OLS, L, taps, channels=C, max_in=N)
            res.append((L, timeit(lambda: f.process_dev(x.data_ptr(), N, N, y.data_ptr(), N))))
        print("order %5d: time domain %8.3f ms   FFT overlap-save: %s" % (order, t_td, "  ".join("L=%d %.3f" % r for r in res)))
