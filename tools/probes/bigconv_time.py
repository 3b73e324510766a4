"""What the any-size FFT filter costs where the transform does not fit one workgroup's LDS (BigConv: passes over device memory):
ms per call and GS/s on device pointers, beside the tuned 16384-point plan. usage: python tools/probes/bigconv_time.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import libsdr_amd as sa

FS = 2.4e6
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = sa.Context(0, stream=stream.cuda_stream)
    print("SDRHIP_FFTCONV_LITERAL =", os.environ.get("SDRHIP_FFTCONV_LITERAL", "(unset: awkward block sizes run as overlap-save on the best power of two)"))
    for Nb, C, dt in ((256, 256, np.float32), (512, 256, np.float32), (1024, 256, np.float32), (2048, 256, np.float32), (4096, 256, np.float32), (1000, 256, np.float32), (6000, 256, np.float32), (1000, 128, np.float64), (8192, 256, np.float32), (16384, 256, np.float32), (12000, 256, np.float32), (1009, 256, np.float32), (10007, 256, np.float32), (8192, 128, np.float64)):
        N = (65536 // Nb + 1) * Nb if 65536 % Nb else 65536
        K = sa.design_fftfilt_spectrum(sa.design_fftfilt_kernel(Nb, 50e3, 150e3, FS, dtype=dt))
        node = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * Nb, K, channels=C, max_in=N, dtype=dt)
        tdt = torch.float64 if dt == np.float64 else torch.float32
        x = torch.randn((C, N, 2), dtype=tdt, device=dev) * 0.3
        y = torch.zeros((C, N, 2), dtype=tdt, device=dev)
        for _ in range(3):
            node.process_dev(x.data_ptr(), N, N, y.data_ptr(), N)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            node.process_dev(x.data_ptr(), N, N, y.data_ptr(), N)
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / reps
        eb = 16 if dt == np.float64 else 8
        print("FilterNode<%s>(%5d): FFT %5d points, %3d channels x %6d samples: %7.3f ms per call = %6.1f GS/s, %4.1f %% of 8 TB/s at %d B per sample"
              % ("double" if dt == np.float64 else "float", Nb, 2 * Nb, C, N, dt_s * 1e3, C * N / dt_s / 1e9, C * N * 2 * eb / dt_s / 8e12 * 100, 2 * eb))
