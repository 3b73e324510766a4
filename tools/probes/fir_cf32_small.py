"""FIRLowPass<complex<float>> (no decimation) on SMALL plans: the time-domain kernel against the overlap-save FFT plan behind
sdrhip_fir (every plan had become a 2048/4096/16384-point block transform, whatever its size). Per call,
device pointers, back to back. usage: python tools/probes/fir_cf32_small.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import libsdr_amd as sa

FS = 2.4e6
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = sa.Context(0, stream=stream.cuda_stream)

    def timeit(call, reps=200):
        for _ in range(10):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    for C, N in ((1, 512), (1, 4096), (1, 65536), (4, 4096), (16, 4096), (64, 4096), (16, 65536), (64, 65536), (256, 16384)):
        x = torch.randn((C, N, 2), dtype=torch.float32, device=dev) * 0.3
        y = torch.zeros((C, N, 2), dtype=torch.float32, device=dev)
        row = []
        for order in (8, 32, 127, 255):
            alpha = sa.design_fir_lowpass(order, 100e3, FS)
            t = {}
            for mode in ("1", "0"):
                os.environ["SDRHIP_FIR_TIME_DOMAIN"] = mode
                os.environ["SDRHIP_FIR_FFT_ALWAYS"] = "1"
                fir = sa.FIR(ctx, sa.FIR_CF32, alpha, channels=C, max_in=N)
                t[mode] = timeit(lambda: fir.process_dev(x.data_ptr(), N, N, y.data_ptr(), N))
                fir.close()
            row.append("order %3d: td %7.1f us  fft %7.1f us" % (order, t["1"], t["0"]))
        print("C=%4d N=%6d  " % (C, N) + "   ".join(row), flush=True)
