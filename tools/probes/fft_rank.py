"""Which transform size the overlap-save plans should pick for N taps, now that the 16384-point kernel is the pipelined one:
ms per call on device pointers for L = 2048 / 4096 / 8192 / 16384 at 1024, 256 and 64 channels x 65536 samples.
usage: python tools/probes/fft_rank.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import libsdr_amd as sa

N = 65536
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = sa.Context(0, stream=stream.cuda_stream)
    rng = np.random.default_rng(3)
    for C in (1024, 256, 64, 16):
        x = torch.randn((C, N, 2), dtype=torch.float32, device=dev) * 0.3
        y = torch.zeros((C, N, 2), dtype=torch.float32, device=dev)
        for taps_n in (256, 384, 512, 640, 768, 1024, 1536, 2048, 3072):
            h = (rng.standard_normal((taps_n, 2)) * 0.02).astype(np.float32)
            res = []
            for L in (2048, 4096, 8192, 16384):
                if L - taps_n + 1 < L // 4:
                    continue
                f = sa.FFTConv(ctx, sa.FFTCONV_OLS, L, h, channels=C, max_in=N)
                for _ in range(3):
                    f.process_dev(x.data_ptr(), N, N, y.data_ptr(), N)
                torch.cuda.synchronize()
                reps = 20
                t0 = time.perf_counter()
                for _ in range(reps):
                    f.process_dev(x.data_ptr(), N, N, y.data_ptr(), N)
                torch.cuda.synchronize()
                res.append((L, (time.perf_counter() - t0) / reps * 1e3))
            best = min(res, key=lambda r: r[1])[0]
            print("C %4d taps %4d: %s   best %d" % (C, taps_n, "  ".join("L=%d %.4f" % r for r in res), best))
