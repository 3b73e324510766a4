"""Diagnostic: per-launch times of the headline kernel right after an idle period (what a 20-step timed region sees)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libsdr_amd as sa
FS = 2.4e6; C, N = 1024, 65536
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = sa.Context(0, stream=stream.cuda_stream)
    taps = sa.design_iqbb_taps(100e3, 50e3, FS, 127); lut = sa.design_freqshift_lut_i16()
    node = sa.IQBaseBandI16(ctx, taps, lut, sa.design_freqshift_inc(100e3, FS), False, 8, channels=C, max_in=N, epilogue=sa.EPI_FM)
    x = [torch.randint(-8000, 8000, (C, N, 2), dtype=torch.int16, device=dev) for _ in range(3)]
    out = torch.zeros((C, N // 8 + 2), dtype=torch.int16, device=dev)
    for idle in (0.0, 0.01, 0.5):
        for rep in range(2):
            torch.cuda.synchronize(); time.sleep(idle)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
            ev[0].record(stream)
            for i in range(40):
                node.process_dev(x[i % 3].data_ptr(), N, N, out.data_ptr(), out.shape[1])
                ev[i + 1].record(stream)
            torch.cuda.synchronize()
            t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(40)]
            print("idle %.2fs: first 8 launches (us): %s | mean 5..25: %.1f | mean 25..40: %.1f" % (idle, " ".join("%.0f" % v for v in t[:8]), sum(t[5:25]) / 20, sum(t[25:]) / 15))
