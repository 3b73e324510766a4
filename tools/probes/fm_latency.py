"""Per-buffer latency of ONE isolated call (host buffers in, host buffers out, synchronised — what a Queue worker sees) of
the reference's FM receiver plans on few channels: the neighbours' handshake inside the hot kernel (one launch) against
the second, tiny launch (SDRHIP_IQBB_FM_HANDSHAKE=0). usage: python tools/probes/fm_latency.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import libsdr_amd as sa

N = 65536
ctx = sa.Context(0)
rng = np.random.default_rng(1)
for name, order, D, fc, fs, width in (("sdr_fm (21 taps, /125, cu8)", 21, 125, 100e3, 1e6, 12.5e3), ("sdr_rec WFM (16 taps, /20, cu8)", 16, 20, 0.0, 1e6, 50e3)):
    taps, lut, inc = sa.design_iqbb_taps(fc, width, fs, order), sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(fc, fs)
    for C in (1, 4, 16):
        x = rng.integers(0, 256, (C, N, 2), dtype=np.uint8)
        res = {}
        for rnd in range(3):
            for hs in (1, 0):
                os.environ["SDRHIP_IQBB_FM_HANDSHAKE"] = str(hs)
                node = sa.IQBaseBandI16(ctx, taps, lut, inc, False, D, channels=C, max_in=N, epilogue=sa.EPI_FM)
                node.set_input_format(sa.abi.IN_CU8)
                for _ in range(20):
                    node.process(x)
                t0 = time.perf_counter()
                for _ in range(300):
                    node.process(x)
                res.setdefault(hs, []).append((time.perf_counter() - t0) / 300 * 1e6)
                names = node.kernel_names
                node.close()
        print("%-34s C=%2d  handshake (1 launch): %s us   fix-up launch (2 launches): %s us per buffer, host to host"
              % (name, C, " ".join("%.1f" % v for v in res[1]), " ".join("%.1f" % v for v in res[0])))
