import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import libsdr_amd as sa
from oracle import pyoracle as orc
FS = 2.4e6
ctx = sa.Context(0)
taps = sa.design_iqbb_taps(100e3, 50e3, FS, 127); lut = sa.design_freqshift_lut_i16(); inc = sa.design_freqshift_inc(100e3, FS)
rng = np.random.default_rng(1)
for epi in (sa.EPI_NONE, sa.EPI_FM, sa.EPI_USB):
    C = 2
    x = rng.integers(-8000, 8000, size=(C, 3 * 8192, 2)).astype(np.int16)
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, False, 8, channels=C, max_in=8192, epilogue=epi)
    bb = [orc.IQBaseBandI16(taps, lut, inc, False, 8) for _ in range(C)]; fm = [orc.FMDemodI16() for _ in range(C)]
    for k in range(3):
        y = node.process(x[:, k * 8192:(k + 1) * 8192])
        for c in range(C):
            r = bb[c].process(x[c, k * 8192:(k + 1) * 8192])
            if epi == sa.EPI_FM: r = fm[c].process(r)
            elif epi == sa.EPI_USB: r = orc.usb_i16(r)
            bad = np.nonzero((y[c] != r).reshape(len(r), -1).any(axis=1))[0]
            print("epi", epi, "call", k, "ch", c, "n_out", len(r), "bad", len(bad), (bad[:6], bad[-6:]) if len(bad) else "")
