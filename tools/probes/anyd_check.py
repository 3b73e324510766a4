import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import libsdr_amd as sa
from oracle import pyoracle as orc
ctx = sa.Context(0)
FS = 1e6
rng = np.random.default_rng(5)
bad = 0
for (order, D, epi, cu8, Fc) in [(21, 125, sa.EPI_FM, True, 100e3), (16, 62, sa.EPI_NONE, False, 100e3), (21, 9, sa.EPI_AM, False, -60e3), (33, 180, sa.EPI_USB, False, 41e3), (16, 12, sa.EPI_FM, False, 100e3), (21, 125, sa.EPI_NONE, False, 100e3)]:
    C = 3
    taps, lut, inc = orc.iqbb_design(abs(Fc), 12.5e3, FS, order), orc.freqshift_lut_i16(), orc.freqshift_inc(Fc, FS)
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, Fc < 0, D, channels=C, max_in=70000, epilogue=epi)
    if cu8: node.set_input_format(sa.abi.IN_CU8)
    print(order, D, epi, cu8, node.kernel_names, node.path, flush=True)
    refs = [orc.IQBaseBandI16(taps, lut, inc, Fc < 0, D) for _ in range(C)]
    fms = [orc.FMDemodI16() for _ in range(C)]
    for n in (65536, 70000, 12345, 1, 40001, 65536):
        if cu8:
            u = rng.integers(0, 256, (C, n, 2), dtype=np.uint8); x = u
        else:
            x = rng.integers(-32768, 32768, (C, n, 2), dtype=np.int16)
        y = node.process(x)
        for c in range(C):
            xi = orc.autocast_cu8_cs16(x[c]) if cu8 else x[c]
            r = refs[c].process(xi)
            if epi == sa.EPI_FM: r = fms[c].process(r)
            elif epi == sa.EPI_AM: r = orc.am_i16(r)
            elif epi == sa.EPI_USB: r = orc.usb_i16(r)
            ok = y[c].shape == r.shape and np.array_equal(y[c], r)
            if not ok:
                bad += 1
                d = np.nonzero(y[c].reshape(len(r), -1) != r.reshape(len(r), -1))[0] if y[c].shape == r.shape else []
                print('  MISMATCH', order, D, epi, cu8, n, c, y[c].shape, r.shape, d[:10], len(d), flush=True)
print('bad', bad)
