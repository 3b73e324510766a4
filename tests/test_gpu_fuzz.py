"""Seeded random sweep of K1 plans against the oracle: order, decimation, shift (sign / zero), epilogue, input format,
channel count and ragged call lengths are drawn at random; whatever kernel the plan picks (VALU, MFMA 32x32x32 for
D = 8, MFMA for any D, one-plane cu8 instantiations) must reproduce the reference arithmetic bit for bit."""
import os

import numpy as np
import pytest

import libsdr_amd as sa

# SDRHIP_FUZZ_EXTRA=N adds N more seeds to every sweep (soak runs; the default set is what CI needs)
EXTRA = int(os.environ.get("SDRHIP_FUZZ_EXTRA", "0"))

pytestmark = pytest.mark.gpu
FS = 2.4e6


@pytest.fixture(scope="module")
def ctx():
    c = sa.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(40 + EXTRA))
def test_iqbb_random_plans(ctx, orc, seed):
    rng = np.random.default_rng(1000 + seed)
    order = int(rng.choice([1, 2, 7, 16, 21, 33, 64, 65, 100, 127, 129, 130, 200, 257, 300]))
    D = int(rng.choice([1, 2, 3, 5, 8, 8, 8, 12, 16, 83, 125, 300]))
    Fc = float(rng.choice([0.0, 100e3, -100e3, 37e3, -250e3, 1.1e6]))
    epi = int(rng.choice([sa.EPI_NONE, sa.EPI_FM, sa.EPI_AM, sa.EPI_USB]))
    cu8 = bool(rng.integers(0, 2))
    C = int(rng.choice([1, 2, 5]))
    taps = sa.design_iqbb_taps(float(rng.choice([0.0, 100e3, -60e3])), float(rng.choice([12.5e3, 50e3, 200e3])), FS, order)
    lut, inc = sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(Fc, FS)
    max_in = 9000
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, Fc < 0, D, channels=C, max_in=max_in, epilogue=epi)
    if cu8:
        node.set_input_format(sa.abi.IN_CU8)
    refs = [orc.IQBaseBandI16(taps, lut, inc, Fc < 0, D) for _ in range(C)]
    fms = [orc.FMDemodI16() for _ in range(C)]
    lens = [int(x) for x in rng.choice([0, 1, 2, 7, 129, 1000, 2048, 4097, 8191, 9000], size=5)]
    for n in lens:
        if cu8:
            x = rng.integers(0, 256, (C, n, 2), dtype=np.uint8)
        else:
            x = rng.integers(-32768, 32768, (C, n, 2), dtype=np.int16)
        y = node.process(x)
        for c in range(C):
            r = refs[c].process(orc.autocast_cu8_cs16(x[c]) if cu8 else x[c])
            if epi == sa.EPI_FM:
                r = fms[c].process(r)
            elif epi == sa.EPI_AM:
                r = orc.am_i16(r)
            elif epi == sa.EPI_USB:
                r = orc.usb_i16(r)
            assert y[c].shape == r.shape, (seed, order, D, Fc, epi, cu8, n, node.path)
            assert np.array_equal(y[c], r), (seed, order, D, Fc, epi, cu8, n, node.path)


@pytest.mark.parametrize("seed", range(16 + EXTRA))
def test_fir_cf32_random_plans(ctx, orc, seed, monkeypatch):
    """complex<float> FIR (+ folded SubSample, + AM / USB) at random orders, decimations and call lengths: <= 1e-5 relative.
    (Without decimation or demodulator the plan is an overlap-save FFT convolution behind the same handle; odd seeds keep the
    time-domain kernel there too: SDRHIP_FIR_TIME_DOMAIN=1.)"""
    rng = np.random.default_rng(2000 + seed)
    if seed & 1:
        monkeypatch.setenv("SDRHIP_FIR_TIME_DOMAIN", "1")
    else:
        monkeypatch.delenv("SDRHIP_FIR_TIME_DOMAIN", raising=False)
    order = int(rng.choice([1, 2, 16, 63, 127, 255, 1000]))
    D = int(rng.choice([1, 2, 3, 8, 8, 16, 50]))
    epi = int(rng.choice([sa.EPI_NONE, sa.EPI_AM, sa.EPI_USB]))
    C = int(rng.choice([1, 3]))
    alpha = sa.design_fir_lowpass(order, float(rng.choice([50e3, 100e3, 400e3])), FS)
    node = sa.FIR(ctx, sa.FIR_CF32, alpha, decim=D, channels=C, max_in=20000, epilogue=epi)
    firs, subs = [orc.FIR(alpha) for _ in range(C)], [orc.SubSample(D) for _ in range(C)]
    for n in [int(v) for v in rng.choice([0, 1, 5, 999, 4096, 12345, 20000], size=4)]:
        x = (rng.standard_normal((C, n, 2)) * 0.3).astype(np.float32)
        y = node.process(x)
        for c in range(C):
            ref = firs[c].process_cf32(x[c])
            if D > 1:
                ref = subs[c].process_cf32(ref)
            if epi == sa.EPI_AM:
                ref = orc.am_f32(ref)
            elif epi == sa.EPI_USB:
                ref = orc.usb_f32(ref)
            assert y[c].shape == ref.shape, (seed, order, D, epi, n)
            if ref.size:
                # relative to the stream's scale (inputs ~0.3): a one-sample output of USB's (re+im)/2 can cancel to ~0
                err = np.abs(y[c].astype(np.float64) - ref).max() / max(np.abs(ref).max(), 0.05)
                assert err <= 1e-5, (seed, order, D, epi, n, err)


@pytest.mark.parametrize("seed", range(8 + EXTRA // 4))
def test_fftconv_random_plans(ctx, seed):
    """Overlap-save at random FFT sizes / tap counts / call lengths against numpy's direct convolution (float64)."""
    rng = np.random.default_rng(3000 + seed)
    L = int(rng.choice([64, 256, 2048, 4096, 16384]))
    n_taps = int(rng.integers(1, L // 2 + 1))
    C = int(rng.choice([1, 2]))
    h = (rng.standard_normal((n_taps, 2)) / np.sqrt(n_taps)).astype(np.float32)
    node = sa.FFTConv(ctx, sa.FFTCONV_OLS, L, h, channels=C, max_in=30000)
    lens = [int(v) for v in rng.choice([1, 100, 4097, 16384, 30000], size=3)]
    x = (rng.standard_normal((C, sum(lens), 2)) * 0.3).astype(np.float32)
    ys, off = [], 0
    for n in lens:
        ys.append(node.process(x[:, off:off + n])); off += n
    y = np.concatenate(ys, axis=1)
    hc = h[:, 0].astype(np.float64) + 1j * h[:, 1]
    for c in range(C):
        xc = x[c, :, 0].astype(np.float64) + 1j * x[c, :, 1]
        ref = np.convolve(xc, hc)[:x.shape[1]]
        got = y[c, :, 0] + 1j * y[c, :, 1]
        assert np.abs(got - ref).max() / np.abs(ref).max() <= 1e-5, (seed, L, n_taps, lens)


@pytest.mark.parametrize("seed", range(12 + EXTRA // 4))
def test_fftconv_bank_random_plans(ctx, seed):
    """Filter banks (one forward transform per block, the spectrum held in registers across the bands for L = 1024 .. 8192,
    one launch per band elsewhere) at random sizes, band counts, channel counts and ragged call lengths against numpy's
    direct convolution per band; the overlap history is rolled by the channel's last block."""
    rng = np.random.default_rng(5000 + seed)
    L = int(rng.choice([32, 128, 512, 1024, 2048, 2048, 4096, 8192]))
    n_taps = int(rng.integers(1, L // 2 + 1))
    B = int(rng.choice([2, 3, 4]))
    C = int(rng.choice([1, 3]))
    hs = [(rng.standard_normal((n_taps, 2)) / np.sqrt(n_taps)).astype(np.float32) for _ in range(B)]
    node = sa.FFTConv(ctx, sa.FFTCONV_OLS, L, hs, channels=C, max_in=20000)
    lens = [int(v) for v in rng.choice([1, 100, L - n_taps + 1, 4097, 12345, 20000], size=4)]
    x = (rng.standard_normal((C, sum(lens), 2)) * 0.3).astype(np.float32)
    ys, off = [], 0
    for n in lens:
        ys.append(node.process(x[:, off:off + n])); off += n
    y = np.concatenate(ys, axis=2)   # [bands, channels, n, 2]
    for b in range(B):
        hc = hs[b][:, 0].astype(np.float64) + 1j * hs[b][:, 1]
        for c in range(C):
            xc = x[c, :, 0].astype(np.float64) + 1j * x[c, :, 1]
            ref = np.convolve(xc, hc)[:x.shape[1]]
            got = y[b, c, :, 0] + 1j * y[b, c, :, 1]
            assert np.abs(got - ref).max() / np.abs(ref).max() <= 1e-5, (seed, L, n_taps, B, C, lens, b, c)


@pytest.mark.parametrize("seed", range(16 + EXTRA // 2))
def test_fft_plan_random_sizes(ctx, seed):
    """FFTPlan at RANDOM sizes — whatever form the plan picks (in LDS, four-step, chirp in LDS, chirp over a four-step plan;
    fftany.hpp) — float and double, both directions, a batch of 2, against numpy's double FFT."""
    rng = np.random.default_rng(9100 + seed)
    n = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 20000)), int(rng.integers(20000, 70000))]))
    f64 = bool(rng.integers(0, 2))
    cdt, tol = (np.complex128, 2e-12) if f64 else (np.complex64, 1e-5)
    plan = sa.FFTPlan(ctx, n, cdt)
    x = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).astype(cdt)
    xd = x.astype(np.complex128)
    for sign, ref in ((-1, np.fft.fft(xd, axis=1)), (+1, np.fft.ifft(xd, axis=1) * n)):
        y = plan.exec_batch(x, sign)
        assert np.abs(y - ref).max() <= tol * max(np.abs(ref).max(), 1e-30), (seed, n, plan.form, sign)


@pytest.mark.parametrize("seed", range(10 + EXTRA // 4))
def test_fftconv_random_sizes_beyond_the_tuned_plans(ctx, seed):
    """Overlap-save at RANDOM FFT sizes (any factorisation, in or beyond one workgroup's LDS: GenConv / BigConv), random tap
    counts, 1-3 channels, 1-2 bands, ragged calls, float and double — against a direct convolution in double."""
    from scipy.signal import fftconvolve
    rng = np.random.default_rng(9300 + seed)
    L = int(rng.choice([int(rng.integers(8, 3000)), int(rng.integers(3000, 20000)), int(rng.integers(20000, 40000))]))
    n_taps = int(rng.integers(1, max(2, L // 2)))
    C, B = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    f64 = bool(rng.integers(0, 2))
    dt, tol = (np.float64, 1e-10) if f64 else (np.float32, 1e-5)
    hs = [(rng.standard_normal((n_taps, 2)) / np.sqrt(n_taps)).astype(dt) for _ in range(B)]
    max_in = 3 * L
    node = sa.FFTConv(ctx, sa.FFTCONV_OLS, L, hs if B > 1 else hs[0], channels=C, max_in=max_in, dtype=dt)
    lens = [int(v) for v in rng.choice([1, 7, L - n_taps + 1, L, 2 * L + 3, max_in], size=3)]
    x = (rng.standard_normal((C, sum(lens), 2)) * 0.3).astype(dt)
    ys, off = [], 0
    for n in lens:
        ys.append(node.process(x[:, off:off + n])); off += n
    y = np.concatenate(ys, axis=-2)
    if B == 1:
        y = y[None]
    for b in range(B):
        hc = hs[b][:, 0].astype(np.float64) + 1j * hs[b][:, 1]
        for c in range(C):
            xc = x[c, :, 0].astype(np.float64) + 1j * x[c, :, 1]
            ref = fftconvolve(xc, hc)[:x.shape[1]]
            got = y[b, c, :, 0].astype(np.float64) + 1j * y[b, c, :, 1]
            assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), (seed, L, n_taps, C, B, f64, lens, b, c)


@pytest.mark.parametrize("seed", range(10 + EXTRA))
def test_fir_cs16_exact_random_plans(ctx, orc, seed):
    """Exact per-tap-truncating complex<int16> FIR (+ FM / AM / USB) at random orders and ragged calls: bit-exact, including
    tap sets whose partial sums can leave int16 (the wrap variant)."""
    rng = np.random.default_rng(4000 + seed)
    order = int(rng.choice([1, 3, 16, 63, 127, 255]))
    epi = int(rng.choice([sa.EPI_NONE, sa.EPI_FM, sa.EPI_AM, sa.EPI_USB]))
    C = int(rng.choice([1, 3]))
    alpha = sa.design_fir_lowpass(order, float(rng.choice([50e3, 100e3, 600e3])), FS)
    if seed % 4 == 3:
        alpha = alpha * 2.5            # sum |alpha| > 1: partial sums may wrap
    node = sa.FIR(ctx, sa.FIR_CS16_EXACT, alpha, channels=C, max_in=6000, epilogue=epi)
    firs, fms = [orc.FIR(alpha) for _ in range(C)], [orc.FMDemodI16() for _ in range(C)]
    for n in [int(v) for v in rng.choice([0, 1, 2, 777, 2048, 6000], size=4)]:
        x = rng.integers(-32768, 32768, (C, n, 2), dtype=np.int16)
        y = node.process(x)
        for c in range(C):
            r = firs[c].process_cs16(x[c])
            if epi == sa.EPI_FM:
                r = fms[c].process(r)
            elif epi == sa.EPI_AM:
                r = orc.am_i16(r)
            elif epi == sa.EPI_USB:
                r = orc.usb_i16(r)
            assert y[c].shape == r.shape and np.array_equal(y[c], r), (seed, order, epi, n)


@pytest.mark.parametrize("seed", range(10 + EXTRA))
def test_real_baseband_random_plans(ctx, orc, seed):
    rng = np.random.default_rng(5000 + seed)
    order = int(rng.choice([1, 2, 21, 64, 127, 300]))
    D = int(rng.choice([1, 3, 8, 12, 125]))
    Fc = float(rng.choice([0.0, 100e3, -100e3, 333e3]))
    epi = int(rng.choice([sa.EPI_NONE, sa.EPI_FM, sa.EPI_AM, sa.EPI_USB]))
    C = int(rng.choice([1, 4]))
    taps = sa.design_bb_taps(float(rng.choice([50e3, 100e3, 230e3])), float(rng.choice([20e3, 80e3])), 1e6, order)
    lut, inc = sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(Fc, 1e6)
    node = sa.BaseBandI16(ctx, taps, lut, inc, Fc < 0, D, channels=C, max_in=5000, epilogue=epi)
    refs, fms = [orc.BaseBandI16(taps, lut, inc, Fc < 0, D) for _ in range(C)], [orc.FMDemodI16() for _ in range(C)]
    for n in [int(v) for v in rng.choice([0, 1, 9, 1000, 4097, 5000], size=4)]:
        x = rng.integers(-32768, 32768, (C, n), dtype=np.int16)
        y = node.process(x)
        for c in range(C):
            r = refs[c].process(x[c])
            if epi == sa.EPI_FM:
                r = fms[c].process(r)
            elif epi == sa.EPI_AM:
                r = orc.am_i16(r)
            elif epi == sa.EPI_USB:
                r = orc.usb_i16(r)
            assert y[c].shape == r.shape and np.array_equal(y[c], r), (seed, order, D, Fc, epi, n)


@pytest.mark.parametrize("seed", range(6 + EXTRA))
def test_subsample_and_demods_random(ctx, orc, seed):
    rng = np.random.default_rng(6000 + seed)
    C, n_sub = int(rng.choice([1, 3])), int(rng.choice([1, 2, 3, 8, 100]))
    sub = sa.SubSample(ctx, sa.T_CS16, n_sub, channels=C, max_in=5000)
    subf = sa.SubSample(ctx, sa.T_CF32, n_sub, channels=C, max_in=5000)
    fm = sa.Demod(ctx, sa.EPI_FM, sa.T_CS16, channels=C, max_in=5000, inplace_fm0=True)
    refs, refsf, fms = [orc.SubSample(n_sub) for _ in range(C)], [orc.SubSample(n_sub) for _ in range(C)], [orc.FMDemodI16() for _ in range(C)]
    for n in [int(v) for v in rng.choice([0, 1, 7, 1001, 4096, 5000], size=4)]:
        x = rng.integers(-32768, 32768, (C, n, 2), dtype=np.int16)
        xf = (rng.standard_normal((C, n, 2)) * 0.5).astype(np.float32)
        y, yf = sub.process(x), subf.process(xf)
        for c in range(C):
            assert np.array_equal(y[c], refs[c].process_cs16(x[c]))
            rf = refsf[c].process_cf32(xf[c])
            assert yf[c].shape == rf.shape and (rf.size == 0 or np.abs(yf[c] - rf).max() <= 1e-5 * max(np.abs(rf).max(), 1e-30))
        if n:
            z = fm.process(x)
            for c in range(C):
                assert np.array_equal(z[c], fms[c].process(x[c]))


def _hot_fuzz(ctx, orc, rng, order, cu8, decim=8):
    Fc = float(rng.choice([100e3, -100e3, 0.0, 333e3]))
    epi = int(rng.choice([sa.EPI_NONE, sa.EPI_FM, sa.EPI_FM, sa.EPI_AM, sa.EPI_USB]))
    width = float(rng.choice([12.5e3, 50e3, 200e3, 600e3]))
    C = int(rng.choice([1, 3, 7, 33]))
    taps = sa.design_iqbb_taps(float(rng.choice([0.0, 100e3, -60e3])), width, FS, order)
    lut, inc = sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(Fc, FS)
    lens = [int(rng.integers(4000, 70000)) for _ in range(3)] + [int(rng.choice([1, 500, 2047, 4031, 4032, 4033, 65536]))] + [int(rng.integers(4000, 30000))]
    rng.shuffle(lens)
    # (any-D forms, FM: the slices' first angle differences by the fix-up launch or inside the hot kernel — read at create)
    resident = int(rng.integers(0, 2))
    os.environ["SDRHIP_IQBB_FM_RESIDENT"] = str(resident)
    try:
        node = sa.IQBaseBandI16(ctx, taps, lut, inc, Fc < 0, decim, channels=C, max_in=max(lens), epilogue=epi)
    finally:
        del os.environ["SDRHIP_IQBB_FM_RESIDENT"]
    fix = ["iqbb_fm_fixup_kernel"] if epi == sa.EPI_FM and not resident else []
    if 257 <= decim <= 464 or decim > 512:   # (the large-decimation form: partial box sums + a finishing launch, whatever the demodulator)
        fix = [] if resident else ["iqbb_bigd_finish_kernel"]
    if cu8:
        node.set_input_format(sa.abi.IN_CU8)
    if decim == 8:
        assert node.kernel_names == ["iqbb_hot_kernel"]
    elif decim < 8:   # (17 K steps whose taps need every high plane, WITHOUT a shift: two sample arrays beside 34 KB of tap fragments fit no workgroup — the general kernel)
        assert node.path == 3 and (node.kernel_names == ["iqbb_hot_sd_kernel"] + fix or (order > 129 and inc == 0 and node.kernel_names == ["iqbb_i16_mfmag_kernel"]))
    else:
        assert node.path == 3 and node.kernel_names == ["iqbb_hot_anyd_kernel"] + fix
    refs = [orc.IQBaseBandI16(taps, lut, inc, Fc < 0, decim) for _ in range(C)]
    fms = [orc.FMDemodI16() for _ in range(C)]
    for n in lens:
        ev = int(rng.integers(0, 4))   # between buffers: retune the shift, swap the filter, both, or nothing (src/baseband.hh:82-112)
        if rng.integers(0, 4) == 0:    # ... or _reconfigure (:156-194): counters restart, the FIR ring stays (rotated); FMDemod goes on
            node.reset(keep_history=True, keep_fm=True)
            for r_ in refs:
                r_.reset()
        if ev & 1:
            Fc2 = float(rng.choice([100e3, -100e3, 0.0, 333e3, -41e3]))
            node.set_shift(sa.design_freqshift_inc(Fc2, FS), Fc2 < 0)
            for r_ in refs:
                r_.set_shift(sa.design_freqshift_inc(Fc2, FS), Fc2 < 0)
        if ev & 2:
            t2 = sa.design_iqbb_taps(float(rng.choice([0.0, 100e3, -60e3])), float(rng.choice([12.5e3, 50e3, 200e3, 600e3])), FS, order)
            node.set_taps(t2)
            for r_ in refs:
                r_.set_taps(t2)
        if cu8:
            x = rng.integers(0, 256, (C, n, 2), dtype=np.uint8)
        else:
            x = rng.integers(-32768, 32768, (C, n, 2), dtype=np.int16)
        y = node.process(x)
        for c in range(C):
            r = refs[c].process(orc.autocast_cu8_cs16(x[c]) if cu8 else x[c])
            if epi == sa.EPI_FM:
                r = fms[c].process(r)
            elif epi == sa.EPI_AM:
                r = orc.am_i16(r)
            elif epi == sa.EPI_USB:
                r = orc.usb_i16(r)
            assert y[c].shape == r.shape and np.array_equal(y[c], r), (order, cu8, Fc, epi, width, C, n, ev)


@pytest.mark.parametrize("seed", range(24 + EXTRA))
def test_one_launch_kernel_random_long_calls(ctx, orc, seed):
    """The 127-tap / decimation-8 plan's hot kernel (hot loop + cold phase in one launch): random long ragged calls (the
    last tile ends in every way, calls too short for a hot tile in between), channel counts, epilogues, shift signs and
    filter widths (narrow: few K steps carry the taps' high plane; wide: all nine), retuned between buffers (shift,
    filter), state carried from call to call."""
    rng = np.random.default_rng(7000 + seed)
    _hot_fuzz(ctx, orc, rng, int(rng.choice([127, 127, 127, 113, 128, 129])), False)


@pytest.mark.parametrize("cu8", [False, True])
@pytest.mark.parametrize("seed", range(20 + EXTRA))
def test_hot_kernel_every_length_class_random_long_calls(ctx, orc, seed, cu8):
    """The same sweep over every filter-length class of the hot kernel (2, 3, 5, 9, 17 K steps: the reference's own plans
    are order 16, examples/sdr_rec.cc:68, and 21, examples/sdr_fm.cc:40) and both input kinds (complex<int16>;
    complex<uint8> with AutoCast fused, src/autocast.hh:187-194)."""
    rng = np.random.default_rng(17000 + 2 * seed + int(cu8))
    _hot_fuzz(ctx, orc, rng, int(rng.choice([3, 16, 17, 21, 33, 34, 64, 65, 66, 100, 127, 130, 200, 255, 257, 258, 300, 400, 513])), cu8)


@pytest.mark.parametrize("cu8", [False, True])
@pytest.mark.parametrize("seed", range(16 + EXTRA))
def test_hot_kernel_any_decimation_random_long_calls(ctx, orc, seed, cu8):
    """The hot kernel's any-decimation form (9 <= D <= 512, here up to 129 taps, shifted or not): random plans incl. the reference receivers' (16 taps / 62, 21 taps / 125), ragged long and short calls,
    retuning (also to and from no shift at all), filter swaps and _reconfigure between buffers."""
    rng = np.random.default_rng(23000 + 2 * seed + int(cu8))
    order = int(rng.choice([3, 16, 17, 21, 33, 34, 64, 65, 100, 127, 129]))
    decim = int(rng.choice([9, 10, 12, 31, 50, 62, 100, 125, 180, 200, 256, 300, 512, 257, 464, 465, 513, 700, 1024]))
    _hot_fuzz(ctx, orc, rng, order, cu8, decim)


@pytest.mark.parametrize("cu8", [False, True])
@pytest.mark.parametrize("seed", range(8 + EXTRA))
def test_hot_kernel_any_decimation_long_filters_random_long_calls(ctx, orc, seed, cu8):
    """The any-decimation form's 17-K-step class (orders 130 ... 257: 8- or 16-wave workgroups by the taps' high-plane
    range — a filter swap between buffers moves a plan from one to the other)."""
    rng = np.random.default_rng(31000 + 2 * seed + int(cu8))
    order = int(rng.choice([130, 161, 200, 255, 257, 300, 513]))
    decim = int(rng.choice([9, 12, 31, 62, 125, 200, 300, 512, 640, 1000]))
    _hot_fuzz(ctx, orc, rng, order, cu8, decim)


@pytest.mark.parametrize("cu8", [False, True])
@pytest.mark.parametrize("seed", range(16 + EXTRA))
def test_hot_kernel_small_decimation_random_long_calls(ctx, orc, seed, cu8):
    """The hot kernel's small-decimation form (2 <= D <= 7, up to 129 taps, shifted or not; the reference accepts any
    sub_sample, src/baseband.hh:159-162): random plans, ragged long and short calls, retuning (also to and from no shift
    at all: the two forms differ in their LDS arrays), filter swaps and _reconfigure between buffers."""
    rng = np.random.default_rng(29000 + 2 * seed + int(cu8))
    order = int(rng.choice([3, 16, 17, 21, 33, 34, 64, 65, 100, 127, 129, 130, 200, 257]))
    decim = int(rng.choice([1, 2, 3, 4, 5, 6, 7]))
    _hot_fuzz(ctx, orc, rng, order, cu8, decim)


@pytest.mark.parametrize("hot", [True, False])
@pytest.mark.parametrize("seed", range(16 + EXTRA))
def test_bb_real_mfma_random_long_calls(ctx, orc, seed, hot, monkeypatch):
    """The real-input BaseBand<int16> at decimation 8 on the matrix cores (3, 5 or 9 K steps by order), long ragged calls:
    the hot kernel's real-input instantiation, and the general kernel alone."""
    monkeypatch.setenv("SDRHIP_IQBB_HOT", "1" if hot else "0")
    rng = np.random.default_rng(9000 + seed)
    order = int(rng.choice([5, 17, 21, 33, 64, 100, 127, 128, 160, 200, 255, 273]))
    Fc = float(rng.choice([100e3, -100e3, 0.0, 41e3]))
    epi = int(rng.choice([sa.EPI_NONE, sa.EPI_FM, sa.EPI_AM, sa.EPI_USB]))
    C = int(rng.choice([1, 2, 5]))
    Fs = 1e6
    taps = orc.bb_design(float(rng.choice([60e3, 120e3, 200e3])), float(rng.choice([20e3, 60e3, 150e3])), Fs, order)
    lut, inc = orc.freqshift_lut_i16(), orc.freqshift_inc(Fc, Fs)
    lens = [int(rng.integers(2000, 40000)) for _ in range(3)] + [int(rng.choice([0, 1, 7, 2015, 2016, 2017, 2048]))]
    rng.shuffle(lens)
    bb = sa.BaseBandI16(ctx, taps, lut, inc, Fc < 0, 8, channels=C, max_in=max(max(lens), 1), epilogue=epi)
    assert bb.kernel_names == (["iqbb_hot_kernel"] if hot else ["bb_real_mfma_kernel"])
    refs = [orc.BaseBandI16(taps, lut, inc, Fc < 0, 8) for _ in range(C)]
    fms = [orc.FMDemodI16() for _ in range(C)]
    for n in lens:
        x = rng.integers(-32768, 32768, (C, n), dtype=np.int16)
        y = bb.process(x)
        for c in range(C):
            r = refs[c].process(x[c])
            if epi == sa.EPI_FM:
                r = fms[c].process(r)
            elif epi == sa.EPI_AM:
                r = orc.am_i16(r)
            elif epi == sa.EPI_USB:
                r = orc.usb_i16(r)
            assert y[c].shape == r.shape and np.array_equal(y[c], r), (seed, order, Fc, epi, C, n, bb.kernel_names)


@pytest.mark.parametrize("tpw", [0, 3])
@pytest.mark.parametrize("seed", range(10 + EXTRA // 2))
def test_float_baseband_random_calls(ctx, orc, seed, tpw, monkeypatch):
    """The float baseband (shift fused into the register-tiled FIR's staging, decimation folded): random shift, decimation,
    channel count and ragged call lengths — odd lengths move the rows over every alignment of the 16-byte staging loads and
    the phase tables over every tile offset. <= 1e-5 against the float64-phasor oracle chain. tpw = 3: the decimation-8
    plans run the software-pipelined kernel (3 tiles per workgroup; small batches would take the plain one)."""
    if tpw:
        monkeypatch.setenv("SDRHIP_FIR_TPW", str(tpw))
    else:
        monkeypatch.delenv("SDRHIP_FIR_TPW", raising=False)
    rng = np.random.default_rng(11000 + seed)
    Fc = float(rng.choice([100e3, -100e3, 37e3, 1.1e6, 0.0]))
    D = int(rng.choice([8, 8, 8, 5, 1, 16]))
    C = int(rng.choice([1, 2, 5]))
    order = int(rng.choice([127, 63, 255]))
    alpha = sa.design_fir_lowpass(order, 100e3, FS)
    lens = [int(rng.integers(1, 40000)) for _ in range(4)] + [int(rng.choice([4096, 8192, 1, 2, 3]))]
    rng.shuffle(lens)
    node = sa.FloatBaseBand(ctx, Fc, FS, alpha, D, channels=C, max_in=max(lens))
    firs, subs = [orc.FIR(alpha) for _ in range(C)], [orc.SubSample(D) for _ in range(C)]
    n0 = 0
    for n in lens:
        x = (rng.standard_normal((C, n, 2)) * 0.3).astype(np.float32)
        y = node.process(x)
        for c in range(C):
            ref = firs[c].process_cf32(orc.freqshift_cf32(x[c], n0, Fc, FS))
            if D > 1:
                ref = subs[c].process_cf32(ref)
            assert y[c].shape == ref.shape, (seed, Fc, D, C, order, n)
            if ref.size:
                err = np.abs(y[c].astype(np.float64) - ref).max() / max(np.abs(ref).max(), 0.05)
                assert err <= 1e-5, (seed, Fc, D, C, order, n, err)
        n0 += n


@pytest.mark.parametrize("seed", range(24 + EXTRA))
def test_fmdeemph_random_rows(ctx, orc, seed, monkeypatch):
    """FMDeemph<int16_t>: alpha, channel count, call lengths, the number of lanes per channel (the plan's own choice or forced),
    the run-in (default / none / one / two groups of 64) and the rows' character — noise, full scale, constant, plateaus,
    a few spikes on silence, mixed per channel — drawn at random; the segmented kernel's guesses, checks and repairs must
    leave the sequential recursion's rows (src/demod.hh:342-351), whatever they were worth."""
    import ctypes
    rng = np.random.default_rng(7000 + seed)
    alpha = int(rng.choice([2, 3, 4, 5, 8, 10, 13, 16, 17, 40, 1000]))
    C = int(rng.choice([1, 2, 7, 33, 130]))
    P = rng.choice([None, None, 4, 8, 16, 32, 64])
    wc = rng.choice([None, None, 0, 1, 2])
    for k, v in (("SDRHIP_DEEMPH_SPEC", P), ("SDRHIP_DEEMPH_WC", wc)):
        if v is None:
            monkeypatch.delenv(k, raising=False)
        else:
            monkeypatch.setenv(k, str(int(v)))
    monkeypatch.delenv("SDRHIP_DEEMPH_TILED", raising=False)
    max_in = 12000
    node = sa.FMDeemphI16(ctx, alpha, channels=C, max_in=max_in)
    avgs = [np.zeros(1, np.int16) for _ in range(C)]

    def row(n):
        kind = rng.integers(0, 6)
        if kind == 0:
            return (rng.normal(0, float(rng.choice([3, 300, 5000])), n) + 2000 * np.sin(np.arange(n) * rng.uniform(0.001, 0.3))).clip(-32768, 32767)
        if kind == 1:
            return rng.integers(-32768, 32768, n)
        if kind == 2:
            return np.full(n, rng.integers(-32768, 32768))
        if kind == 3:
            w = int(rng.integers(20, 3000))
            return np.repeat(rng.integers(-30000, 30000, n // w + 1), w)[:n]
        if kind == 4:
            x = np.zeros(n)
            x[rng.integers(0, n, max(1, n // 500))] = rng.integers(-32768, 32768, max(1, n // 500))
            return x
        return np.concatenate([rng.integers(-100, 100, n // 2), np.full(n - n // 2, rng.integers(-5, 5))])

    for n in [int(v) for v in rng.choice([12000, 8192, 3276, 2049, 1000, 524, 65, 7, 1], size=5)]:
        x = np.stack([row(n) for _ in range(C)]).astype(np.int16)
        got = node.process(x)
        for c in range(C):
            o = np.zeros(n, np.int16)
            orc.lib().orc_fmdeemph_i16(orc._p(np.ascontiguousarray(x[c]), ctypes.c_int16), n, alpha, orc._p(avgs[c], ctypes.c_int16), orc._p(o, ctypes.c_int16))
            assert np.array_equal(got[c], o), (seed, alpha, C, P, wc, n, c, np.argwhere(got[c] != o)[:3].tolist())
