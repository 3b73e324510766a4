"""GPU parity tests: the HIP path, called through the C ABI, against (a) golden vectors cut from the
compiled reference and (b) the CPU oracle on seeded inputs. Bit-exact for every int16 path;
max|y - y_ref| / max|y_ref| <= 1e-5 for float / FFT paths (BASELINE.json north_star).
Run with `pytest -m gpu` on an MI355X."""
import os

import ctypes
import numpy as np
import pytest

import libsdr_amd as sa

try:   # torch brings its own HIP runtime: it only finds the GPU when it initialises before libsdrhip.so does
    import torch
    if torch.cuda.device_count() > 0:
        torch.cuda.init()
except Exception:   # pragma: no cover
    torch = None

pytestmark = pytest.mark.gpu

FS = 2.4e6
RTOL = 1e-5


def split(x, lens):
    out, off = [], 0
    for n in lens:
        out.append(x[off:off + n])
        off += n
    return out


def rel_err(y, ref):
    y, ref = np.asarray(y, np.float64), np.asarray(ref, np.float64)
    return np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30)


@pytest.fixture(scope="module")
def ctx():
    c = sa.Context(0)
    yield c
    c.close()


@pytest.fixture(params=["auto", "valu", "general"])
def k1path(request, monkeypatch):
    """K1's bit-exact formulations: the int8-MFMA block-Toeplitz GEMM on 32x32x32 tiles — decimation 8 (path 1), any other
    decimation (path 3), real input (path 4), each with a hot kernel (one launch: persistent grid over the call's interior
    wave slices + cold phase; long calls) and a general kernel (short calls; "general": SDRHIP_IQBB_HOT=0 runs it for every
    call) — and the VALU dot2 kernel (path 0: the int8 chain, taps that do not fit two byte planes, longer filters;
    "valu": SDRHIP_IQBB_PATH=valu runs it for every plan). Every K1 test runs under each selection. (Rounds 1-3 also
    carried a 16x16x64 MFMA shape and a register-staged general kernel; neither served a plan by default: removed.)"""
    monkeypatch.delenv("SDRHIP_IQBB_HOT", raising=False)
    monkeypatch.delenv("SDRHIP_IQBB_PATH", raising=False)
    if request.param == "general":
        monkeypatch.setenv("SDRHIP_IQBB_HOT", "0")
    elif request.param == "valu":
        monkeypatch.setenv("SDRHIP_IQBB_PATH", "valu")
    return request.param


def iqbb_from_case(ctx, golden, case, suffix, epilogue, channels=1, max_in=4096):
    m = golden.meta(case + suffix)
    tcase = case if (case + "_taps") in golden.manifest else "g3_iqbb127d8"
    node = sa.IQBaseBandI16(ctx, golden.load(tcase + "_taps"), golden.load(tcase + "_lut"), m["lut_inc"], m["negative"],
                            m["decim"], channels=channels, max_in=max_in, epilogue=epilogue)
    return m, node


# ---- K1 against golden vectors ----------------------------------------------------------------------

@pytest.mark.parametrize("case,inp", [
    ("g3_iqbb127d8", "g1_iq_cs16"), ("g8_neg_o16_d1", "g1_iq_cs16_tone_m100k"), ("g8_o21_d3", "g1_iq_cs16"),
    ("g8_o33_d5", "g1_iq_cs16"), ("g8_o16_d4_even", "g1_iq_cs16"), ("g8_o255_d8", "g1_iq_cs16"),
    ("g8_noshift_o21_d8", "g1_iq_cs16"), ("g8_ofs_d300", "g1_iq_cs16"), ("g8_irregular", "g1_iq_cs16")])
def test_iqbb_golden(ctx, golden, case, inp, k1path):
    m, node = iqbb_from_case(ctx, golden, case, "_out", sa.EPI_NONE)
    outs = [node.process(c)[0] for c in split(golden.load(inp), m["in_lens"])]
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load(case + "_out"))


@pytest.mark.parametrize("case,inp,demod", [
    ("g4_iqbb127d8", "g1_iq_cs16", "fm"), ("g4_iqbb127d8", "g1_iq_cs16", "am"), ("g4_iqbb127d8", "g1_iq_cs16", "usb"),
    ("g8_o33_d5", "g1_iq_cs16", "fm"), ("g8_irregular", "g1_iq_cs16", "fm"), ("g8_irregular", "g1_iq_cs16", "usb"),
    ("g8_loud_iqbb127d8", "g8_iq_cs16_loud", "fm"), ("g8_loud_iqbb127d8", "g8_iq_cs16_loud", "am")])
def test_iqbb_demod_golden(ctx, golden, case, inp, demod, k1path):
    epi = {"fm": sa.EPI_FM, "am": sa.EPI_AM, "usb": sa.EPI_USB}[demod]
    m, node = iqbb_from_case(ctx, golden, case, "_" + demod, epi)
    outs = [node.process(c)[0] for c in split(golden.load(inp), m["in_lens"])]
    if demod == "fm":   # FMDemod does not send on an empty buffer
        outs = [o for o in outs if len(o)]
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load(case + "_" + demod))


# ---- K1 batched against the oracle -------------------------------------------------------------------

def synth_channels(orc, C, N, seed=0x5D2):
    """SURVEY §8d config-3 style channels: two tones + uniform integer noise, per channel."""
    x = np.zeros((C, N, 2), np.int16)
    for c in range(C):
        g = orc.IQSigGen(FS, [(50e3 + 97 * c, 7000, 0.1 * c), (-200e3 - 53 * c, 5000, 0.1 * c)])
        s = g.next_cs16(N).astype(np.int32)
        rng = np.random.default_rng(seed + c)
        s += rng.integers(-64, 65, size=s.shape)
        x[c] = s.astype(np.int16)
    return x


@pytest.mark.parametrize("epi", [sa.EPI_NONE, sa.EPI_FM, sa.EPI_AM, sa.EPI_USB])
@pytest.mark.parametrize("order,decim,Fc", [(127, 8, 100e3), (21, 8, -100e3), (33, 5, 100e3), (16, 1, 50e3),
                                             (129, 8, 0.0), (64, 8, 30e3), (9, 8, -250e3), (1, 8, 100e3), (150, 8, 70e3), (257, 8, 100e3), (200, 8, -40e3), (300, 8, 100e3), (513, 8, -60e3), (258, 8, 0.0), (129, 5, 100e3), (200, 3, -60e3), (64, 12, 30e3)])
def test_iqbb_batched_vs_oracle(ctx, orc, epi, order, decim, Fc, k1path):
    C, chunks = 5, [8192, 3000, 1, 7, 5000, 8192]
    taps = sa.design_iqbb_taps(Fc, 50e3, FS, order)
    lut = sa.design_freqshift_lut_i16()
    inc = sa.design_freqshift_inc(Fc, FS)
    x = synth_channels(orc, C, sum(chunks))
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, Fc < 0, decim, channels=C, max_in=8192, epilogue=epi)
    refs = [(orc.IQBaseBandI16(taps, lut, inc, Fc < 0, decim), orc.FMDemodI16()) for _ in range(C)]
    off = 0
    for n in chunks:
        y = node.process(x[:, off:off + n])
        for c in range(C):
            bb, fm = refs[c]
            r = bb.process(x[c, off:off + n])
            if epi == sa.EPI_FM:
                r = fm.process(r) if len(r) else np.zeros(0, np.int16)
            elif epi == sa.EPI_AM:
                r = orc.am_i16(r)
            elif epi == sa.EPI_USB:
                r = orc.usb_i16(r)
            assert y[c].shape == r.shape, (c, n, y[c].shape, r.shape)
            assert np.array_equal(y[c], r), (c, n, off)
        off += n


ANYD_CASES = [(21, 125, 100e3, True), (16, 62, 100e3, False), (21, 9, -60e3, False), (33, 180, 41e3, False), (16, 12, 100e3, False),
              (17, 31, -250e3, True), (3, 100, 100e3, False), (33, 10, 70e3, True),
              (64, 20, 100e3, False), (65, 125, -100e3, True), (127, 125, 100e3, False), (129, 9, 30e3, True), (100, 50, 100e3, False),
              (21, 200, 100e3, True), (16, 256, -100e3, False), (64, 181, 41e3, False), (21, 300, 100e3, True), (16, 512, 100e3, False),
              (33, 257, -60e3, False),
              (16, 20, 0.0, True), (16, 83, 0.0, True), (21, 125, 0.0, False), (64, 100, 0.0, False), (127, 300, 0.0, True), (16, 9, 0.0, False),
              (21, 45, 0.0, True),
              # orders 130 ... 257 (17 K steps): the any-D form in the 8- and 16-wave workgroups of the /8 kernel's long-filter class
              (255, 125, 100e3, True), (200, 20, -60e3, False), (257, 9, 0.0, False), (130, 62, 100e3, False), (161, 300, 0.0, True),
              # orders 258 ... 513 (33 K steps): one 8-wave workgroup per CU, 1024-sample windows; short calls run the VALU kernel
              (300, 125, 100e3, True), (513, 20, -60e3, False), (400, 9, 0.0, False), (258, 300, 100e3, True), (350, 62, 41e3, False), (300, 1000, 100e3, True),
              # decimations above 512: a group spans slices — the hot kernel leaves partial box sums, iqbb_bigd_finish_kernel finishes the groups
              # (up to 2048, with FM 1024, a plan has a general kernel for its short calls; beyond — up to 32768 — the large-
              # decimation form serves every call, and a plan without the hot kernel does not exist)
              (21, 2048, 100e3, True), (16, 5000, 0.0, False), (64, 30000, -60e3, False), (200, 1025, 41e3, True),
              (21, 600, 100e3, True), (16, 1000, 0.0, False), (64, 513, -60e3, False), (127, 900, 100e3, True), (21, 1023, 0.0, False),
              (200, 700, 100e3, False), (33, 777, 41e3, True), (21, 1024, 100e3, False),
              # decimations 2 ... 7: the hot kernel's small-decimation form (a slice holds 73 ... 256 groups: lane l finishes
              # the groups l, l + 64, ... of every slice)
              (21, 1, 100e3, False), (16, 1, -100e3, True), (127, 1, 0.0, False), (21, 2, 100e3, False), (16, 3, -100e3, True), (33, 4, 70e3, False), (64, 5, 100e3, True), (127, 6, -60e3, False), (21, 7, 100e3, True),
              (129, 2, 30e3, True), (16, 4, 0.0, False), (21, 7, 0.0, True), (65, 3, 0.0, False), (127, 5, 0.0, True),
              # ... in 8- and 16-wave workgroups: 17 K steps (orders 130 ... 257), and 9 K steps without a shift (127 / 5 above)
              (255, 6, 100e3, True), (200, 7, 0.0, False), (130, 7, -60e3, False), (257, 6, 0.0, True), (100, 6, 0.0, False), (129, 2, 0.0, True),
              # (17 K steps at decimations 2 ... 5: the general kernel of these plans needs 65 ... 79 KB of LDS)
              (255, 4, 100e3, True), (130, 2, -60e3, False), (257, 5, 0.0, True), (200, 3, 100e3, False)]   # (no shift: examples/sdr_rec.cc:42-58 tunes every mode to the centre; 21 taps / 45: examples/sdr_pocsag.cc:117 and sdr_ax25.cc:117 behind a 1 MS/s RTL source)


@pytest.mark.parametrize("epi,hot", [(e, h) for e in (sa.EPI_NONE, sa.EPI_FM, sa.EPI_AM, sa.EPI_USB) for h in (True, False)] + [(sa.EPI_FM, "resident")])
@pytest.mark.parametrize("order,decim,Fc,cu8", ANYD_CASES)
def test_iqbb_any_decimation_long_calls_vs_oracle(ctx, orc, order, decim, Fc, cu8, epi, hot, monkeypatch):
    """The reference's own receivers decimate by 62 (examples/sdr_rec.cc:68, 16 taps) and 125 (examples/sdr_fm.cc:40, 21
    taps on complex<uint8> input): plans of up to 257 taps, shifted or not, with 9 <= D <= 512 run the hot kernel's any-D form
    on the interior tiles of a long call and the general any-D kernel on the border tiles (two launches, seam tiles
    written by both). Ragged long and short calls, state carried across them, against the oracle; `hot` = False: the
    general kernel alone (SDRHIP_IQBB_HOT=0); "resident" (FM only): whole channels as the hot kernel's units, which then
    completes the slices' first angle differences itself instead of leaving them to iqbb_fm_fixup_kernel (what 1024 or 8192
    channels get by themselves: SDRHIP_IQBB_FM_RESIDENT forces it on these 3)."""
    monkeypatch.setenv("SDRHIP_IQBB_HOT", "1" if hot else "0")
    monkeypatch.setenv("SDRHIP_IQBB_FM_RESIDENT", "1" if hot == "resident" else "0")
    monkeypatch.delenv("SDRHIP_IQBB_FM_HANDSHAKE", raising=False)
    monkeypatch.delenv("SDRHIP_IQBB_PATH", raising=False)
    monkeypatch.delenv("SDRHIP_IQBB_BIGD_MIN", raising=False)
    # decimations 257 ... 512 run either of two hot forms (by default the faster one: the large-decimation form below 465):
    # both are tested — SDRHIP_IQBB_BIGD_MIN=n sends exactly the decimations >= n to the large-decimation form
    for bigd_min in ([513, 257] if 257 <= decim <= 512 and hot else [None]):
        if bigd_min is not None:
            monkeypatch.setenv("SDRHIP_IQBB_BIGD_MIN", str(bigd_min))
        bigd = decim > 512 or bigd_min == 257
        # (whole channels as units: the large-decimation form then finishes its groups inside the hot kernel, whatever the
        # demodulator — both ways for every epilogue)
        for resident in ([False, True] if bigd and hot is True else [hot == "resident"]):
            monkeypatch.setenv("SDRHIP_IQBB_FM_RESIDENT", "1" if resident else "0")
            _any_decimation_case(ctx, orc, order, decim, Fc, cu8, epi, hot, bigd, resident)


def _any_decimation_case(ctx, orc, order, decim, Fc, cu8, epi, hot, bigd, resident, handshake=False):
    FSr, C = 1e6, 3
    rng = np.random.default_rng(order * 1000 + decim)
    taps, lut, inc = orc.iqbb_design(abs(Fc), 12.5e3, FSr, order), orc.freqshift_lut_i16(), orc.freqshift_inc(Fc, FSr)
    assert (inc == 0) == (Fc == 0.0)
    if not hot and decim > (1024 if epi == sa.EPI_FM else 2048):   # (no general kernel for such a plan: the hot kernel or nothing)
        with pytest.raises(sa.abi.SdrHipError) as e:
            sa.IQBaseBandI16(ctx, taps, lut, inc, Fc < 0, decim, channels=C, max_in=70000, epilogue=epi)
        assert e.value.code == sa.abi.E_UNSUPPORTED and "too large" in str(e.value)
        return
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, Fc < 0, decim, channels=C, max_in=70000, epilogue=epi)
    if cu8:
        node.set_input_format(sa.abi.IN_CU8)
    hot_name = "iqbb_hot_sd_kernel" if decim < 8 else "iqbb_hot_anyd_kernel"
    # (the small-decimation form's sample arrays must fit a workgroup's LDS beside the tap fragments: where they do not
    # in the class's own workgroup — 9 or 17 K steps WITHOUT a shift: two arrays of 18-bit values — the plan runs in one of
    # twice the waves sharing the fragments; every plan of up to 257 taps has a hot form)
    launches = [hot_name] + (["iqbb_fm_fixup_kernel"] if epi == sa.EPI_FM and not resident and not handshake else [])
    if bigd:
        launches = [hot_name] + ([] if resident else ["iqbb_bigd_finish_kernel"])
    if order > 257 and not hot:   # (the 33-step class exists as hot forms only: SDRHIP_IQBB_HOT=0 leaves such a plan the VALU kernel)
        assert node.path == 0 and node.kernel_names == ["iqbb_i16_kernel"]
    else:
        assert node.path == 3
        assert node.kernel_names == (launches if hot else ["iqbb_i16_mfmag_kernel"])
    refs = [orc.IQBaseBandI16(taps, lut, inc, Fc < 0, decim) for _ in range(C)]
    fms = [orc.FMDemodI16() for _ in range(C)]
    for n in (65536, 70000, 12345, 1, 40001, 2 * decim + 1, 65536):
        x = rng.integers(0, 256, (C, n, 2), dtype=np.uint8) if cu8 else rng.integers(-32768, 32768, (C, n, 2), dtype=np.int16)
        y = node.process(x)
        for c in range(C):
            r = refs[c].process(orc.autocast_cu8_cs16(x[c]) if cu8 else x[c])
            if epi == sa.EPI_FM:
                r = fms[c].process(r) if len(r) else np.zeros(0, np.int16)
            elif epi == sa.EPI_AM:
                r = orc.am_i16(r)
            elif epi == sa.EPI_USB:
                r = orc.usb_i16(r)
            assert y[c].shape == r.shape and np.array_equal(y[c], r), (n, c)


@pytest.mark.parametrize("hot", [True, False])
@pytest.mark.parametrize("mode,Ff,width,epi", [("USB", 1500.0, 3e3, sa.EPI_USB), ("LSB", -1500.0, 3e3, sa.EPI_USB), ("AM", 0.0, 15e3, sa.EPI_AM),
                                               ("USB-shifted", 1500.0, 3e3, sa.EPI_USB)])
@pytest.mark.parametrize("cu8", [True, False])
def test_iqbb_sdr_rec_ssb_am_modes_vs_oracle(ctx, orc, mode, Ff, width, epi, cu8, hot, monkeypatch):
    """examples/sdr_rec.cc:50-58 in its AM / USB / LSB modes: f_center = 0 (NO frequency shift) but the 16-tap filter centred
    on f_filter = +-1500 Hz — complex taps without a rotation, the unshifted any-decimation form with 18-bit values in both
    components — at 1 MS/s to 12 kS/s (decimation 83), demodulated by USBDemod / AMDemod; "USB-shifted": the same filter
    with a 100 kHz shift in front (filter centre and shift differ). Ragged calls, state carried, both inputs."""
    monkeypatch.setenv("SDRHIP_IQBB_HOT", "1" if hot else "0")
    monkeypatch.delenv("SDRHIP_IQBB_PATH", raising=False)
    FSr, C, D = 1e6, 3, 83
    Fc = 100e3 if mode == "USB-shifted" else 0.0
    rng = np.random.default_rng(len(mode) + int(cu8))
    taps, lut, inc = orc.iqbb_design(Ff, width, FSr, 16), orc.freqshift_lut_i16(), orc.freqshift_inc(Fc, FSr)
    if Ff:
        assert np.any(taps[:, 1] != 0)
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, False, D, channels=C, max_in=70000, epilogue=epi)
    if cu8:
        node.set_input_format(sa.abi.IN_CU8)
    assert node.kernel_names == (["iqbb_hot_anyd_kernel"] if hot else ["iqbb_i16_mfmag_kernel"])
    refs = [orc.IQBaseBandI16(taps, lut, inc, False, D) for _ in range(C)]
    for n in (65536, 70000, 777, 40001, 65536):
        x = rng.integers(0, 256, (C, n, 2), dtype=np.uint8) if cu8 else rng.integers(-32768, 32768, (C, n, 2), dtype=np.int16)
        y = node.process(x)
        for c in range(C):
            r = refs[c].process(orc.autocast_cu8_cs16(x[c]) if cu8 else x[c])
            r = orc.am_i16(r) if epi == sa.EPI_AM else orc.usb_i16(r)
            assert y[c].shape == r.shape and np.array_equal(y[c], r), (mode, n, c)


def test_iqbb_any_decimation_full_size(ctx, orc):
    """The sdr_fm plan (21 taps, decimation 125, complex<uint8> input, FM) at the BASELINE batch: 1024 channels x 65536
    samples, two calls: batching invariance over all channels, 8 patterns against the oracle."""
    C, N, D = 1024, 65536, 125
    FSr = 1e6
    taps, lut, inc = orc.iqbb_design(100e3, 12.5e3, FSr, 21), orc.freqshift_lut_i16(), orc.freqshift_inc(100e3, FSr)
    rng = np.random.default_rng(77)
    base = rng.integers(0, 256, (8, 2 * N, 2), dtype=np.uint8)
    x = np.ascontiguousarray(base[np.arange(C) % 8])
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, False, D, channels=C, max_in=N, epilogue=sa.EPI_FM)
    node.set_input_format(sa.abi.IN_CU8)
    # (1024 channels deal evenly over the 4 x 256 workgroups of the persistent grid: a unit is a channel, and the hot kernel
    # completes the slices' first angle differences itself — one launch; a part with another CU count may add the fix-up)
    assert node.kernel_names[0] == "iqbb_hot_anyd_kernel"
    import torch
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert node.kernel_names == ["iqbb_hot_anyd_kernel"]
    ys = [node.process(x[:, :N]), node.process(x[:, N:])]
    for y in ys:
        for k in range(8):
            assert (y[k::8] == y[k]).all()
    for k in range(8):
        bb, fm = orc.IQBaseBandI16(taps, lut, inc, False, D), orc.FMDemodI16()
        for i, y in enumerate(ys):
            assert np.array_equal(y[k], fm.process(bb.process(orc.autocast_cu8_cs16(base[k, i * N:(i + 1) * N])))), (k, i)


@pytest.mark.parametrize("epi,resident", [(sa.EPI_FM, False), (sa.EPI_FM, True), (sa.EPI_NONE, False)])
def test_iqbb_any_decimation_more_channels_than_workgroups(ctx, orc, epi, resident, monkeypatch):
    """More channels (1100) than the persistent grid has workgroups (1024): the hot units and the cold slices of the any-D
    form wrap around. 16 taps at decimation 62 on complex<int16>, two calls (the second starts inside a group). FM with
    units of 4 tiles + the fix-up launch (what 1100 channels get), and with whole channels as units (forced: some workgroups
    then walk two channels and complete both themselves)."""
    handshake = False
    monkeypatch.setenv("SDRHIP_IQBB_FM_RESIDENT", "1" if resident else "0")
    monkeypatch.delenv("SDRHIP_IQBB_FM_HANDSHAKE", raising=False)
    C, N, D = 1100, 40001, 62
    FSr = 1e6
    taps, lut, inc = orc.iqbb_design(100e3, 12.5e3, FSr, 16), orc.freqshift_lut_i16(), orc.freqshift_inc(-100e3, FSr)
    rng = np.random.default_rng(78)
    base = rng.integers(-32768, 32768, (8, 2 * N, 2), dtype=np.int16)
    x = np.ascontiguousarray(base[np.arange(C) % 8])
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, True, D, channels=C, max_in=N, epilogue=epi)
    assert node.kernel_names == ["iqbb_hot_anyd_kernel"] + (["iqbb_fm_fixup_kernel"] if epi == sa.EPI_FM and not resident and not handshake else [])
    ys = [node.process(x[:, :N]), node.process(x[:, N:])]
    for y in ys:
        for k in range(8):
            assert (y[k::8] == y[k]).all()
    for k in range(8):
        bb, fm = orc.IQBaseBandI16(taps, lut, inc, True, D), orc.FMDemodI16()
        for i, y in enumerate(ys):
            r = bb.process(base[k, i * N:(i + 1) * N])
            if epi == sa.EPI_FM:
                r = fm.process(r)
            assert np.array_equal(y[k], r), (k, i)


@pytest.mark.parametrize("C", [1, 16, 128, 1100])
@pytest.mark.parametrize("order,decim,Fc,cu8", [(21, 125, 100e3, True), (16, 20, 0.0, True), (21, 4, 100e3, True), (127, 125, -100e3, False)])
def test_iqbb_fm_at_any_channel_count(ctx, orc, C, order, decim, Fc, cu8, monkeypatch):
    """The reference's graphs are ONE channel (examples/sdr_fm.cc:40-43: 21 taps, /125; sdr_rec.cc:66-72): FM at a decimation
    other than 8 on 1, 16, 128 and 1100 channels — the hot kernel + the tiny fix-up launch (or, in a -DK1_FM_HANDSHAKE build under
    SDRHIP_IQBB_FM_HANDSHAKE=1, ONE launch: the neighbouring slices' handshake) — with ragged calls (a call that ends inside
    a group, a one-sample call) and the state carried across them, bit-exact against the oracle on every channel. (The
    handshake measured no faster at any channel count — profiles/r17_ab_fm_handshake.txt — and is not in the shipped build.)"""
    for k in ("SDRHIP_IQBB_HOT", "SDRHIP_IQBB_FM_RESIDENT", "SDRHIP_IQBB_PATH"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SDRHIP_IQBB_FM_HANDSHAKE", "1")
    FSr = 1e6
    taps, lut, inc = orc.iqbb_design(abs(Fc), 12.5e3, FSr, order), orc.freqshift_lut_i16(), orc.freqshift_inc(Fc, FSr)
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, Fc < 0, decim, channels=C, max_in=65536, epilogue=sa.EPI_FM)
    if cu8:
        node.set_input_format(sa.abi.IN_CU8)
    hot_name = "iqbb_hot_sd_kernel" if decim < 8 else "iqbb_hot_anyd_kernel"
    assert node.kernel_names in ([hot_name], [hot_name, "iqbb_fm_fixup_kernel"])   # (one launch only in a handshake build)
    rng = np.random.default_rng(C * 1000 + decim)
    nb = min(C, 8)
    refs = [(orc.IQBaseBandI16(taps, lut, inc, Fc < 0, decim), orc.FMDemodI16()) for _ in range(nb)]
    for n in (65536, 40001, 1, 65536, 12345, 65536):
        base = rng.integers(0, 256, (nb, n, 2), dtype=np.uint8) if cu8 else rng.integers(-32768, 32768, (nb, n, 2), dtype=np.int16)
        x = np.ascontiguousarray(base[np.arange(C) % nb])
        y = node.process(x)
        for k in range(nb):
            bb, fm = refs[k]
            r = bb.process(orc.autocast_cu8_cs16(base[k]) if cu8 else base[k])
            r = fm.process(r) if len(r) else np.zeros(0, np.int16)
            assert y[k].shape == r.shape and np.array_equal(y[k], r), (n, k)
            assert (y[k::nb] == y[k]).all(), (n, k)


def test_big_lds_plans_on_a_second_device(orc):
    """HIP function attributes are per device: the plans that need more than 64 KB of dynamic LDS (255 taps at /125 — 17 K
    steps in a 16-wave workgroup —, the small-decimation form, the large-decimation form) raise the limit once per DEVICE.
    Runs them on device 1 of a process that has already run them on device 0 (needs two GPUs: skipped on the one-GPU boxes)."""
    if sa.device_count() < 2:
        pytest.skip("one HIP device")
    FSr = 1e6
    rng = np.random.default_rng(5)
    for order, decim in ((255, 125), (255, 4), (21, 300)):
        taps, lut, inc = orc.iqbb_design(100e3, 12.5e3, FSr, order), orc.freqshift_lut_i16(), orc.freqshift_inc(100e3, FSr)
        x = rng.integers(-32768, 32768, (2, 65536, 2), dtype=np.int16)
        ref = orc.FMDemodI16().process(orc.IQBaseBandI16(taps, lut, inc, False, decim).process(x[1]))
        for dev in (0, 1):
            c = sa.Context(dev)
            node = sa.IQBaseBandI16(c, taps, lut, inc, False, decim, channels=2, max_in=65536, epilogue=sa.EPI_FM)
            y = node.process(x)
            assert np.array_equal(y[1], ref), (order, decim, dev)
            node.close()
            c.close()


def test_iqbb_random_fullscale_vs_oracle(ctx, orc, k1path):
    """Full-range random int16 input (worst case for the int32 accumulators and the >>14/>>16 steps)."""
    rng = np.random.default_rng(7)
    C, N = 3, 20000
    x = rng.integers(-32768, 32768, size=(C, N, 2)).astype(np.int16)
    taps = sa.design_iqbb_taps(100e3, 50e3, FS, 127)
    lut = sa.design_freqshift_lut_i16()
    node = sa.IQBaseBandI16(ctx, taps, lut, 1365, False, 8, channels=C, max_in=N, epilogue=sa.EPI_NONE)
    y = node.process(x)
    for c in range(C):
        assert np.array_equal(y[c], orc.IQBaseBandI16(taps, lut, 1365, False, 8).process(x[c]))


def test_iqbb_path_selection(ctx, golden, monkeypatch):
    monkeypatch.delenv("SDRHIP_IQBB_PATH", raising=False)
    taps, lut = golden.load("g3_iqbb127d8_taps"), golden.load("g3_iqbb127d8_lut")
    assert sa.IQBaseBandI16(ctx, taps, lut, 1365, 0, 8).path == 1          # north-star chain -> MFMA
    assert sa.IQBaseBandI16(ctx, taps, lut, 1365, 0, 5).path == 3          # other decimations, long filter -> MFMA + LDS windows
    assert sa.IQBaseBandI16(ctx, golden.load("g8_o21_d3_taps"), lut, 1365, 0, 3).path == 3
    assert sa.IQBaseBandI16(ctx, golden.load("g8_o255_d8_taps"), lut, 1365, 0, 8).path == 1   # 17 K steps
    long300 = sa.IQBaseBandI16(ctx, sa.design_iqbb_taps(100e3, 50e3, FS, 300), lut, 1365, 0, 8)   # orders 258 ... 513: 33 K steps, hot forms only
    assert long300.path == 1 and long300.plan_info["S"] == 33 and long300.kernel_names == ["iqbb_hot_kernel"]
    assert sa.IQBaseBandI16(ctx, sa.design_iqbb_taps(100e3, 50e3, FS, 300), lut, 1365, 0, 4).path == 0    # ... and no small-decimation form
    assert sa.IQBaseBandI16(ctx, sa.design_iqbb_taps(100e3, 50e3, FS, 600), lut, 1365, 0, 8).path == 0    # order > 513
    monkeypatch.setenv("SDRHIP_IQBB_HOT", "0")
    assert sa.IQBaseBandI16(ctx, sa.design_iqbb_taps(100e3, 50e3, FS, 300), lut, 1365, 0, 8).path == 0    # (no general matrix kernel for that class)
    monkeypatch.delenv("SDRHIP_IQBB_HOT")
    big = np.array(taps).reshape(-1, 2).copy(); big[5, 0] = 32700          # high byte would not fit int8
    assert sa.IQBaseBandI16(ctx, big, lut, 1365, 0, 8).path == 0
    monkeypatch.setenv("SDRHIP_IQBB_PATH", "valu")
    assert sa.IQBaseBandI16(ctx, taps, lut, 1365, 0, 8).path == 0
    assert sa.IQBaseBandI16(ctx, taps, lut, 1365, 0, 5).path == 0


def test_iqbb_reset_semantics(ctx, orc, golden, k1path):
    m, node = iqbb_from_case(ctx, golden, "g3_iqbb127d8", "_out", sa.EPI_NONE)
    x = golden.load("g1_iq_cs16")
    first = node.process(x[:4096])[0]
    node.reset(keep_history=False)
    assert np.array_equal(node.process(x[:4096])[0], first)
    # _reconfigure keeps the ring (src/baseband.hh:175-177): same as the oracle's reset()
    bb = orc.IQBaseBandI16(golden.load("g3_iqbb127d8_taps"), golden.load("g3_iqbb127d8_lut"), 1365, 0, 8)
    bb.process(x[:4096]); bb.reset()
    node.reset(keep_history=False); node.process(x[:4096]); node.reset(keep_history=True)
    assert np.array_equal(node.process(x[4096:8192])[0], bb.process(x[4096:8192]))


# ---- the int8 chain (reference src/sdr.hh:225-240): IQBaseBand<int8_t> -> FMDemod<int8_t,int16_t> ------

@pytest.mark.parametrize("case", ["g13_i8_o21_d8", "g13_i8_doc_o16", "g13_i8_neg_o33_d5"])
def test_iqbb_i8_chain_golden(ctx, golden, case):
    m = golden.meta(case + "_out")
    x = golden.load("g13_iq_cs8").reshape(-1, 2)
    Fs = float(int(m["Fs"]))
    taps, lut, inc = sa.design_iqbb_taps(m["Ff"], m["width"], Fs, m["order"]), sa.design_freqshift_lut_i8(), sa.design_freqshift_inc(m["Fc"], Fs)
    node = sa.IQBaseBandI8(ctx, taps, lut, inc, m["Fc"] < 0, m["decim"], max_in=4096)
    outs = [node.process(c)[0] for c in split(x, m["in_lens"])]
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs).ravel(), golden.load(case + "_out"))
    # FM fused into the launch, and the stand-alone FMDemod<int8_t,int16_t> behind the complex<int8> output
    fused = sa.IQBaseBandI8(ctx, taps, lut, inc, m["Fc"] < 0, m["decim"], max_in=4096, epilogue=sa.EPI_FM)
    assert np.array_equal(np.concatenate([fused.process(c)[0] for c in split(x, m["in_lens"])]), golden.load(case + "_fm"))
    dem = sa.Demod(ctx, sa.EPI_FM, sa.abi.T_CS8, max_in=4096)
    fm = [dem.process(o)[0] for o in outs]
    assert np.array_equal(np.concatenate(fm), golden.load(case + "_fm"))


def test_iqbb_i8_batched_random_vs_oracle(ctx, orc):
    rng = np.random.default_rng(17)
    C, chunks = 5, [4096, 777, 1, 3000]
    x = rng.integers(-128, 128, size=(C, sum(chunks), 2)).astype(np.int8)
    for order, D, Fc in ((21, 8, 100e3), (127, 8, -250e3), (16, 3, 0.0), (64, 1, 50e3)):
        taps, lut, inc = sa.design_iqbb_taps(Fc, 50e3, FS, order), sa.design_freqshift_lut_i8(), sa.design_freqshift_inc(Fc, FS)
        node = sa.IQBaseBandI8(ctx, taps, lut, inc, Fc < 0, D, channels=C, max_in=4096, epilogue=sa.EPI_FM)
        refs = [(orc.IQBaseBandI8(taps, lut, inc, Fc < 0, D), orc.FMDemodI8()) for _ in range(C)]
        off = 0
        for n in chunks:
            y = node.process(x[:, off:off + n])
            for c in range(C):
                r = refs[c][0].process(x[c, off:off + n])
                r = refs[c][1].process(r) if len(r) else np.zeros(0, np.int16)
                assert np.array_equal(y[c], r), (order, D, c, n)
            off += n


@pytest.mark.parametrize("epi", [sa.EPI_NONE, sa.EPI_FM])
@pytest.mark.parametrize("order,decim,Fc", [(16, 24, 0.0), (21, 8, 100e3), (127, 8, -250e3), (64, 8, 0.0), (129, 8, 100e3), (16, 10, 0.0), (21, 125, 100e3),
                                             (33, 62, -60e3), (100, 300, 41e3), (127, 9, 100e3), (9, 512, 0.0), (130, 8, 100e3), (21, 5, 100e3)])
def test_iqbb_i8_long_calls_vs_oracle(ctx, orc, order, decim, Fc, epi, monkeypatch):
    """IQBaseBand<int8_t> (-> FMDemod<int8_t,int16_t>) on the matrix cores: the complex<int8> sample IS one signed byte plane, so
    the hot kernels take it without any conversion — decimation 8 and 9 ... 512, up to 129 taps (the reference's documented chain,
    src/sdr.hh:225-240, is 16 taps unshifted at 2.4 MS/s -> 100 kS/s: decimation 24); everything else, and the calls too short for a
    hot tile, the VALU kernel. Ragged long and short calls of full-scale random bytes, state carried."""
    monkeypatch.delenv("SDRHIP_IQBB_HOT", raising=False)
    monkeypatch.delenv("SDRHIP_IQBB_PATH", raising=False)
    C = 3
    rng = np.random.default_rng(order * 100 + decim)
    taps, lut, inc = sa.design_iqbb_taps(Fc if Fc else 100e3, 50e3, FS, order), sa.design_freqshift_lut_i8(), sa.design_freqshift_inc(Fc, FS)
    node = sa.IQBaseBandI8(ctx, taps, lut, inc, Fc < 0, decim, channels=C, max_in=70000, epilogue=epi)
    on_matrix = order <= 129 and (decim == 8 or 9 <= decim <= 512)
    if not on_matrix:
        assert node.path == 0 and node.kernel_names == ["iqbb_i16_kernel"]
    elif decim == 8:
        assert node.path == 1 and node.kernel_names == ["iqbb_hot_kernel"]
    else:
        assert node.path == 3 and node.kernel_names[0] == "iqbb_hot_anyd_kernel"
    refs = [(orc.IQBaseBandI8(taps, lut, inc, Fc < 0, decim), orc.FMDemodI8()) for _ in range(C)]
    for n in (65536, 70000, 12345, 1, 40001, 2 * decim + 1, 0, 65535):
        x = rng.integers(-128, 128, size=(C, n, 2)).astype(np.int8)
        y = node.process(x)
        for c in range(C):
            r = refs[c][0].process(x[c])
            if epi == sa.EPI_FM:
                r = refs[c][1].process(r) if len(r) else np.zeros(0, np.int16)
            assert y[c].shape == r.shape and np.array_equal(y[c], r), (n, c)


class _GpuRetune:
    def __init__(self, ctx, Ff, width, Fc, epi, order=127, D=8):
        self.order, self.ctx, self.epi, self.D, self.max_in = order, ctx, epi, D, 4096
        self.node = sa.IQBaseBandI16(ctx, sa.design_iqbb_taps(Ff, width, FS, order), sa.design_freqshift_lut_i16(),
                                     sa.design_freqshift_inc(Fc, FS), Fc < 0, D, max_in=4096, epilogue=epi)

    def process(self, x):
        return self.node.process(x)[0]

    def set_shift_hz(self, Fc):
        self.node.set_shift(sa.design_freqshift_inc(Fc, FS), Fc < 0)

    def set_filter(self, Ff, width):
        self.node.set_taps(sa.design_iqbb_taps(Ff, width, FS, self.order))

    def reconfigure(self):
        self.node.reset(keep_history=True, keep_fm=True)   # (the FMDemod behind the baseband sees no Config change)

    def _replan(self, Ff, width, Fc, D, max_in, what):
        """A new device plan for another geometry that takes the old one's streaming state over."""
        neu = sa.IQBaseBandI16(self.ctx, sa.design_iqbb_taps(Ff, width, FS, self.order), sa.design_freqshift_lut_i16(),
                               sa.design_freqshift_inc(Fc, FS), Fc < 0, D, max_in=max_in, epilogue=self.epi)
        neu.adopt_state(self.node, what)
        self.node, self.D, self.max_in = neu, D, max_in

    def regeometry(self, D, bufsize, Ff, width, Fc):
        # _reconfigure (src/baseband.hh:156-194): counters reset, ring kept where it lies; the FMDemod behind the node is
        # reset (every g14 event of this kind changes the Config it receives)
        self._replan(Ff, width, Fc, D, bufsize, sa.abi.KEEP_RING)

    def set_order(self, order, Ff, width, Fc):
        # setOrder (src/baseband.hh:69-79): kernel and ring only
        self.order = order
        # (KEEP_FM only between plans that both fuse the FM demodulator: the ABI refuses it otherwise, as the C++ node masks it)
        self._replan(Ff, width, Fc, self.D, self.max_in, (sa.abi.KEEP_FM if self.epi == sa.EPI_FM else 0) | sa.abi.KEEP_COUNTERS)


@pytest.mark.parametrize("which,epi", [("g12_retune_out", sa.EPI_NONE), ("g12_retune_fm", sa.EPI_FM)])
def test_iqbb_retune_midstream_golden(ctx, golden, which, epi, k1path):
    """The reference node retuned between buffers (setCenterFrequency: LUT phase restarts only; setFilterFrequency /
    setFilterWidth: kernel only; setSubsample: _reconfigure with the ring kept) — set_shift / set_taps / reset(keep)."""
    from test_oracle_golden import replay_retune
    m = golden.meta(which)
    outs = replay_retune(m, golden.load("g1_iq_cs16"), lambda Ff, w, Fc: _GpuRetune(ctx, Ff, w, Fc, epi))
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load(which))


@pytest.mark.parametrize("which,epi", [("g14_regeom_out", sa.EPI_NONE), ("g14_regeom_fm", sa.EPI_FM)])
def test_iqbb_regeometry_midstream_golden(ctx, golden, which, epi, k1path):
    """The reference node's GEOMETRY changed between buffers — setSubsample(4), setOutputSampleRate (÷24), a new source
    buffer size (all _reconfigure: the ring survives, rotated) and setOrder(161) (kernel and ring only; decimator, counters
    and LUT phase go on): every change needs a new device plan, which adopts the old plan's state
    (sdrhip_iqbb_i16_adopt_state). Every output the reference defines must match."""
    from test_oracle_golden import replay_retune, defined_mask
    m = golden.meta(which)
    ok = defined_mask(m)
    outs = replay_retune(m, golden.load("g1_iq_cs16"), lambda Ff, w, Fc: _GpuRetune(ctx, Ff, w, Fc, epi))
    assert [len(o) for o in outs] == m["out_lens"]
    want = golden.load(which)
    if epi == sa.EPI_NONE:
        want = want.reshape(-1, 2)
    assert np.array_equal(np.concatenate(outs)[ok], want[ok])


@pytest.mark.parametrize("cu8", [False, True])
def test_iqbb_adopt_state_vs_oracle(ctx, orc, k1path, cu8):
    """adopt_state over many channels and plan pairs (÷8 hot form -> any-D form -> small-decimation form -> long filter),
    ragged calls, against the oracle: _reconfigure's rotated ring for every ring position, setOrder's continuing decimator;
    complex<int16> and complex<uint8> (AutoCast fused: what sdr::gpu::IQBaseBand<uint8_t>'s setters go through)."""
    rng = np.random.default_rng(77)
    C, lut = 5, sa.design_freqshift_lut_i16()
    if cu8:
        xu = rng.integers(0, 256, (C, 40000, 2), dtype=np.uint8)
        x = np.stack([orc.autocast_cu8_cs16(xu[c]) for c in range(C)])
    else:
        x = rng.integers(-20000, 20000, (C, 40000, 2)).astype(np.int16)
    inc = sa.design_freqshift_inc(-150e3, FS)

    def make(taps_, D_, epi_):
        nd = sa.IQBaseBandI16(ctx, taps_, lut, inc, True, D_, channels=C, max_in=8192, epilogue=epi_)
        if cu8:
            nd.set_input_format(sa.abi.IN_CU8)
        return nd
    for epi in (sa.EPI_NONE, sa.EPI_FM, sa.EPI_USB):
        order, D = 127, 8
        taps = sa.design_iqbb_taps(-150e3, 60e3, FS, order)
        node = make(taps, D, epi)
        refs = [orc.IQBaseBandI16(taps, lut, inc, True, D) for _ in range(C)]
        fms = [orc.FMDemodI16() for _ in range(C)]
        off = 0
        steps = [("feed", 5000), ("geom", 12), ("feed", 6001), ("geom", 3), ("feed", 777), ("order", 200), ("feed", 8000),
                 ("geom", 8), ("feed", 4100), ("order", 300), ("feed", 5003), ("geom", 125), ("feed", 8192)]
        for kind, v in steps:
            if kind == "feed":
                y = node.process(xu[:, off:off + v] if cu8 else x[:, off:off + v])
                for c in range(C):
                    r = refs[c].process(x[c, off:off + v])
                    if epi == sa.EPI_FM:
                        r = fms[c].process(r)
                    elif epi == sa.EPI_USB:
                        r = orc.usb_i16(r)
                    assert np.array_equal(y[c], r), (epi, kind, v, off, c)
                off += v
            elif kind == "geom":
                D = v
                neu = make(taps, D, epi)
                neu.adopt_state(node, sa.abi.KEEP_RING)
                node = neu
                for c in range(C):
                    refs[c].set_decim(D); refs[c].set_taps(taps); refs[c].set_shift(inc, True); refs[c].reset()
                fms = [orc.FMDemodI16() for _ in range(C)]
            else:
                order = v
                taps = sa.design_iqbb_taps(-150e3, 60e3, FS, order)
                neu = make(taps, D, epi)
                neu.adopt_state(node, (sa.abi.KEEP_FM if epi == sa.EPI_FM else 0) | sa.abi.KEEP_COUNTERS)
                if epi != sa.EPI_FM:   # (the flag without the FM epilogue on both plans is an argument error)
                    with pytest.raises(sa.SdrHipError) as e:
                        make(taps, D, epi).adopt_state(node, sa.abi.KEEP_FM | sa.abi.KEEP_COUNTERS)
                    assert e.value.code == sa.abi.E_INVALID
                node = neu
                for c in range(C):
                    refs[c].set_order(taps)


# ---- "next" rows: cu8 input with AutoCast fused into the K1 load, FMDeemph behind the demodulator ----

@pytest.mark.parametrize("order", [21, 127])
def test_sdr_fm_chain_cu8_golden(ctx, golden, order, k1path):
    """cu8 -> [AutoCast + IQBaseBand + FMDemod in one launch] -> FMDeemph, against the reference chain."""
    name = "g9_cu8_iqbb%dd8" % order
    m = golden.meta(name + "_taps")
    u = golden.load("g9_iq_cu8").reshape(-1, 2)
    node = sa.IQBaseBandI16(ctx, golden.load(name + "_taps"), sa.design_freqshift_lut_i16(), m["lut_inc"], 0, 8,
                            max_in=4096, epilogue=sa.EPI_FM)
    node.set_input_format(sa.abi.IN_CU8)
    alpha = sa.design_fmdeemph_alpha(1e6 / 8)
    assert alpha == 10
    de = sa.FMDeemphI16(ctx, alpha, max_in=4096)
    f = [node.process(u[b * 4096:(b + 1) * 4096])[0] for b in range(3)]
    assert np.array_equal(np.concatenate(f), golden.load(name + "_fm"))
    d = np.concatenate([de.process(x)[0] for x in f])
    assert np.array_equal(d, golden.load(name + "_fm_deemph"))


def test_cu8_batched_random_vs_oracle(ctx, orc, k1path):
    rng = np.random.default_rng(21)
    C, chunks = 3, [4096, 1, 777, 4096]
    u = rng.integers(0, 256, size=(C, sum(chunks), 2), dtype=np.uint8)
    taps = sa.design_iqbb_taps(-100e3, 50e3, FS, 127)
    lut, inc = sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(-100e3, FS)
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, True, 8, channels=C, max_in=4096, epilogue=sa.EPI_NONE)
    node.set_input_format(sa.abi.IN_CU8)
    refs = [orc.IQBaseBandI16(taps, lut, inc, True, 8) for _ in range(C)]
    off = 0
    for n in chunks:
        y = node.process(u[:, off:off + n])
        for c in range(C):
            assert np.array_equal(y[c], refs[c].process(orc.autocast_cu8_cs16(u[c, off:off + n])))
        off += n
    with pytest.raises(sa.SdrHipError):
        node.set_input_format(sa.abi.IN_CS16)       # only before the first buffer / after a reset


# ---- "next" row 3: BaseBand<int16_t>, real input (reference src/baseband.hh:305-529) ----------------------------

BB_REAL_CASES = [("g10_bb21d8", "g10_real_in"), ("g10_bb127d8_neg_ragged", "g10_real_in"), ("g10_bb64d5", "g10_real_in"),
                 ("g10_bb16d1_noshift", "g10_real_in"), ("g10_bb1d3", "g10_real_in"), ("g10_bb127d8_loud", "g10_real_loud_in")]


@pytest.fixture(params=["auto", "general", "valu"])
def bbpath(request, monkeypatch):
    """The real-input node has three bit-exact kernels: the int8-MFMA formulation (path 4: decimation 8, taps within two
    byte planes, up to 273 taps) as the hot kernel's real-input instantiation (calls of >= 3 tiles) and as the general
    kernel (short calls, and alone with SDRHIP_IQBB_HOT=0), and the VALU kernel (everything else, and on request)."""
    monkeypatch.delenv("SDRHIP_IQBB_PATH", raising=False)
    monkeypatch.delenv("SDRHIP_IQBB_HOT", raising=False)
    if request.param == "valu":
        monkeypatch.setenv("SDRHIP_IQBB_PATH", "valu")
    elif request.param == "general":
        monkeypatch.setenv("SDRHIP_IQBB_HOT", "0")
    return request.param


BB_REAL_KERNEL = {"auto": ["iqbb_hot_kernel"], "general": ["bb_real_mfma_kernel"], "valu": ["iqbb_i16_kernel"]}


def test_bb_real_retune_midstream_golden(ctx, golden, bbpath):
    """g16: the reference's real-input node through setFrequencyShift (LUT increment, sign and phase; nothing else) and a
    new source Config with another buffer size (BaseBand::config: counters and LUT phase restart, the ring's contents stay
    where they lie — a new device plan that adopts the old plan's ring), between buffers."""
    from test_oracle_golden import replay_bb_real_retune
    m = golden.meta("g16_bb_real_retune_out")
    Fs = float(m["Fs"])
    taps, lut = sa.design_bb_taps(m["Ff"], m["width"], Fs, m["order"]), sa.design_freqshift_lut_i16()

    class Node:
        def __init__(self, F, bufsize=4096, old=None):
            self.node = sa.BaseBandI16(ctx, taps, lut, sa.design_freqshift_inc(F, Fs), F < 0, m["decim"], max_in=bufsize)
            if old is not None:
                self.node.adopt_state(old.node, sa.abi.KEEP_RING)

        def process(self, x):
            return self.node.process(x)[0]

        def set_shift_hz(self, F):
            self.node.set_shift(sa.design_freqshift_inc(F, Fs), F < 0)

        def reconfigured(self, bufsize, F):
            return Node(F, bufsize, self)

    outs = replay_bb_real_retune(m, golden.load("g10_real_in"), Node)
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load("g16_bb_real_retune_out"))


@pytest.mark.parametrize("case,inp", BB_REAL_CASES)
def test_bb_real_golden(ctx, golden, case, inp, bbpath):
    m = golden.meta(case + "_out")
    assert np.array_equal(sa.design_bb_taps(m["Ff"], m["width"], m["Fs"], m["order"]), golden.load(case + "_taps").reshape(-1, 2))
    bb = sa.BaseBandI16(ctx, golden.load(case + "_taps"), sa.design_freqshift_lut_i16(), m["lut_inc"], m["negative"], m["decim"],
                        max_in=4096)
    if m["decim"] == 8:
        assert bb.kernel_names == BB_REAL_KERNEL[bbpath]
    x = golden.load(inp)
    outs, off = [], 0
    for n in m["in_lens"]:
        outs.append(bb.process(x[off:off + n])[0]); off += n
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load(case + "_out"))


@pytest.mark.parametrize("epi", [sa.EPI_NONE, sa.EPI_FM, sa.EPI_USB])
@pytest.mark.parametrize("order,decim,Fc", [(127, 8, 100e3), (33, 5, -80e3), (16, 1, 0.0), (255, 12, 100e3)])
def test_bb_real_batched_vs_oracle(ctx, orc, epi, order, decim, Fc, bbpath):
    """5 channels of full-scale random real samples in ragged calls; demodulators chained as the reference would."""
    Fs, C = 1e6, 5
    rng = np.random.default_rng(order * 7 + decim)
    taps, lut, inc = orc.bb_design(abs(Fc) if Fc else 120e3, 60e3, Fs, order), orc.freqshift_lut_i16(), orc.freqshift_inc(Fc, Fs)
    bb = sa.BaseBandI16(ctx, taps, lut, inc, Fc < 0, decim, channels=C, max_in=3000, epilogue=epi)
    refs = [orc.BaseBandI16(taps, lut, inc, Fc < 0, decim) for _ in range(C)]
    fms = [orc.FMDemodI16() for _ in range(C)]
    for n in (3000, 1, 0, 777, 2048, 13):
        x = rng.integers(-32768, 32768, (C, n), dtype=np.int16)
        y = bb.process(x)
        for c in range(C):
            r = refs[c].process(x[c])
            if epi == sa.EPI_FM:
                r = fms[c].process(r)
                assert np.array_equal(y[c], r)              # index 0 = y[0].re, what the in-place buffer holds
            elif epi == sa.EPI_USB:
                assert np.array_equal(y[c], orc.usb_i16(r))
            else:
                assert np.array_equal(y[c], r)


def test_bb_real_reset_semantics(ctx, orc, bbpath):
    Fs, order = 1e6, 21
    rng = np.random.default_rng(5)
    taps, lut, inc = orc.bb_design(100e3, 50e3, Fs, order), orc.freqshift_lut_i16(), orc.freqshift_inc(100e3, Fs)
    bb = sa.BaseBandI16(ctx, taps, lut, inc, 0, 8, max_in=1000)
    ref = orc.BaseBandI16(taps, lut, inc, 0, 8)
    x = rng.integers(-20000, 20000, (3, 1000), dtype=np.int16)
    assert np.array_equal(bb.process(x[0])[0], ref.process(x[0]))
    bb.reset(keep_history=True); ref.reset()            # what config() does: counters reset, ring kept (and rotated)
    assert np.array_equal(bb.process(x[1])[0], ref.process(x[1]))
    bb.reset(keep_history=False); ref = orc.BaseBandI16(taps, lut, inc, 0, 8)
    assert np.array_equal(bb.process(x[2])[0], ref.process(x[2]))


@pytest.mark.parametrize("epi", [sa.EPI_NONE, sa.EPI_FM, sa.EPI_AM])
@pytest.mark.parametrize("order,Fc", [(127, 100e3), (64, -60e3), (9, 0.0), (273, 100e3)])
def test_bb_real_long_calls_vs_oracle(ctx, orc, epi, order, Fc, bbpath):
    """Real-input calls long enough for interior tiles of the MFMA formulation (2016- or 2048-sample tiles, windows by
    2-byte aligned 16-byte loads: odd call lengths move them over every alignment), retuned mid-stream, state carried."""
    Fs, C, chunks = 1e6, 3, [20000, 13333, 7, 1, 9999, 4097]
    rng = np.random.default_rng(order + 17)
    taps, lut, inc = orc.bb_design(abs(Fc) if Fc else 120e3, 60e3, Fs, order), orc.freqshift_lut_i16(), orc.freqshift_inc(Fc, Fs)
    bb = sa.BaseBandI16(ctx, taps, lut, inc, Fc < 0, 8, channels=C, max_in=max(chunks), epilogue=epi)
    assert bb.kernel_names == BB_REAL_KERNEL[bbpath]
    refs = [orc.BaseBandI16(taps, lut, inc, Fc < 0, 8) for _ in range(C)]
    fms = [orc.FMDemodI16() for _ in range(C)]
    for n in chunks:
        x = rng.integers(-32768, 32768, (C, n), dtype=np.int16)
        y = bb.process(x)
        for c in range(C):
            r = refs[c].process(x[c])
            if epi == sa.EPI_FM:
                assert np.array_equal(y[c], fms[c].process(r))
            elif epi == sa.EPI_AM:
                assert np.array_equal(y[c], orc.am_i16(r))
            else:
                assert np.array_equal(y[c], r)


MULTI_CASES = [("cs16", 127, 8, 100e3), ("cs16", 21, 8, -100e3), ("cu8", 21, 125, 100e3), ("cs16", 16, 83, 0.0), ("cs16", 33, 5, 70e3),
               ("cu8", 64, 600, 41e3), ("cs16", 255, 20, 100e3), ("real", 127, 8, 100e3), ("real", 127, 20, -60e3), ("real", 33, 5, 0.0),
               ("cs16", 16, 1, 50e3), ("cs16", 300, 8, 100e3)]


@pytest.mark.parametrize("epi", [sa.EPI_NONE, sa.EPI_FM, sa.EPI_AM, sa.EPI_USB])
@pytest.mark.parametrize("kind,order,decim,Fc", MULTI_CASES)
def test_iqbb_multi_buffer_call_equals_separate_calls(ctx, orc, kind, order, decim, Fc, epi):
    """sdrhip_iqbb_i16_process_dev_multi: B reference-sized buffers per channel in ONE launch, buffer boundaries kept — the
    outputs are those of B successive calls (FMDemod starts every buffer anew: index 0 in place, index 1 from the previous
    buffer's last angle, src/demod.hh:242-254), against the ORACLE fed buffer by buffer; then the stream goes on with
    ordinary calls (state handed over), and with multi calls whose buffers are too short for one launch (fallback)."""
    FSr, C = 1e6, 3
    rng = np.random.default_rng(order * 31 + decim)
    real, cu8 = kind == "real", kind == "cu8"
    lut, inc = orc.freqshift_lut_i16(), orc.freqshift_inc(Fc, FSr)
    if real:
        taps = orc.bb_design(abs(Fc) if Fc else 120e3, 60e3, FSr, order)
        node = sa.BaseBandI16(ctx, taps, lut, inc, Fc < 0, decim, channels=C, max_in=70000, epilogue=epi)
        refs = [orc.BaseBandI16(taps, lut, inc, Fc < 0, decim) for _ in range(C)]
    else:
        taps = orc.iqbb_design(abs(Fc) if Fc else 100e3, 12.5e3, FSr, order)
        node = sa.IQBaseBandI16(ctx, taps, lut, inc, Fc < 0, decim, channels=C, max_in=70000, epilogue=epi)
        if cu8:
            node.set_input_format(sa.abi.IN_CU8)
        refs = [orc.IQBaseBandI16(taps, lut, inc, Fc < 0, decim) for _ in range(C)]
    fms = [orc.FMDemodI16() for _ in range(C)]

    def gen(n):
        if real:
            return rng.integers(-32768, 32768, (C, n), dtype=np.int16)
        return rng.integers(0, 256, (C, n, 2), dtype=np.uint8) if cu8 else rng.integers(-32768, 32768, (C, n, 2), dtype=np.int16)

    def ref_buffer(c, xb):
        r = refs[c].process(orc.autocast_cu8_cs16(xb) if cu8 else xb)
        if epi == sa.EPI_FM:
            return fms[c].process(r) if len(r) else np.zeros(0, np.int16)
        return orc.am_i16(r) if epi == sa.EPI_AM else orc.usb_i16(r) if epi == sa.EPI_USB else r

    # (B, n_per_buffer): four reference-sized buffers; odd sizes (groups straddle the boundaries); an ordinary call in between;
    # buffers too short for FM's two-output rule (one launch per buffer inside the library); one buffer
    for B, nb in ((4, 16384), (3, 23333), (0, 5000), (5, 2 * decim + 3), (7, max(1, decim // 2)), (1, 30000), (2, 35000)):
        if B == 0:
            x = gen(nb)
            y = node.process(x)
            for c in range(C):
                r = ref_buffer(c, x[c])
                assert y[c].shape == r.shape and np.array_equal(y[c], r)
            continue
        x = gen(B * nb)
        y, counts = node.process_multi(x, B)
        if B == 4 and nb == 16384 and epi == sa.EPI_FM and decim == 8 and order <= 257:
            # equal buffers at decimation 8: every boundary lies inside a hot slice — the hot kernel wrote them, no second launch
            assert node.plan_info["multi_left"] == 0
        elif B >= 2 and epi == sa.EPI_FM and decim != 8 and nb >= 3 * decim:
            assert node.plan_info["multi_left"] == B - 1
        for c in range(C):
            rs = [ref_buffer(c, x[c, j * nb:(j + 1) * nb]) for j in range(B)]
            assert counts == [len(r) for r in rs], (B, nb, counts)
            r = np.concatenate(rs) if rs else np.zeros(0, np.int16)
            assert y[c].shape == r.shape, (B, nb, c)
            if not np.array_equal(y[c], r):
                bad = np.flatnonzero((y[c] != r).reshape(len(r), -1).any(axis=1))
                raise AssertionError("multi call B=%d nb=%d channel %d: %d outputs differ, first at %d (buffer starts %s)"
                                     % (B, nb, c, bad.size, bad[0], np.cumsum([0] + counts).tolist()))


@pytest.mark.parametrize("demod", ["fm", "am", "usb"])
def test_iqbb_multi_buffer_golden(ctx, golden, demod):
    """g4: the reference's own IQBaseBand<int16>(127, /8) -> FMDemod / AMDemod / USBDemod output of four 4096-sample buffers,
    through ONE multi-buffer call."""
    epi = {"fm": sa.EPI_FM, "am": sa.EPI_AM, "usb": sa.EPI_USB}[demod]
    m, node = iqbb_from_case(ctx, golden, "g4_iqbb127d8", "_" + demod, epi, max_in=16384)
    assert m["in_lens"] == [4096] * 4
    y, counts = node.process_multi(golden.load("g1_iq_cs16")[:16384][None], 4)
    assert counts == m["out_lens"]
    assert np.array_equal(y[0], golden.load("g4_iqbb127d8_" + demod))


BB_REAL_ANYD_CASES = [(127, 5, 100e3), (127, 20, 100e3), (127, 125, 100e3), (21, 125, -100e3), (33, 9, 41e3), (64, 62, 0.0), (81, 83, 100e3),
                      (145, 300, -60e3), (146, 12, 100e3), (273, 45, 0.0), (200, 512, 100e3), (16, 257, 0.0), (100, 180, 70e3), (9, 181, 100e3),
                      # decimations 1 ... 7: the small-decimation form (9 K steps WITHOUT a shift: its two sample arrays do not fit
                      # the real-input kernel's 4-wave workgroup — the VALU kernel keeps that plan)
                      (21, 1, 100e3), (127, 2, -60e3), (64, 3, 0.0), (81, 4, 100e3), (145, 6, 0.0), (200, 7, 100e3), (16, 5, 0.0), (273, 3, 0.0)]


@pytest.mark.parametrize("epi,resident", [(sa.EPI_NONE, False), (sa.EPI_FM, False), (sa.EPI_FM, True), (sa.EPI_AM, False), (sa.EPI_USB, False)])
@pytest.mark.parametrize("order,decim,Fc", BB_REAL_ANYD_CASES)
def test_bb_real_any_decimation_long_calls_vs_oracle(ctx, orc, order, decim, Fc, epi, resident, monkeypatch):
    """BaseBand<int16_t> takes any sub_sample (src/baseband.hh:305-529): up to 273 taps within two byte planes and decimations
    up to 512 run the hot kernel's any-D / small-decimation forms on the matrix cores (the real-input tile hands every lane 8
    consecutive samples, as the permuted complex tile does: one epilogue for both), the VALU kernel only the calls too
    short to hold a hot tile. Ragged long and short calls, odd lengths (2-byte aligned windows), state carried."""
    monkeypatch.delenv("SDRHIP_IQBB_HOT", raising=False)
    monkeypatch.delenv("SDRHIP_IQBB_PATH", raising=False)
    monkeypatch.setenv("SDRHIP_IQBB_FM_RESIDENT", "1" if resident else "0")
    FSr, C = 1e6, 3
    rng = np.random.default_rng(order * 1000 + decim + 7)
    taps, lut, inc = orc.bb_design(abs(Fc) if Fc else 120e3, 60e3, FSr, order), orc.freqshift_lut_i16(), orc.freqshift_inc(Fc, FSr)
    node = sa.BaseBandI16(ctx, taps, lut, inc, Fc < 0, decim, channels=C, max_in=70000, epilogue=epi)
    assert node.path == 4
    no_form = decim < 8 and order > 145 and Fc == 0.0
    hot_name = "iqbb_hot_sd_kernel" if decim < 8 else "iqbb_hot_anyd_kernel"
    expect = ["iqbb_i16_kernel"] if no_form else [hot_name] + (["iqbb_fm_fixup_kernel"] if epi == sa.EPI_FM and not resident else [])
    assert node.kernel_names == expect
    refs = [orc.BaseBandI16(taps, lut, inc, Fc < 0, decim) for _ in range(C)]
    fms = [orc.FMDemodI16() for _ in range(C)]
    for n in (65536, 70000, 12345, 1, 40001, 2 * decim + 1, 65535, 0, 33333):
        x = rng.integers(-32768, 32768, (C, n), dtype=np.int16)
        y = node.process(x)
        for c in range(C):
            r = refs[c].process(x[c])
            if epi == sa.EPI_FM:
                r = fms[c].process(r) if len(r) else np.zeros(0, np.int16)
            elif epi == sa.EPI_AM:
                r = orc.am_i16(r)
            elif epi == sa.EPI_USB:
                r = orc.usb_i16(r)
            assert y[c].shape == r.shape and np.array_equal(y[c], r), (n, c)


def test_bb_real_wide_taps_fall_back(ctx, orc):
    """Q16 taps beyond two byte planes (only very short filters reach 2^15) keep the VALU kernel — and stay bit-exact."""
    Fs = 1e6
    taps = orc.bb_design(200e3, 900e3, Fs, 4)
    assert np.abs(taps).max() >= 32640
    lut, inc = orc.freqshift_lut_i16(), orc.freqshift_inc(200e3, Fs)
    bb = sa.BaseBandI16(ctx, taps, lut, inc, False, 8, max_in=5000)
    assert bb.kernel_names == ["iqbb_i16_kernel"]
    x = np.random.default_rng(3).integers(-32768, 32768, 5000, dtype=np.int16)
    assert np.array_equal(bb.process(x)[0], orc.BaseBandI16(taps, lut, inc, False, 8).process(x))


def test_bb_real_full_size_properties(ctx, orc):
    """1024 channels x 65536 real samples (the BASELINE batch): batching invariance + 8 oracle comparisons over 2 calls."""
    C, N, D = 1024, 65536, 8
    taps, lut, inc = orc.bb_design(100e3, 50e3, FS, 127), orc.freqshift_lut_i16(), orc.freqshift_inc(100e3, FS)
    base = np.ascontiguousarray(synth_channels(orc, 8, N)[..., 0])
    x = np.ascontiguousarray(base[np.arange(C) % 8])
    node = sa.BaseBandI16(ctx, taps, lut, inc, False, D, channels=C, max_in=N, epilogue=sa.EPI_FM)
    assert node.kernel_names == ["iqbb_hot_kernel"]
    y1, y2 = node.process(x), node.process(x)
    for k in range(8):
        assert (y1[k::8] == y1[k]).all() and (y2[k::8] == y2[k]).all()
        bb, fm = orc.BaseBandI16(taps, lut, inc, False, D), orc.FMDemodI16()
        assert np.array_equal(y1[k], fm.process(bb.process(x[k])))
        assert np.array_equal(y2[k], fm.process(bb.process(x[k])))


@pytest.mark.parametrize("epi", [sa.EPI_NONE, sa.EPI_FM])
def test_cu8_long_calls_vs_oracle(ctx, orc, epi, k1path):
    """complex<uint8> calls long enough for interior tiles (2048 samples each) of the one-plane MFMA instantiation:
    odd call lengths move the 2-byte aligned 8-byte loads over every alignment."""
    rng = np.random.default_rng(33)
    C, chunks = 2, [20000, 13333, 7, 9999]
    u = rng.integers(0, 256, size=(C, sum(chunks), 2), dtype=np.uint8)
    taps = sa.design_iqbb_taps(100e3, 50e3, FS, 127)
    lut, inc = sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(100e3, FS)
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, False, 8, channels=C, max_in=20000, epilogue=epi)
    node.set_input_format(sa.abi.IN_CU8)
    refs = [orc.IQBaseBandI16(taps, lut, inc, False, 8) for _ in range(C)]
    fms = [orc.FMDemodI16() for _ in range(C)]
    off = 0
    for n in chunks:
        y = node.process(u[:, off:off + n])
        for c in range(C):
            r = refs[c].process(orc.autocast_cu8_cs16(u[c, off:off + n]))
            if epi == sa.EPI_FM:
                r = fms[c].process(r)
            assert np.array_equal(y[c], r)
        off += n


def test_cu8_other_decimation_vs_oracle(ctx, orc, k1path):
    """complex<uint8> input through the any-decimation MFMA path (65 taps, D = 5) and, with k1path = valu, the VALU kernel."""
    rng = np.random.default_rng(8)
    C, chunks = 2, [9000, 4101, 3]
    u = rng.integers(0, 256, size=(C, sum(chunks), 2), dtype=np.uint8)
    taps = sa.design_iqbb_taps(-60e3, 40e3, FS, 65)
    lut, inc = sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(-60e3, FS)
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, True, 5, channels=C, max_in=9000, epilogue=sa.EPI_USB)
    node.set_input_format(sa.abi.IN_CU8)
    assert node.path == (0 if k1path == "valu" else 3)
    refs = [orc.IQBaseBandI16(taps, lut, inc, True, 5) for _ in range(C)]
    off = 0
    for n in chunks:
        y = node.process(u[:, off:off + n])
        for c in range(C):
            assert np.array_equal(y[c], orc.usb_i16(refs[c].process(orc.autocast_cu8_cs16(u[c, off:off + n]))))
        off += n


@pytest.mark.parametrize("rate", [125000, 48000])
def test_fmdeemph_golden_and_batched(ctx, golden, orc, rate):
    x = golden.load("g9_deemph_in")
    alpha = sa.design_fmdeemph_alpha(float(rate))
    de = sa.FMDeemphI16(ctx, alpha, max_in=512)
    y = np.concatenate([de.process(x[i * 512:(i + 1) * 512])[0] for i in range(3)])
    assert np.array_equal(y, golden.load("g9_deemph_out_%d" % rate))
    rng = np.random.default_rng(rate)
    C = 70                                             # more channels than one 64-lane workgroup, odd sizes
    z = rng.integers(-32768, 32768, size=(C, 333), dtype=np.int16)
    de = sa.FMDeemphI16(ctx, alpha, channels=C, max_in=400)
    refs = [orc.FMDeemphI16(float(rate)) for _ in range(C)]
    for lo, hi in ((0, 1), (1, 200), (200, 333)):
        got = de.process(z[:, lo:hi])
        for c in range(C):
            assert np.array_equal(got[c], refs[c].process(z[c, lo:hi]))


@pytest.mark.parametrize("data", ["noise", "fullscale", "constant", "steps", "quiet"])
@pytest.mark.parametrize("alpha,P,wc", [(2, None, None), (4, None, None), (10, None, None), (16, None, None), (4, 32, 0), (10, 8, 0), (10, 32, 1),
                                        (100, 16, 0), (100, 4, 2), (3, 2, 0), (32767, 8, 1), (2, 64, 0), (5, 64, 1)])
def test_fmdeemph_segmented_kernel_vs_oracle(ctx, orc, alpha, P, wc, data, monkeypatch):
    """FMDeemph<int16_t> with P lanes per channel (deemph_i16_spec_kernel): every lane but a channel's first starts its
    segment from a state it GUESSED by running over the samples in front of it; the kernel then checks every guess against
    the final state of the lane before and repeats what was wrong. Whatever the guesses are worth, the rows must be the
    sequential recursion's (src/demod.hh:342-351), bit for bit: rows that forget fast (noise), rows whose runs meet late
    (full-scale data wraps the int16 difference), rows that NEVER meet (constant rows and long steps sit inside the rounding
    dead zone: every lane repeats its whole segment, and its successor after it), with the default run-in, with none at all
    (SDRHIP_DEEMPH_WC=0: the guess is the segment's first sample) and with forced lane counts; unaligned rows, ragged call
    lengths, state carried across calls (short calls between the long ones run the one-lane kernel on the same state)."""
    dev = torch.device("cuda:0")
    C, ld = 37, 9001
    rng = np.random.default_rng(alpha * 7 + (P or 0))
    for k, v in (("SDRHIP_DEEMPH_SPEC", P), ("SDRHIP_DEEMPH_WC", wc)):
        if v is None:
            monkeypatch.delenv(k, raising=False)
        else:
            monkeypatch.setenv(k, str(v))
    monkeypatch.delenv("SDRHIP_DEEMPH_TILED", raising=False)

    def make(n):
        if data == "noise":
            return (3000 * np.sin(np.arange(n) * 0.01)[None, :] + rng.normal(0, 200, (C, n))).astype(np.int16)
        if data == "fullscale":
            return rng.integers(-32768, 32768, (C, n), dtype=np.int16)
        if data == "constant":
            return np.repeat(rng.integers(-32768, 32768, (C, 1), dtype=np.int16), n, axis=1)
        if data == "quiet":
            return rng.integers(-2, 3, (C, n)).astype(np.int16)
        x = np.repeat(rng.integers(-20000, 20000, (C, n // 700 + 1)), 700, axis=1)[:, :n]   # long plateaus, sudden steps
        return x.astype(np.int16)

    node = sa.FMDeemphI16(ctx, alpha, channels=C, max_in=ld)
    if P is None:
        assert node.kernel_names() == node.kernel_names(9001) == ["deemph_i16_spec_kernel"] and node.kernel_names(40) == ["deemph_i16_seq_kernel"]
    avgs = [np.zeros(1, np.int16) for _ in range(C)]
    xin = torch.zeros((C, ld), dtype=torch.int16, device=dev)
    xout_flat = torch.zeros(C * (ld + 5) + 8, dtype=torch.int16, device=dev)
    xout = xout_flat[3:3 + C * (ld + 5)].view(C, ld + 5)
    for n in (9001, 3276, 40, 8192, 1, 2100, 524, 9000):
        x = make(n)
        xin[:, :n] = torch.from_numpy(x).to(dev)
        torch.cuda.synchronize()
        node.process_dev(xin.data_ptr(), n, ld, xout.data_ptr(), ld + 5)
        ctx.synchronize()
        want = np.zeros_like(x)
        for c in range(C):
            o = np.zeros(n, np.int16)
            orc.lib().orc_fmdeemph_i16(orc._p(np.ascontiguousarray(x[c]), ctypes.c_int16), n, alpha, orc._p(avgs[c], ctypes.c_int16), orc._p(o, ctypes.c_int16))
            want[c] = o
        got = xout[:, :n].cpu().numpy()
        assert np.array_equal(got, want), (alpha, P, wc, data, n, np.argwhere(got != want)[:4])


@pytest.mark.parametrize("C", [1, 2, 300])
@pytest.mark.parametrize("data", ["noise", "constant"])
def test_fmdeemph_segmented_kernel_channel_counts(ctx, orc, C, data):
    """... one channel (the drop-in node's case: one workgroup, most of its lanes shadowing the last channel without
    stores), two, and more channels than one workgroup's 8 … 64; through the HOST entry point (staging buffers), rows of a
    call length that leaves a ragged last segment and a tail."""
    alpha, rng = 4, np.random.default_rng(C)
    node = sa.FMDeemphI16(ctx, alpha, channels=C, max_in=5003)
    assert node.kernel_names(5003) == ["deemph_i16_spec_kernel"]
    avgs = [np.zeros(1, np.int16) for _ in range(C)]
    for n in (5003, 4096, 700, 5000):
        x = (rng.normal(0, 500, (C, n)) if data == "noise" else np.repeat(rng.integers(-3000, 3000, (C, 1)), n, axis=1)).astype(np.int16)
        got = node.process(x)
        for c in range(C):
            o = np.zeros(n, np.int16)
            orc.lib().orc_fmdeemph_i16(orc._p(np.ascontiguousarray(x[c]), ctypes.c_int16), n, alpha, orc._p(avgs[c], ctypes.c_int16), orc._p(o, ctypes.c_int16))
            assert np.array_equal(got[c], o), (C, data, n, c)


@pytest.mark.parametrize("alpha", [1, 2, 4, 7, 100, 32767])
def test_fmdeemph_every_kernel_vs_oracle(ctx, orc, alpha, monkeypatch):
    """FMDeemph<int16_t>'s three kernels — the register-walking one, the copy (alpha = 1: the update is avg = x,
    src/demod.hh:342-351 with the alpha of a sample rate below about 11 kS/s) and the LDS-tiled one of rounds 1-2
    (SDRHIP_DEEMPH_TILED=1) — on device rows of every length and alignment class (odd row stride: every lane's row starts
    at its own offset from a 16-byte boundary, input and output rows at different ones), state carried from call to call,
    against the oracle's recursion with the same alpha."""
    dev = torch.device("cuda:0")
    C, ld = 70, 3289
    rng = np.random.default_rng(alpha)

    def ref_rows(x, avgs):
        out = np.zeros_like(x)
        for c in range(x.shape[0]):
            o = np.zeros(x.shape[1], np.int16)
            if x.shape[1]:
                orc.lib().orc_fmdeemph_i16(orc._p(np.ascontiguousarray(x[c]), ctypes.c_int16), x.shape[1], alpha, orc._p(avgs[c], ctypes.c_int16), orc._p(o, ctypes.c_int16))
            out[c] = o
        return out

    for tiled in (False, True):
        if tiled:
            monkeypatch.setenv("SDRHIP_DEEMPH_TILED", "1")
        else:
            monkeypatch.delenv("SDRHIP_DEEMPH_TILED", raising=False)
        node = sa.FMDeemphI16(ctx, alpha, channels=C, max_in=ld)
        avgs = [np.zeros(1, np.int16) for _ in range(C)]
        xin = torch.zeros((C, ld), dtype=torch.int16, device=dev)
        xout_flat = torch.zeros(C * (ld + 3) + 8, dtype=torch.int16, device=dev)
        xout = xout_flat[5:5 + C * (ld + 3)].view(C, ld + 3)   # (another stride and base offset than the input's)
        for n in (3276, 1, 7, 8, 524, 65, 0, 1000, 9, 16):
            x = rng.integers(-32768, 32768, (C, n), dtype=np.int16)
            xin[:, :n] = torch.from_numpy(x).to(dev)
            torch.cuda.synchronize()
            node.process_dev(xin.data_ptr(), n, ld, xout.data_ptr(), ld + 3)
            ctx.synchronize()
            assert np.array_equal(xout[:, :n].cpu().numpy(), ref_rows(x, avgs)), (alpha, tiled, n)
        xal = torch.zeros((C, 3280), dtype=torch.int16, device=dev)   # 16-byte aligned rows on both sides: 16-byte stores
        yal = torch.zeros((C, 3280), dtype=torch.int16, device=dev)
        for n in (3276, 3):
            x = rng.integers(-32768, 32768, (C, n), dtype=np.int16)
            xal[:, :n] = torch.from_numpy(x).to(dev)
            torch.cuda.synchronize()
            node.process_dev(xal.data_ptr(), n, 3280, yal.data_ptr(), 3280)
            ctx.synchronize()
            assert np.array_equal(yal[:, :n].cpu().numpy(), ref_rows(x, avgs)), (alpha, tiled, n, "aligned")


# ---- K1 at the BASELINE size: properties + sampled oracle comparison ------------------------------------

def test_iqbb_full_size_properties(ctx, orc, k1path):
    C, N, D = 1024, 65536, 8
    taps = sa.design_iqbb_taps(100e3, 50e3, FS, 127)
    lut = sa.design_freqshift_lut_i16()
    base = synth_channels(orc, 8, N)
    x = np.ascontiguousarray(base[np.arange(C) % 8])          # channel c carries pattern c % 8
    node = sa.IQBaseBandI16(ctx, taps, lut, 1365, False, D, channels=C, max_in=N, epilogue=sa.EPI_FM)
    y1 = node.process(x)
    y2 = node.process(x)
    assert y1.shape == (C, 8191) and y2.shape == (C, 8192)
    # invariance under channel batching: identical inputs -> identical outputs, wherever the channel sits
    for k in range(8):
        assert (y1[k::8] == y1[k]).all() and (y2[k::8] == y2[k]).all()
    # the 8 distinct patterns against the oracle, both calls (state carried)
    for k in range(8):
        bb, fm = orc.IQBaseBandI16(taps, lut, 1365, False, D), orc.FMDemodI16()
        assert np.array_equal(y1[k], fm.process(bb.process(x[k])))
        assert np.array_equal(y2[k], fm.process(bb.process(x[k])))
    # re-chunking invariance of the complex output (K1 is invariant, FM is not)
    n1 = sa.IQBaseBandI16(ctx, taps, lut, 1365, False, D, channels=C, max_in=N, epilogue=sa.EPI_NONE)
    n2 = sa.IQBaseBandI16(ctx, taps, lut, 1365, False, D, channels=C, max_in=N, epilogue=sa.EPI_NONE)
    whole = n1.process(x)
    parts = np.concatenate([n2.process(x[:, :30001]), n2.process(x[:, 30001:])], axis=1)
    assert np.array_equal(whole, parts)


@pytest.mark.parametrize("order,cu8", [(127, False), (16, False), (127, True)])
def test_timed_workload_every_channel_vs_oracle(ctx, orc, order, cu8):
    """bench.py's own workload — 1024 DISTINCT channels (its on-device generator, two tones + hashed noise per channel),
    65536 samples, two calls — with EVERY channel of both calls against the oracle (threads: the oracle library releases
    the GIL), for the headline plan, the reference's 16-tap plan and complex<uint8> input: what the timed kernel computes
    on the timed data is the reference's arithmetic, channel for channel (bench.py itself samples 32 channels per run)."""
    import concurrent.futures
    import bench
    C, N, D = 1024, 65536, 8
    dev = torch.device("cuda", 0)
    x = np.stack([bench.synth_cs16(torch, C, N, dev, 1234 + b).cpu().numpy() for b in range(2)])   # [call, C, N, 2]
    if cu8:
        x = (((x.astype(np.int32) >> 6) + 127).clip(0, 255)).astype(np.uint8)
    taps = sa.design_iqbb_taps(100e3, 50e3, FS, order)
    lut, inc = sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(100e3, FS)
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, False, D, channels=C, max_in=N, epilogue=sa.EPI_FM)
    if cu8:
        node.set_input_format(sa.abi.IN_CU8)
    assert node.kernel_names == ["iqbb_hot_kernel"]
    y = [node.process(x[0]), node.process(x[1])]

    def check(c):
        bb, fm = orc.IQBaseBandI16(taps, lut, inc, False, D), orc.FMDemodI16()
        for k in range(2):
            r = fm.process(bb.process(orc.autocast_cu8_cs16(x[k, c]) if cu8 else x[k, c]))
            if not np.array_equal(y[k][c], r):
                return c
        return -1
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        bad = [c for c in ex.map(check, range(C)) if c >= 0]
    assert not bad, bad[:10]


@pytest.mark.parametrize("epi,n_out", [(sa.EPI_FM, 630), (sa.EPI_FM, 693), (sa.EPI_NONE, 576), (sa.EPI_USB, 640), (sa.EPI_FM, 631)])
@pytest.mark.parametrize("tail", [0, 1, 5])
def test_iqbb_last_group_at_a_slice_end(ctx, orc, epi, n_out, tail, k1path):
    """The call's last emitted group carries state to the next call (FMDemod's last angle, the open window's carry). Call
    lengths that put it on the LAST group of a 64-group wave slice (FM: slices start at 252 t - 1 + 63 w) — the case in
    which round 2's hot kernel first took that slice for an interior one — and one beside it; `tail` extra samples leave
    a partial window open. The second call shows whether the state arrived."""
    taps, lut, inc = sa.design_iqbb_taps(100e3, 50e3, FS, 127), sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(100e3, FS)
    rng = np.random.default_rng(n_out + tail)
    lens = [8 * n_out + 1 + tail, 5000, 8 * n_out - tail, 4096]   # (the first window of a stream closes after D + 1 samples)
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, False, 8, channels=2, max_in=max(lens), epilogue=epi)
    refs = [orc.IQBaseBandI16(taps, lut, inc, False, 8) for _ in range(2)]
    fms = [orc.FMDemodI16() for _ in range(2)]
    for n in lens:
        x = rng.integers(-32768, 32768, (2, n, 2), dtype=np.int16)
        y = node.process(x)
        for c in range(2):
            r = refs[c].process(x[c])
            r = fms[c].process(r) if epi == sa.EPI_FM else orc.usb_i16(r) if epi == sa.EPI_USB else r
            assert np.array_equal(y[c], r), (n, c)


def test_iqbb_more_channels_than_workgroups(ctx, orc):
    """1 300 channels on a chip whose persistent grid has 1 024 workgroups: the hot ranges and the cold phase both wrap
    around the grid (channels bx, bx + gx). 13 distinct patterns tiled over the channels, two calls."""
    C, D = 1300, 8
    taps, lut, inc = sa.design_iqbb_taps(100e3, 50e3, FS, 127), sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(-100e3, FS)
    rng = np.random.default_rng(99)
    node = sa.IQBaseBandI16(ctx, taps, lut, inc, True, D, channels=C, max_in=9000, epilogue=sa.EPI_FM)
    refs = [(orc.IQBaseBandI16(taps, lut, inc, True, D), orc.FMDemodI16()) for _ in range(13)]
    for n in (9000, 6111):
        base = rng.integers(-32768, 32768, (13, n, 2), dtype=np.int16)
        y = node.process(np.ascontiguousarray(base[np.arange(C) % 13]))
        for k in range(13):
            r = refs[k][1].process(refs[k][0].process(base[k]))
            assert (y[k::13] == y[k]).all() and np.array_equal(y[k], r), (n, k)


# ---- K2: exact int16 FIR --------------------------------------------------------------------------------

@pytest.mark.parametrize("case,order,inp", [("g5_fir127", 127, "g1_iq_cs16"), ("g5_fir255", 255, "g1_iq_cs16"),
                                            ("g8_irregular_fir127", 127, "g1_iq_cs16"), ("g8_loud_fir127", 127, "g8_iq_cs16_loud")])
def test_fir_cs16_golden(ctx, golden, case, order, inp):
    m = golden.meta(case + "_out")
    node = sa.FIR(ctx, sa.FIR_CS16_EXACT, golden.load("g2_firlp_alpha%d" % order), max_in=4096)
    outs = [node.process(c)[0] for c in split(golden.load(inp), m["in_lens"])]
    assert np.array_equal(np.concatenate(outs), golden.load(case + "_out"))


@pytest.mark.parametrize("case,order", [("g5_fir127", 127), ("g5_fir255", 255), ("g8_irregular_fir127", 127)])
def test_fir_cs16_fm_golden(ctx, golden, case, order):
    m = golden.meta(case + "_fm")
    node = sa.FIR(ctx, sa.FIR_CS16_EXACT, golden.load("g2_firlp_alpha%d" % order), max_in=4096, epilogue=sa.EPI_FM)
    outs = [node.process(c)[0] for c in split(golden.load("g1_iq_cs16"), m["in_lens"])]
    outs = [o for o in outs if len(o)]
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load(case + "_fm"))


@pytest.mark.parametrize("order", [255, 16, 1])
def test_fir_cs16_batched_vs_oracle(ctx, orc, order):
    C, chunks = 4, [4096, 1000, 3, 4096]
    alpha = sa.design_fir_lowpass(order, 100e3, FS)
    x = synth_channels(orc, C, sum(chunks))
    rng = np.random.default_rng(3)
    x[3] = rng.integers(-32768, 32768, size=x[3].shape).astype(np.int16)   # full-range channel
    node = sa.FIR(ctx, sa.FIR_CS16_EXACT, alpha, channels=C, max_in=4096, epilogue=sa.EPI_FM)
    refs = [(orc.FIR(alpha), orc.FMDemodI16()) for _ in range(C)]
    off = 0
    for n in chunks:
        y = node.process(x[:, off:off + n])
        for c in range(C):
            f, fm = refs[c]
            assert np.array_equal(y[c], fm.process(f.process_cs16(x[c, off:off + n]))), (c, n)
        off += n


def test_fir_cs16_wrap_variant(ctx, orc):
    """sum|alpha| > 1 can push a partial sum past int16: the reference wraps per tap, so do we."""
    rng = np.random.default_rng(5)
    alpha = rng.uniform(-0.3, 0.3, 31)
    x = rng.integers(-32768, 32768, size=(2, 3000, 2)).astype(np.int16)
    node = sa.FIR(ctx, sa.FIR_CS16_EXACT, alpha, channels=2, max_in=3000)
    y = node.process(x)
    for c in range(2):
        assert np.array_equal(y[c], orc.FIR(alpha).process_cs16(x[c]))


# ---- K3: float FIR (+ folded SubSample, + float demods) --------------------------------------------------

@pytest.mark.parametrize("time_domain", [False, True])
def test_fir_cf32_golden(ctx, golden, time_domain, monkeypatch):
    """FIRLowPass<complex<float>>(127) against the reference's own output (g6). Without decimation the plan is an overlap-save
    FFT convolution on the tuned kernels behind the same handle (6x faster at 127 taps, 130x at 4097); `time_domain`: the
    time-domain kernel (SDRHIP_FIR_TIME_DOMAIN=1)."""
    if time_domain:
        monkeypatch.setenv("SDRHIP_FIR_TIME_DOMAIN", "1")
    else:
        monkeypatch.delenv("SDRHIP_FIR_TIME_DOMAIN", raising=False)
    x = golden.load("g1_iq_cf32")
    node = sa.FIR(ctx, sa.FIR_CF32, golden.load("g2_firlp_alpha127"), max_in=4096)
    assert node.kernel_names() == (["fir_cf32_rt_kernel"] if time_domain else ["fftconv_fused_kernel"])
    y = np.concatenate([node.process(x[i * 4096:(i + 1) * 4096])[0] for i in range(3)])
    assert rel_err(y, golden.load("g6_fir127_cf32_out")) <= RTOL


def test_fir_setfreq_midstream_golden(ctx, golden):
    """FIRLowPass::setFreq between buffers (FIRFilter::setUpperFreq, src/firfilter.hh:165-170,287): only the coefficients
    change, the ring goes on — sdrhip_fir_set_taps on the SAME plan against the reference's own output (g17), bit-exact for
    complex<int16>, <= 1e-5 for complex<float>."""
    a100, a40 = sa.design_fir_lowpass(127, 100e3, FS), sa.design_fir_lowpass(127, 40e3, FS)
    assert np.array_equal(a100, golden.load("g2_firlp_alpha127"))
    x = golden.load("g1_iq_cs16")
    node = sa.FIR(ctx, sa.FIR_CS16_EXACT, a100, max_in=4096)
    outs = []
    for b in range(4):
        if b == golden.meta("g17_fir127_setfreq_cs16")["switch_after_buffers"]:
            node.set_taps(a40)
        outs.append(node.process(x[b * 4096:(b + 1) * 4096])[0])
    assert np.array_equal(np.concatenate(outs), golden.load("g17_fir127_setfreq_cs16"))
    xf = golden.load("g1_iq_cf32")
    for td in ("0", "1"):   # (the overlap-save plan behind the handle, and the time-domain kernel)
        os.environ["SDRHIP_FIR_TIME_DOMAIN"] = td
        try:
            nodef = sa.FIR(ctx, sa.FIR_CF32, a100, max_in=4096)
        finally:
            del os.environ["SDRHIP_FIR_TIME_DOMAIN"]
        _setfreq_cf32(golden, nodef, xf, a40)


def _setfreq_cf32(golden, nodef, xf, a40):
    outs = []
    for b in range(3):
        if b == golden.meta("g17_fir127_setfreq_cf32")["switch_after_buffers"]:
            nodef.set_taps(a40)
        outs.append(nodef.process(xf[b * 4096:(b + 1) * 4096])[0])
    ref = golden.load("g17_fir127_setfreq_cf32")
    assert rel_err(np.concatenate(outs), ref) <= RTOL
    # right behind the switch (the old ring under the new taps), against the stream's scale
    assert np.abs(np.concatenate(outs)[4096:4096 + 200].astype(np.float64) - ref[4096:4096 + 200]).max() <= RTOL * np.abs(ref).max()


def test_float_baseband_setters_keep_the_stream(ctx):
    """IQBaseBand<float> setters on the SAME plan (same order, decimation, buffer size): setFilterWidth = new low-pass
    coefficients over the kept history; setCenterFrequency = the phasor restarts at the current sample, history and
    decimator go on (src/baseband.hh:82-101 semantics) — against the float64 closed form of exactly that definition."""
    D, order, N = 8, 127, 4096
    rng = np.random.default_rng(77)
    x = (rng.standard_normal((2, 4 * N, 2)) * 0.3).astype(np.float32)
    a1, a2 = sa.design_fir_lowpass(order, 25e3, FS), sa.design_fir_lowpass(order, 60e3, FS)
    node = sa.FloatBaseBand(ctx, 100e3, FS, a1, D, channels=2, max_in=N)
    ys = []
    for b in range(4):
        if b == 1:
            node.set_taps(a2)
        if b == 2:
            node.set_shift(-250e3)
        ys.append(node.process(x[:, b * N:(b + 1) * N]))
    y = np.concatenate(ys, axis=1)
    n = np.arange(4 * N, dtype=np.float64)
    for c in range(2):
        xc = x[c, :, 0].astype(np.float64) + 1j * x[c, :, 1]
        ref = []
        # output j averages the FIR outputs at n = j*D .. j*D + D-1; a FIR output at n uses x[n - k] * phasor(n - k), k < order
        for seg, (alpha, fc, n_sw) in enumerate([(a1, 100e3, 0), (a2, 100e3, 0), (a2, -250e3, 2 * N), (a2, -250e3, 2 * N)]):
            sh = xc * np.exp(-2j * np.pi * np.fmod(fc * (n - n_sw) / FS, 1.0))
            f = np.convolve(sh, alpha[::-1])[:4 * N]   # alpha[order-1] multiplies the newest sample (src/firfilter.hh:237-243)
            box = f.reshape(-1, D).mean(axis=1)
            ref.append(box[seg * N // D:(seg + 1) * N // D])
        ref = np.concatenate(ref)
        yc = y[c, :, 0].astype(np.float64) + 1j * y[c, :, 1]
        assert yc.shape == ref.shape
        assert np.abs(yc - ref).max() / np.abs(ref).max() <= RTOL, c


@pytest.mark.parametrize("n", [8, 3])
def test_fir_cf32_decimated_golden(ctx, golden, n):
    x = golden.load("g1_iq_cf32")
    node = sa.FIR(ctx, sa.FIR_CF32, golden.load("g2_firlp_alpha127"), decim=n, max_in=4096)
    outs = [node.process(x[i * 4096:(i + 1) * 4096])[0] for i in range(3)]
    assert [len(o) for o in outs] == golden.meta("g6_fir127_cf32_sub%d" % n)["out_lens"]
    assert rel_err(np.concatenate(outs), golden.load("g6_fir127_cf32_sub%d" % n)) <= RTOL


@pytest.mark.parametrize("demod", ["am", "usb"])
def test_fir_cf32_demod_golden(ctx, golden, demod):
    x = golden.load("g1_iq_cf32")
    node = sa.FIR(ctx, sa.FIR_CF32, golden.load("g2_firlp_alpha127"), max_in=4096,
                  epilogue=sa.EPI_AM if demod == "am" else sa.EPI_USB)
    y = np.concatenate([node.process(x[i * 4096:(i + 1) * 4096])[0] for i in range(3)])
    assert rel_err(y, golden.load("g6_fir127_cf32_" + demod)) <= RTOL


@pytest.mark.parametrize("time_domain", [False, True])
def test_fir_cf32_4097_golden(ctx, golden, time_domain, monkeypatch):
    if time_domain:
        monkeypatch.setenv("SDRHIP_FIR_TIME_DOMAIN", "1")
    else:
        monkeypatch.delenv("SDRHIP_FIR_TIME_DOMAIN", raising=False)
    x = golden.load("g1_iq_cf32")
    node = sa.FIR(ctx, sa.FIR_CF32, golden.load("g2_firlp_alpha4097"), max_in=4096)
    y = np.concatenate([node.process(x[i * 4096:(i + 1) * 4096])[0] for i in range(3)])
    assert rel_err(y, golden.load("g6_fir4097_cf32_out")) <= RTOL


# ---- K4/K5/K6 stand-alone -------------------------------------------------------------------------------

def test_demods_standalone_golden(ctx, golden):
    x = golden.load("g1_iq_cs16")
    for kind, name in ((sa.EPI_AM, "g4_raw_am"), (sa.EPI_USB, "g4_raw_usb")):
        node = sa.Demod(ctx, kind, sa.T_CS16, max_in=4096)
        y = np.concatenate([node.process(x[i * 4096:(i + 1) * 4096])[0] for i in range(4)])
        assert np.array_equal(y, golden.load(name))
    node = sa.Demod(ctx, sa.EPI_FM, sa.T_CS16, max_in=4096, inplace_fm0=False)
    y = np.concatenate([node.process(x[i * 4096:(i + 1) * 4096])[0] for i in range(4)])   # out[0] stays the caller's 0
    assert np.array_equal(y, golden.load("g4_raw_fm_masked0"))
    xf = golden.load("g6_fir127_cf32_out")
    for kind, name in ((sa.EPI_AM, "g6_fir127_cf32_am"), (sa.EPI_USB, "g6_fir127_cf32_usb")):
        node = sa.Demod(ctx, kind, sa.T_CF32, max_in=3 * 4096)
        assert np.array_equal(node.process(xf)[0], golden.load(name))


def test_demod_fm_odd_sizes_vs_oracle(ctx, orc):
    rng = np.random.default_rng(11)
    x = rng.integers(-32768, 32768, size=(3, 1237, 2)).astype(np.int16)
    node = sa.Demod(ctx, sa.EPI_FM, sa.T_CS16, channels=3, max_in=2000, inplace_fm0=True)
    fms = [orc.FMDemodI16() for _ in range(3)]
    for lo, hi in ((0, 1), (1, 2), (2, 700), (700, 1237)):
        y = node.process(x[:, lo:hi])
        for c in range(3):
            assert np.array_equal(y[c], fms[c].process(x[c, lo:hi]))


@pytest.mark.parametrize("inplace_fm0", [True, False])
def test_fm_demod_long_rows_state_carried(ctx, orc, inplace_fm0):
    """Stand-alone FMDemod<int16> on rows long enough for the 8- and 16-samples-per-lane forms (>= 2048 / 4096 samples),
    mixed with short calls: the last angle travels from call to call whatever form ran."""
    rng = np.random.default_rng(12)
    C = 3
    node = sa.Demod(ctx, sa.EPI_FM, sa.T_CS16, channels=C, max_in=70000, inplace_fm0=inplace_fm0)
    fms = [orc.FMDemodI16() for _ in range(C)]
    for n in [5000, 2048, 3000, 4096, 100, 9001, 1, 65536, 2047, 4095, 70000]:
        x = rng.integers(-32768, 32768, (C, n, 2), dtype=np.int16)
        y = node.process(x)
        for c in range(C):
            r = fms[c].process(x[c])
            lo = 0 if inplace_fm0 else 1            # out[0] is never written by FMDemod: only the in-place convention defines it
            assert np.array_equal(y[c][lo:], r[lo:]), (n, c)


def test_subsample8_fast_and_general_calls_vs_oracle(ctx, orc):
    """SubSample<cs16>(8): calls that start on a group boundary and hold whole groups take the coalesced kernel, the others
    the general one; full-scale samples, state carried across both kinds of call, 3 channels."""
    rng = np.random.default_rng(77)
    C, chunks = 3, [4096, 13, 8000, 3, 4096, 8, 65536, 5, 3, 16]
    node = sa.SubSample(ctx, sa.T_CS16, 8, channels=C, max_in=65536)
    refs = [orc.SubSample(8) for _ in range(C)]
    for n in chunks:
        x = rng.integers(-32768, 32768, (C, n, 2), dtype=np.int16)
        y = node.process(x)
        for c in range(C):
            assert np.array_equal(y[c], refs[c].process_cs16(x[c]))


@pytest.mark.parametrize("n", [8, 3])
def test_subsample_golden(ctx, golden, n):
    x = golden.load("g1_iq_cs16")
    node = sa.SubSample(ctx, sa.T_CS16, n, max_in=4096)
    outs = [node.process(x[i * 4096:(i + 1) * 4096])[0] for i in range(4)]
    assert [len(o) for o in outs] == golden.meta("g6_subsample_cs16_n%d" % n)["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load("g6_subsample_cs16_n%d" % n))
    xf = golden.load("g6_fir127_cf32_out")
    node = sa.SubSample(ctx, sa.T_CF32, n, max_in=4096)
    outs = [node.process(xf[i * 4096:(i + 1) * 4096])[0] for i in range(3)]
    assert np.array_equal(np.concatenate(outs), golden.load("g6_fir127_cf32_sub%d" % n))   # same op order: exact


# ---- K7: FFT and FFT convolution ---------------------------------------------------------------------------

@pytest.mark.parametrize("n", [16, 64, 2048, 8192, 16384])
def test_fft_vs_numpy(ctx, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((3, n, 2)).astype(np.float32)
    xc = x[..., 0].astype(np.float64) + 1j * x[..., 1]
    for sign, ref in ((-1, np.fft.fft(xc, axis=1)), (+1, np.fft.ifft(xc, axis=1) * n)):
        y = sa.fft_c2c(ctx, x, sign)
        yc = y[..., 0].astype(np.float64) + 1j * y[..., 1]
        assert np.abs(yc - ref).max() / np.abs(ref).max() < 2e-6, (n, sign)


@pytest.mark.parametrize("n", [2, 4, 64, 1024, 8192])
def test_fft_double_vs_numpy(ctx, n):
    """FFTPlan<double> (reference src/fftplan_fftw3.hh:12-76; FFTW is not in /root/reference: held to numpy's double FFT)."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal((3, n, 2))
    xc = x[..., 0] + 1j * x[..., 1]
    for sign, ref in ((-1, np.fft.fft(xc, axis=1)), (+1, np.fft.ifft(xc, axis=1) * n)):
        y = sa.fft_c2c_f64(ctx, x, sign)
        assert np.abs(y[..., 0] + 1j * y[..., 1] - ref).max() / np.abs(ref).max() < 1e-13, (n, sign)


@pytest.mark.parametrize("dt,tol", [(np.complex64, 2e-6), (np.complex128, 1e-13)])
def test_fft_exec_host_buffers(ctx, dt, tol):
    """FFT::exec on host buffers, forward then backward = n * x."""
    rng = np.random.default_rng(9)
    x = (rng.standard_normal(4096) + 1j * rng.standard_normal(4096)).astype(dt)
    X = sa.fft_exec(ctx, x, -1)
    assert np.abs(X - np.fft.fft(x.astype(np.complex128))).max() / np.abs(X).max() < tol
    back = sa.fft_exec(ctx, X, +1)
    assert np.abs(back / 4096 - x).max() < tol * 10
    y = sa.fft_exec(ctx, x[:1003], -1)   # 1003 = 17 x 59: a prime factor above 13 — the chirp transform
    ref = np.fft.fft(x[:1003].astype(np.complex128))
    assert np.abs(y - ref).max() / np.abs(ref).max() < tol * 5


@pytest.mark.parametrize("n", [1, 2, 3, 5, 6, 12, 100, 1000, 1001, 2000, 3000, 5000, 6006, 15000, 16380])
def test_fft_any_size_vs_numpy(ctx, n):
    """FFTPlan<float> / FFT::exec plan ANY size in the reference (fftw_plan_dft_1d(in.size(), ...), src/fftplan_fftw3.hh:34-36):
    sizes that are not powers of two run the general mixed-radix plan (factors 2 ... 13; csrc/fftgen.hpp) — held to numpy's
    double FFT like the power-of-two plans (FFTW is not in /root/reference)."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal((3, n, 2)).astype(np.float32)
    xc = x[..., 0].astype(np.float64) + 1j * x[..., 1]
    for sign, ref in ((-1, np.fft.fft(xc, axis=1)), (+1, np.fft.ifft(xc, axis=1) * n)):
        y = sa.fft_c2c(ctx, x, sign)
        yc = y[..., 0].astype(np.float64) + 1j * y[..., 1]
        assert np.abs(yc - ref).max() / np.abs(ref).max() < 3e-6, (n, sign)


@pytest.mark.parametrize("n", [1, 3, 5, 7, 30, 1000, 2000, 2002, 3000, 5000, 6006, 8190])
def test_fft_double_any_size_vs_numpy(ctx, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((3, n, 2))
    xc = x[..., 0] + 1j * x[..., 1]
    for sign, ref in ((-1, np.fft.fft(xc, axis=1)), (+1, np.fft.ifft(xc, axis=1) * n)):
        y = sa.fft_c2c_f64(ctx, x, sign)
        assert np.abs(y[..., 0] + 1j * y[..., 1] - ref).max() / np.abs(ref).max() < 1e-13, (n, sign)


@pytest.mark.parametrize("n,dtype,tol", [(17, np.float32, 3e-6), (34, np.float32, 3e-6), (97, np.float32, 3e-6), (1003, np.float32, 5e-6),
                                         (4099, np.float32, 1e-5), (8191, np.float32, 1e-5), (19, np.float64, 1e-13), (2053, np.float64, 1e-12),
                                         (4093, np.float64, 1e-12)])
def test_fft_prime_sizes_vs_numpy(ctx, n, dtype, tol):
    """Sizes with a prime factor above 13 (FFTW plans them like any other): Bluestein's chirp transform over the next power
    of two >= 2n - 1, evaluated with the general in-LDS plan — both directions against numpy's double FFT."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal((2, n, 2)).astype(dtype)
    xc = x[..., 0].astype(np.float64) + 1j * x[..., 1]
    fn = sa.fft_c2c_f64 if np.dtype(dtype) == np.float64 else sa.fft_c2c
    for sign, ref in ((-1, np.fft.fft(xc, axis=1)), (+1, np.fft.ifft(xc, axis=1) * n)):
        y = fn(ctx, x, sign)
        yc = y[..., 0].astype(np.float64) + 1j * y[..., 1]
        assert np.abs(yc - ref).max() / np.abs(ref).max() < tol, (n, sign)


@pytest.mark.parametrize("n,dtype,tol", [(32768, np.float32, 3e-6), (65536, np.float32, 3e-6), (100000, np.float32, 5e-6), (3 * 16384, np.float32, 3e-6),
                                         (1 << 20, np.float32, 5e-6), (16384, np.float64, 1e-13), (10000, np.float64, 1e-13), (200000, np.float64, 1e-12)])
def test_fft_long_sizes_vs_numpy(ctx, n, dtype, tol):
    """Transforms longer than one workgroup's LDS holds (FFTW has no such limit): the four-step plan n = n1 x n2 — n2
    column transforms with the twiddle, n1 row transforms, both through the general in-LDS passes — against numpy, both
    directions."""
    rng = np.random.default_rng(n % 1000)
    x = rng.standard_normal((2, n, 2)).astype(dtype)
    xc = x[..., 0].astype(np.float64) + 1j * x[..., 1]
    fn = sa.fft_c2c_f64 if np.dtype(dtype) == np.float64 else sa.fft_c2c
    for sign, ref in ((-1, np.fft.fft(xc, axis=1)), (+1, np.fft.ifft(xc, axis=1) * n)):
        y = fn(ctx, x, sign)
        yc = y[..., 0].astype(np.float64) + 1j * y[..., 1]
        assert np.abs(yc - ref).max() / np.abs(ref).max() < tol, (n, sign)


@pytest.mark.parametrize("n,dtype,tol,form", [(2 * 16411, np.float32, 1e-5, "chirp over four-step"), (8209, np.float32, 1e-5, "chirp over four-step"),
                                              (2 * 8209, np.float64, 1e-12, "chirp over four-step"), (4099, np.float64, 1e-12, "chirp over four-step"),
                                              (20014, np.float32, 1e-5, "chirp over four-step"), (2018, np.float32, 5e-6, "chirp"),
                                              (24000, np.float32, 3e-6, "four-step"), (32768, np.float32, 3e-6, "four-step"),
                                              (16384, np.float64, 1e-13, "four-step"), (16384, np.float32, 3e-6, "radix-16 lds"),
                                              (6000, np.float32, 3e-6, "lds"), (4096, np.float64, 1e-13, "radix-2 lds (double)")])
def test_fft_plan_any_size_planned_once(ctx, n, dtype, tol, form):
    """FFTPlan<Scalar> as the reference builds it (src/fftplan_fftw3.hh:34-36,59,64): planned ONCE — for any size, including
    those round 4 refused (a large prime factor in a long transform, a chirp transform beyond one workgroup's LDS) — then
    executed several times, both directions, batched and on host buffers, against numpy's double FFT."""
    cdt = np.complex128 if np.dtype(dtype) == np.float64 else np.complex64
    plan = sa.FFTPlan(ctx, n, cdt)
    assert plan.form == form, plan.form
    rng = np.random.default_rng(n % 977)
    for rep in range(2):   # (the same plan again: its tables and scratch are reused)
        x = (rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))).astype(cdt)
        xd = x.astype(np.complex128)
        for sign, ref in ((-1, np.fft.fft(xd, axis=1)), (+1, np.fft.ifft(xd, axis=1) * n)):
            y = plan.exec_batch(x, sign)
            assert np.abs(y - ref).max() / np.abs(ref).max() < tol, (n, sign, rep)
        y1 = plan.exec(x[1], -1)
        assert np.abs(y1 - np.fft.fft(xd[1])).max() / np.abs(np.fft.fft(xd[1])).max() < tol
    # the one-shot entry point serves the same sizes from the context's plan cache
    x2 = rng.standard_normal((2, n, 2)).astype(dtype)
    fn = sa.fft_c2c_f64 if np.dtype(dtype) == np.float64 else sa.fft_c2c
    y2 = fn(ctx, x2, -1)
    ref2 = np.fft.fft(x2[..., 0].astype(np.float64) + 1j * x2[..., 1], axis=1)
    assert np.abs(y2[..., 0].astype(np.float64) + 1j * y2[..., 1] - ref2).max() / np.abs(ref2).max() < tol
    assert np.array_equal(fn(ctx, x2, -1), y2)   # (second call: the cached plan)


@pytest.mark.parametrize("big,small,dtype,tol", [(15360, 10240, np.float32, 3e-6), (12288, 9216, np.float32, 3e-6), (7680, 5120, np.float64, 1e-13),
                                                  (2048 * 24, 1024 * 24, np.float32, 3e-6)])
def test_fft_plans_of_one_kernel_keep_their_lds(ctx, big, small, dtype, tol):
    """Two live plans on ONE kernel instance whose LDS needs differ and both exceed 64 KB (in-LDS transforms of 15360 / 10240
    points = 120 / 80 KB; four-step passes of 2048 / 1024 columns): the kernel's dynamic-LDS limit belongs to the (kernel,
    device) pair, so building the smaller plan must not lower it under the larger one — the larger plan
    executes again AFTER the smaller one was built and run, and once more from the context's plan cache."""
    cdt = np.complex128 if np.dtype(dtype) == np.float64 else np.complex64
    rng = np.random.default_rng(big)

    def check(plan, n):
        x = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).astype(cdt)
        ref = np.fft.fft(x.astype(np.complex128), axis=1)
        assert np.abs(plan.exec_batch(x, -1) - ref).max() / np.abs(ref).max() < tol, n

    pb = sa.FFTPlan(ctx, big, cdt)
    check(pb, big)
    ps = sa.FFTPlan(ctx, small, cdt)
    check(ps, small)
    check(pb, big)
    for n in (big, small, big):   # the one-shot entry point: plans out of the context's cache, in the order that used to lower the cap
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(cdt)
        y, ref = sa.fft_exec(ctx, x, -1), np.fft.fft(x.astype(np.complex128))
        assert np.abs(y - ref).max() / np.abs(ref).max() < tol, n


def test_fft_bad_arguments(ctx):
    """sizes below 1 are E_INVALID before anything is allocated (a negative n used to become a huge allocation)"""
    for n in (0, -5):
        with pytest.raises(sa.abi.SdrHipError) as e:
            sa.abi.check(sa.abi.lib().sdrhip_fft_exec(ctx.handle, sa.abi.T_CF32, n, -1, np.zeros(4, np.float32).ctypes.data_as(ctypes.c_void_p),
                                                      np.zeros(4, np.float32).ctypes.data_as(ctypes.c_void_p)))
        assert e.value.code == sa.abi.E_INVALID
    with pytest.raises(sa.abi.SdrHipError) as e:
        sa.FFTPlan(ctx, 0)
    assert e.value.code == sa.abi.E_INVALID


@pytest.mark.parametrize("N,dtype,tol,nblk", [(16384, np.float32, RTOL, 3), (12000, np.float32, RTOL, 3), (1009, np.float32, RTOL, 5),
                                              (8192, np.float64, 1e-11, 3), (10007, np.float32, RTOL, 2), (17, np.float32, RTOL, 9),
                                              (6007, np.float64, 1e-11, 2)])
@pytest.mark.parametrize("literal", [False, True])
def test_filternode_any_block_size_beyond_one_workgroup(ctx, orc, N, dtype, tol, nblk, literal, monkeypatch):
    """FilterNode<float>(16384), (12000), (1009), FilterNode<double>(8192) ... (src/filternode.hh:236-245: any block size,
    FFTW plans any 2N): transforms beyond one workgroup's LDS (four-step) and sizes with a prime factor above 13 (chirp
    transform, in LDS or over a four-step plan) — a 2-band bank, 3 channels, two ragged calls, against the oracle's
    FilterSink / FilterSource blocks AND the closed form y = h (*) x / (sqrt(2N) ||h||_2). `literal`: the 2N-point transform
    itself (four-step / chirp plans); otherwise overlap-save on the best power of two where N leaves one a quarter of its points
    (12000, 1009, 10007, 17 in float; 6007 in double), the 2N-point plans where it does not (16384; 8192 in double)."""
    from scipy.signal import fftconvolve
    if literal:
        monkeypatch.setenv("SDRHIP_FFTCONV_LITERAL", "1")
    else:
        monkeypatch.delenv("SDRHIP_FFTCONV_LITERAL", raising=False)
    f64 = np.dtype(dtype) == np.float64
    bands = [(-350e3, -250e3), (50e3, 150e3)]
    hs = [sa.design_fftfilt_kernel(N, lo, hi, FS, dtype=dtype) for lo, hi in bands]
    Ks = [sa.design_fftfilt_spectrum(h) for h in hs]
    rng = np.random.default_rng(N)
    x = (rng.standard_normal((3, nblk * N, 2)) * 0.3).astype(dtype)
    bank = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * N, Ks, channels=3, max_in=nblk * N, dtype=dtype)
    cut = N + N // 3   # (not a multiple of the block)
    y = np.concatenate([bank.process(x[:, :cut]), bank.process(x[:, cut:])], axis=2)
    assert y.shape == (2, 3, nblk * N, 2) and y.dtype == np.dtype(dtype)
    for b, h in enumerate(hs):
        hc = h[:, 0].astype(np.float64) + 1j * h[:, 1]
        for c in range(3):
            xc = x[c, :, 0].astype(np.float64) + 1j * x[c, :, 1]
            closed = fftconvolve(xc, hc)[:len(xc)] / (np.sqrt(2 * N) * np.sqrt((np.abs(hc) ** 2).sum()))
            yc = y[b, c, :, 0].astype(np.float64) + 1j * y[b, c, :, 1]
            assert np.abs(yc - closed).max() / np.abs(closed).max() <= tol, (N, b, c)
    flt = orc.FFTFilterF64(orc.fftfilt_design_K_f64(hs[1])) if f64 else orc.FFTFilter(orc.fftfilt_design_K(hs[1]))
    ref = np.concatenate([flt.process(x[2, i * N:(i + 1) * N]) for i in range(nblk)])
    assert rel_err(y[1, 2], ref) <= tol, N
    # the device-pointer entry point, reset, and a kernel swap between calls
    bank.reset()
    assert np.array_equal(bank.process(x[:, :cut]), y[:, :, :cut])
    K2 = sa.design_fftfilt_spectrum(sa.design_fftfilt_kernel(N, 100e3, 300e3, FS, dtype=dtype))
    bank.set_kernel(0, K2)
    fresh = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * N, [K2, Ks[1]], channels=3, max_in=nblk * N, dtype=dtype)
    fresh.process(x[:, :cut])
    assert np.array_equal(bank.process(x[:, cut:]), fresh.process(x[:, cut:]))


def test_filternode_big_block_many_channels(ctx):
    """FilterNode<float>(16384) on 600 channels: more than one pass of the scratch-bounded channel groups (512 channels per
    pass at this size), rows of 8 distinct streams tiled over the channels — every copy equal, the distinct ones against the
    closed form y = h (*) x / (sqrt(2N) ||h||_2)."""
    from scipy.signal import fftconvolve
    N, C, nblk = 16384, 600, 2
    h = sa.design_fftfilt_kernel(N, 50e3, 150e3, FS)
    K = sa.design_fftfilt_spectrum(h)
    rng = np.random.default_rng(99)
    base = (rng.standard_normal((8, nblk * N, 2)) * 0.3).astype(np.float32)
    x = np.ascontiguousarray(base[np.arange(C) % 8])
    node = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * N, K, channels=C, max_in=nblk * N)
    y = np.concatenate([node.process(x[:, :N + 5]), node.process(x[:, N + 5:])], axis=1)
    hc = h[:, 0].astype(np.float64) + 1j * h[:, 1]
    for k in range(8):
        assert np.array_equal(y[k::8], np.broadcast_to(y[k], y[k::8].shape)), k
        xc = base[k, :, 0].astype(np.float64) + 1j * base[k, :, 1]
        closed = fftconvolve(xc, hc)[:len(xc)] / (np.sqrt(2 * N) * np.sqrt((np.abs(hc) ** 2).sum()))
        yc = y[k, :, 0].astype(np.float64) + 1j * y[k, :, 1]
        assert np.abs(yc - closed).max() / np.abs(closed).max() <= RTOL, k


def test_fftconv_ols_long_transforms(ctx):
    """overlap-save with time-domain taps on transforms beyond the LDS: 65536 points / 30001 taps (float, four-step) and
    20014 points / 5000 taps (float, chirp over four-step) against a direct convolution in double"""
    from scipy.signal import fftconvolve
    rng = np.random.default_rng(11)
    for L, M in ((65536, 30001), (20014, 5000)):
        taps = (rng.standard_normal((M, 2)) / M).astype(np.float32)
        x = rng.standard_normal((2, 90000, 2)).astype(np.float32)
        node = sa.FFTConv(ctx, sa.FFTCONV_OLS, L, taps, channels=2, max_in=50000)
        y = np.concatenate([node.process(x[:, :41111]), node.process(x[:, 41111:])], axis=1)
        tc = taps[:, 0].astype(np.float64) + 1j * taps[:, 1]
        for c in range(2):
            xc = x[c, :, 0].astype(np.float64) + 1j * x[c, :, 1]
            ref = fftconvolve(xc, tc)[:90000]
            yc = y[c, :, 0].astype(np.float64) + 1j * y[c, :, 1]
            assert np.abs(yc - ref).max() / np.abs(ref).max() <= RTOL, (L, c)


@pytest.mark.parametrize("N,dtype,tol", [(1000, np.float32, RTOL), (1500, np.float32, RTOL), (1000, np.float64, 1e-12), (1024, np.float64, 1e-12),
                                         (7, np.float32, RTOL)])
@pytest.mark.parametrize("literal", [False, True])
def test_fftconv_any_block_size_and_double(ctx, golden, orc, N, dtype, tol, literal, monkeypatch):
    """FilterNode(size_t block_size) for block sizes that are not powers of two, and FilterNode<double>
    (src/filternode.hh:230-245): a 3-band bank behind one plan, two calls (history carried), ragged call lengths —
    against the oracle's FilterSink / FilterSource blocks AND the closed form y = h (*) x / (sqrt(2N) ||h||_2).
    By default such a block size runs as overlap-save with the same N taps on the power-of-two transform that costs least
    (the tuned kernels in float); `literal`: the 2N-point transform itself (SDRHIP_FFTCONV_LITERAL=1: the general plans)."""
    if literal:
        monkeypatch.setenv("SDRHIP_FFTCONV_LITERAL", "1")
    else:
        monkeypatch.delenv("SDRHIP_FFTCONV_LITERAL", raising=False)
    f64 = np.dtype(dtype) == np.float64
    bands = [(-350e3, -250e3), (50e3, 150e3), (-20e3, 20e3)]
    hs = [sa.design_fftfilt_kernel(N, lo, hi, FS, dtype=dtype) for lo, hi in bands]
    if N == 1000:
        assert np.array_equal(hs[0].ravel(), golden.load("g15_fftfilt_h1000_f64") if f64 else golden.load("g15_fftfilt_h1000").ravel())
    Ks = [sa.design_fftfilt_spectrum(h) for h in hs]
    nblk = 5
    rng = np.random.default_rng(N)
    x = (rng.standard_normal((2, nblk * N, 2)) * 0.3).astype(dtype)
    bank = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * N, Ks, channels=2, max_in=nblk * N, dtype=dtype)
    cut = 2 * N + N // 3   # (not a multiple of the block: the overlap-save evaluation takes any call length)
    y = np.concatenate([bank.process(x[:, :cut]), bank.process(x[:, cut:])], axis=2)
    assert y.shape == (3, 2, nblk * N, 2) and y.dtype == np.dtype(dtype)
    for b, (h, K) in enumerate(zip(hs, Ks)):
        flt = orc.FFTFilterF64(orc.fftfilt_design_K_f64(h)) if f64 else orc.FFTFilter(orc.fftfilt_design_K(h))
        for c in range(2):
            ref = np.concatenate([flt.process(x[c, i * N:(i + 1) * N]) for i in range(nblk)]) if c == 1 else None
            yc = y[b, c, :, 0].astype(np.float64) + 1j * y[b, c, :, 1]
            if ref is not None:
                assert rel_err(y[b, c], ref) <= tol, (N, b, c)
            hc = h[:, 0].astype(np.float64) + 1j * h[:, 1]
            xc = x[c, :, 0].astype(np.float64) + 1j * x[c, :, 1]
            closed = np.convolve(xc, hc)[:len(xc)] / (np.sqrt(2 * N) * np.sqrt((np.abs(hc) ** 2).sum()))
            assert np.abs(yc - closed).max() / np.abs(closed).max() <= tol, (N, b, c)
    # one band retuned between calls (FilterSource::setFreq): equals a plan made with the new kernel from the second call on
    K2 = sa.design_fftfilt_spectrum(sa.design_fftfilt_kernel(N, 100e3, 300e3, FS, dtype=dtype))
    b2 = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * N, Ks, channels=2, max_in=nblk * N, dtype=dtype)
    b2.process(x[:, :2 * N]); b2.set_kernel(1, K2)
    fresh = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * N, [Ks[0], K2, Ks[2]], channels=2, max_in=nblk * N, dtype=dtype)
    fresh.process(x[:, :2 * N])
    assert np.array_equal(b2.process(x[:, 2 * N:]), fresh.process(x[:, 2 * N:]))


def test_fftconv_ols_taps_any_fft_size(ctx, orc):
    """overlap-save with time-domain taps on FFT sizes that are not powers of two (float: 3000 points / 701 taps; double:
    2000 / 301) against a direct convolution"""
    rng = np.random.default_rng(5)
    for L, M, dtype, tol in ((3000, 701, np.float32, RTOL), (2000, 301, np.float64, 1e-12)):
        taps = (rng.standard_normal((M, 2)) / M).astype(dtype)
        x = rng.standard_normal((2, 7000, 2)).astype(dtype)
        node = sa.FFTConv(ctx, sa.FFTCONV_OLS, L, taps, channels=2, max_in=4000, dtype=dtype)
        y = np.concatenate([node.process(x[:, :3111]), node.process(x[:, 3111:])], axis=1)
        tc = taps[:, 0].astype(np.float64) + 1j * taps[:, 1]
        for c in range(2):
            xc = x[c, :, 0].astype(np.float64) + 1j * x[c, :, 1]
            ref = np.convolve(xc, tc)[:7000]
            yc = y[c, :, 0].astype(np.float64) + 1j * y[c, :, 1]
            assert np.abs(yc - ref).max() / np.abs(ref).max() <= tol, (L, c)


@pytest.mark.parametrize("N", [1024, 8192])
def test_fftconv_reference_mode_vs_oracle(ctx, golden, orc, N):
    """FilterSink+FilterSource (overlap-add, 2N-point FFT, N taps) — oracle is FFTW-unpinned; both are
    also held to the closed form y = h (*) x / (sqrt(2N) ||h||)."""
    h = golden.load("g7_fftfilt_h%d" % N)
    K = sa.design_fftfilt_spectrum(h)
    nblk = 12288 // N if N <= 4096 else 1
    x = golden.load("g1_iq_cf32")[:nblk * N]
    node = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * N, K, max_in=nblk * N)
    y = node.process(x)[0]
    flt = orc.FFTFilter(orc.fftfilt_design_K(h))
    ref = np.concatenate([flt.process(x[i * N:(i + 1) * N]) for i in range(nblk)])
    assert rel_err(y, ref) <= RTOL
    hc = h[:, 0].astype(np.float64) + 1j * h[:, 1]
    xc = x[:, 0].astype(np.float64) + 1j * x[:, 1]
    closed = np.convolve(xc, hc)[:len(xc)] / (np.sqrt(2 * N) * np.sqrt((np.abs(hc) ** 2).sum()))
    yc = y[:, 0].astype(np.float64) + 1j * y[:, 1]
    assert np.abs(yc - closed).max() / np.abs(closed).max() <= RTOL


@pytest.mark.parametrize("N", [1024, 8192])
def test_fftconv_filter_bank(ctx, golden, orc, N):
    """FilterNode's shape: several band kernels behind ONE forward transform per block (src/filternode.hh:81-88,257-270).
    The bank's rows equal the single-band plans' bit for bit and the oracle within 1e-5; N = 8192 (FFT 16384) does not fit
    the LDS twice and takes the per-band launches."""
    bands = [(50e3, 150e3), (-400e3, -250e3), (-20e3, 20e3)]
    Ks = [sa.design_fftfilt_spectrum(sa.design_fftfilt_kernel(N, lo, hi, FS)) for lo, hi in bands]
    nblk = 4 if N == 1024 else 2
    rng = np.random.default_rng(31)
    x = (rng.standard_normal((2, nblk * N, 2)) * 0.3).astype(np.float32)
    bank = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * N, Ks, channels=2, max_in=nblk * N)
    y = np.concatenate([bank.process(x[:, :N]), bank.process(x[:, N:])], axis=2)     # two calls: history carried
    assert y.shape == (3, 2, nblk * N, 2)
    for b, K in enumerate(Ks):
        one = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * N, K, channels=2, max_in=nblk * N)
        assert np.array_equal(np.concatenate([one.process(x[:, :N]), one.process(x[:, N:])], axis=1), y[b])
        for c in range(2):   # every band and channel against the oracle's FilterSink / FilterSource blocks and the closed form
            flt = orc.FFTFilter(K)
            ref = np.concatenate([flt.process(x[c, i * N:(i + 1) * N]) for i in range(nblk)])
            assert rel_err(y[b, c], ref) <= RTOL, (b, c)
        h = sa.design_fftfilt_kernel(N, bands[b][0], bands[b][1], FS)
        hc = h[:, 0].astype(np.float64) + 1j * h[:, 1]
        xc = x[0, :, 0].astype(np.float64) + 1j * x[0, :, 1]
        closed = np.convolve(xc, hc)[:len(xc)] / (np.sqrt(2 * N) * np.sqrt((np.abs(hc) ** 2).sum()))
        assert np.abs(y[b, 0, :, 0].astype(np.float64) + 1j * y[b, 0, :, 1] - closed).max() / np.abs(closed).max() <= RTOL, b
    # a band retuned between calls: from the second block on it equals a plan made with the new kernel from the start
    # (overlap-save: history is INPUT, so one block of transient — see sdrhip.h)
    K2 = sa.design_fftfilt_spectrum(sa.design_fftfilt_kernel(N, 100e3, 300e3, FS))
    bank2 = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * N, Ks, channels=2, max_in=nblk * N)
    bank2.process(x[:, :N]); bank2.set_kernel(1, K2)
    y2 = bank2.process(x[:, N:])
    fresh = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * N, [Ks[0], K2, Ks[2]], channels=2, max_in=nblk * N)
    fresh.process(x[:, :N])
    assert np.array_equal(y2, fresh.process(x[:, N:]))


def test_fftconv_ols_4097_vs_reference_fir(ctx, golden):
    """BASELINE config 4: L=16384, 4097 real taps by overlap-save vs the reference's time-domain
    FIRLowPass<cf32>(4097) output (golden), streaming over 3 calls."""
    a = golden.load("g2_firlp_alpha4097")
    # FIRFilter pairs alpha[order-1] with the newest sample (src/firfilter.hh:237-243), so the causal
    # convolution kernel is h[k] = alpha[order-1-k]
    taps = np.stack([a[::-1], np.zeros_like(a)], axis=1).astype(np.float32)
    node = sa.FFTConv(ctx, sa.FFTCONV_OLS, 16384, taps, max_in=4096)
    x = golden.load("g1_iq_cf32")
    y = np.concatenate([node.process(x[i * 4096:(i + 1) * 4096])[0] for i in range(3)])
    assert rel_err(y, golden.load("g6_fir4097_cf32_out")) <= RTOL


@pytest.mark.parametrize("odd", [True, False])
def test_fftconv_ols_even_taps_odd_hop(ctx, orc, odd, monkeypatch):
    """Overlap-save with an even tap count: hop = L - n_taps + 1 is odd, so blocks start on every alignment (the
    16-byte load/store variants of the first and last pass must step aside). Plans round such a hop down to even by
    themselves (one more sample of history); SDRHIP_FFTCONV_ODD_HOP keeps the odd one for this test."""
    if odd:
        monkeypatch.setenv("SDRHIP_FFTCONV_ODD_HOP", "1")
    rng = np.random.default_rng(12)
    n_taps, L, C = 1000, 16384, 3
    h = (rng.standard_normal((n_taps, 2)) * 0.05).astype(np.float32)
    x = (rng.standard_normal((C, 40000, 2)) * 0.3).astype(np.float32)
    node = sa.FFTConv(ctx, sa.FFTCONV_OLS, L, h, channels=C, max_in=40000)
    y = np.concatenate([node.process(x[:, :25000]), node.process(x[:, 25000:])], axis=1)
    hc = h[:, 0].astype(np.float64) + 1j * h[:, 1]
    for c in range(C):
        xc = x[c, :, 0].astype(np.float64) + 1j * x[c, :, 1]
        ref = np.convolve(xc, hc)[:40000]
        got = y[c, :, 0] + 1j * y[c, :, 1]
        assert np.abs(got - ref).max() / np.abs(ref).max() <= RTOL


@pytest.mark.parametrize("C,n_taps,grid", [(3, 4097, 3), (3, 1000, 2), (16, 4097, 8), (16, 4097, 16), (8, 2049, 8), (8, 2050, 8), (5, 4097, 1), (24, 4097, 40), (4, 8192, 5), (4, 8193, 3)])
def test_fftconv_pipelined_walk_equals_one_block_per_workgroup(ctx, golden, C, n_taps, grid, monkeypatch):
    if n_taps in (1000, 2049 + 1):   # (the 8-byte variant of the pipelined form: an odd hop kept odd)
        monkeypatch.setenv("SDRHIP_FFTCONV_ODD_HOP", "1")
    """The 16384-point plan's pipelined form (persistent workgroups walking their blocks, the next block's inputs in flight
    during the last two inverse passes; fftconv_fused_kernel PIPE) against the one-block-per-workgroup kernel: bit for bit
    (same butterflies, same twiddle products), over three calls of ragged lengths with the history carried, aligned and
    unaligned hops, the XCD-ordered walk (channels and workgroups in multiples of 8) and the plain one, one workgroup
    walking everything, and more workgroups than some calls have blocks (those calls take the old kernel)."""
    rng = np.random.default_rng(100 + C + n_taps)
    if n_taps == 4097:
        a = golden.load("g2_firlp_alpha4097")
        h = np.stack([a[::-1], np.zeros_like(a)], axis=1).astype(np.float32)
    else:
        h = (rng.standard_normal((n_taps, 2)) * 0.05).astype(np.float32)
    x = (rng.standard_normal((C, 90000, 2)) * 0.3).astype(np.float32)
    cuts = [0, 50000, 50001, 90000]
    ys = []
    for g in (0, grid):
        monkeypatch.setenv("SDRHIP_K7_PIPE_GRID", str(g))
        node = sa.FFTConv(ctx, sa.FFTCONV_OLS, 16384, h, channels=C, max_in=50000)
        ys.append(np.concatenate([node.process(x[:, cuts[i]:cuts[i + 1]]) for i in range(3)], axis=1))
    assert np.array_equal(ys[0], ys[1])
    hc = h[:, 0].astype(np.float64) + 1j * h[:, 1]
    for c in (0, C - 1):
        xc = x[c, :, 0].astype(np.float64) + 1j * x[c, :, 1]
        ref = np.convolve(xc, hc)[:90000]
        got = ys[1][c, :, 0] + 1j * ys[1][c, :, 1]
        assert np.abs(got - ref).max() / np.abs(ref).max() <= RTOL


def test_fftconv_matches_time_domain_kernel(ctx, golden):
    """config 4 'fftplan filter vs FIRFilter': both GPU paths agree on a long multi-channel stream."""
    a = golden.load("g2_firlp_alpha4097")
    rng = np.random.default_rng(2)
    x = rng.standard_normal((3, 40000, 2)).astype(np.float32)
    taps = np.stack([a[::-1], np.zeros_like(a)], axis=1).astype(np.float32)
    f1 = sa.FFTConv(ctx, sa.FFTCONV_OLS, 16384, taps, channels=3, max_in=40000)
    f2 = sa.FIR(ctx, sa.FIR_CF32, a, channels=3, max_in=40000)
    assert rel_err(f1.process(x), f2.process(x)) <= RTOL


# ---- float baseband (config 2) -----------------------------------------------------------------------------

def test_float_baseband_vs_oracle(ctx, golden, orc):
    alpha = golden.load("g2_firlp_alpha127")
    x = golden.load("g1_iq_cf32")
    node = sa.FloatBaseBand(ctx, 100e3, FS, alpha, 8, max_in=4096)
    fir, sub = orc.FIR(alpha), orc.SubSample(8)
    for i in range(3):
        xs = x[i * 4096:(i + 1) * 4096]
        y = node.process(xs)[0]
        ref = sub.process_cf32(fir.process_cf32(orc.freqshift_cf32(xs, i * 4096, 100e3, FS)))
        assert y.shape == ref.shape and rel_err(y, ref) <= RTOL


@pytest.mark.parametrize("D,epi", [(8, sa.EPI_NONE), (3, sa.EPI_NONE), (8, sa.EPI_AM), (1, sa.EPI_USB), (16, sa.EPI_NONE)])
def test_fir_cf32_long_calls_vs_oracle(ctx, golden, orc, D, epi):
    """Calls long enough for interior tiles of the register-tiled kernel (a tile is 1024 outputs = 1024*D samples):
    3 channels, ragged call lengths, decimation 8 (specialised instance), 3 and 16 (generic), 1."""
    alpha = golden.load("g2_firlp_alpha127")
    rng = np.random.default_rng(D)
    C = 3
    node = sa.FIR(ctx, sa.FIR_CF32, alpha, decim=D, channels=C, max_in=40000, epilogue=epi)
    firs, subs = [orc.FIR(alpha) for _ in range(C)], [orc.SubSample(D) for _ in range(C)]
    for n in (40000, 1, 33333, 8192 * 2 + 5):
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
            assert y[c].shape == ref.shape
            if ref.size:
                assert rel_err(y[c], ref) <= RTOL


@pytest.mark.parametrize("D", [8, 5])
def test_float_baseband_full_buffers_vs_oracle(ctx, golden, orc, D):
    """BASELINE config 2 at its buffer size: 65536-sample calls (interior tiles, phasor advanced across tiles and calls);
    D = 5 takes the generic-decimation instance of the kernel."""
    alpha = golden.load("g2_firlp_alpha127")
    rng = np.random.default_rng(11)
    C, N = 2, 65536
    node = sa.FloatBaseBand(ctx, 100e3, FS, alpha, D, channels=C, max_in=N)
    firs, subs = [orc.FIR(alpha) for _ in range(C)], [orc.SubSample(D) for _ in range(C)]
    n0 = 0
    for n in (N, N, 12345):
        x = (rng.standard_normal((C, n, 2)) * 0.3).astype(np.float32)
        y = node.process(x)
        for c in range(C):
            ref = subs[c].process_cf32(firs[c].process_cf32(orc.freqshift_cf32(x[c], n0, 100e3, FS)))
            assert y[c].shape == ref.shape and rel_err(y[c], ref) <= RTOL
        n0 += n


@pytest.mark.parametrize("tpw", [2, 3, 16])
@pytest.mark.parametrize("shift", [True, False])
def test_fir_cf32_pipelined_kernel_small_batches(ctx, golden, orc, tpw, shift, monkeypatch):
    """The software-pipelined form of the decimation-8 float kernel (several consecutive tiles per workgroup, the next
    tile prefetched into registers) is chosen for batches that fill the chip; SDRHIP_FIR_TPW forces it on 3 channels:
    ragged calls (first tile with history, unaligned rows after an odd call, partial last tile, calls shorter than a
    tile) against the oracle, and bit for bit against the one-tile-per-workgroup kernel."""
    alpha = golden.load("g2_firlp_alpha127")
    rng = np.random.default_rng(100 + tpw)
    C, lens = 3, [40000, 1, 33333, 16389, 65536, 4096, 20000]
    mk = (lambda: sa.FloatBaseBand(ctx, 100e3, FS, alpha, 8, channels=C, max_in=65536)) if shift else \
         (lambda: sa.FIR(ctx, sa.FIR_CF32, alpha, decim=8, channels=C, max_in=65536))
    monkeypatch.setenv("SDRHIP_FIR_PIPE", "0")
    plain = mk()
    monkeypatch.setenv("SDRHIP_FIR_PIPE", "1")
    monkeypatch.setenv("SDRHIP_FIR_TPW", str(tpw))
    node = mk()
    assert node.kernel_names(65536) == ["fir_cf32_pipe_kernel"] and plain.kernel_names(65536) == ["fir_cf32_rt_kernel"]
    firs, subs = [orc.FIR(alpha) for _ in range(C)], [orc.SubSample(8) for _ in range(C)]
    n0 = 0
    for n in lens:
        x = (rng.standard_normal((C, n, 2)) * 0.3).astype(np.float32)
        y, yp = node.process(x), plain.process(x)
        for c in range(C):
            xs = orc.freqshift_cf32(x[c], n0, 100e3, FS) if shift else x[c]
            ref = subs[c].process_cf32(firs[c].process_cf32(xs))
            assert y[c].shape == ref.shape
            assert np.array_equal(y[c], yp[c]), (n, c)
            if ref.size:
                assert rel_err(y[c], ref) <= RTOL, (n, c)
        n0 += n


def test_fftconv_config4_many_channels(ctx, golden, orc):
    """BASELINE config 4 (ii) at C = 64, N = 65536: the fused overlap-save kernel against the time-domain oracle FIR
    on three of the channels, and against itself run one channel at a time."""
    a = golden.load("g2_firlp_alpha4097")
    taps = np.stack([a[::-1], np.zeros_like(a)], axis=1).astype(np.float32)
    rng = np.random.default_rng(4)
    C, N = 64, 65536
    x = (rng.standard_normal((C, N, 2)) * 0.3).astype(np.float32)
    y = sa.FFTConv(ctx, sa.FFTCONV_OLS, 16384, taps, channels=C, max_in=N).process(x)
    for c in (0, 31, 63):
        ref = orc.FIR(a).process_cf32(x[c])
        assert rel_err(y[c], ref) <= RTOL
    one = sa.FFTConv(ctx, sa.FFTCONV_OLS, 16384, taps, channels=1, max_in=N)
    assert np.array_equal(one.process(x[5:6])[0], y[5])


# ---- the other BASELINE configs at their batch size (1024 channels x 65536 samples per call) -------------------
# Grid-dimension / stride / >2^31-byte-offset bugs only show at full size: 8 distinct channel patterns are tiled over
# the 1024 channels; batching invariance (equal inputs -> equal outputs wherever the channel sits) plus the 8 patterns
# against the oracle over two calls (state carried).

def _tiled(base, C):
    return np.ascontiguousarray(base[np.arange(C) % base.shape[0]])


def test_fir255_fm_full_size(ctx, orc):
    """BASELINE config 3: 1024 int16 IQ channels, FIRLowPass<cs16>(255) -> FMDemod, 65536 samples per call."""
    C, N = 1024, 65536
    alpha = sa.design_fir_lowpass(255, 100e3, FS)
    base = synth_channels(orc, 8, 2 * N)
    x = _tiled(base, C)
    node = sa.FIR(ctx, sa.FIR_CS16_EXACT, alpha, channels=C, max_in=N, epilogue=sa.EPI_FM)
    ys = [node.process(x[:, :N]), node.process(x[:, N:])]
    for y in ys:
        assert y.shape == (C, N)
        for k in range(8):
            assert (y[k::8] == y[k]).all()
    for k in range(8):
        f, fm = orc.FIR(alpha), orc.FMDemodI16()
        for i, y in enumerate(ys):
            assert np.array_equal(y[k], fm.process(f.process_cs16(base[k, i * N:(i + 1) * N]))), (k, i)


def test_float_baseband_full_size(ctx, golden, orc):
    """BASELINE config 2's chain (shift -> FIRLowPass<cf32>(127) -> /8) at 1024 channels x 65536 samples per call."""
    C, N, D = 1024, 65536, 8
    alpha = golden.load("g2_firlp_alpha127")
    rng = np.random.default_rng(21)
    base = (rng.standard_normal((8, 2 * N, 2)) * 0.3).astype(np.float32)
    x = _tiled(base, C)
    node = sa.FloatBaseBand(ctx, 100e3, FS, alpha, D, channels=C, max_in=N)
    ys = [node.process(x[:, :N]), node.process(x[:, N:])]
    for y in ys:
        for k in range(8):
            assert np.array_equal(y[k::8], np.broadcast_to(y[k], y[k::8].shape))
    for k in range(8):
        fir, sub = orc.FIR(alpha), orc.SubSample(D)
        for i, y in enumerate(ys):
            ref = sub.process_cf32(fir.process_cf32(orc.freqshift_cf32(base[k, i * N:(i + 1) * N], i * N, 100e3, FS)))
            assert y[k].shape == ref.shape and rel_err(y[k], ref) <= RTOL, (k, i)


def test_fir_cf32_decimated_full_size(ctx, golden, orc):
    """K3 alone (FIRLowPass<cf32>(127) -> SubSample(8)) at 1024 channels x 65536 samples per call."""
    C, N, D = 1024, 65536, 8
    alpha = golden.load("g2_firlp_alpha127")
    rng = np.random.default_rng(22)
    base = (rng.standard_normal((8, 2 * N, 2)) * 0.3).astype(np.float32)
    x = _tiled(base, C)
    node = sa.FIR(ctx, sa.FIR_CF32, alpha, decim=D, channels=C, max_in=N)
    ys = [node.process(x[:, :N]), node.process(x[:, N:])]
    for y in ys:
        for k in range(8):
            assert np.array_equal(y[k::8], np.broadcast_to(y[k], y[k::8].shape))
    for k in range(8):
        fir, sub = orc.FIR(alpha), orc.SubSample(D)
        for i, y in enumerate(ys):
            ref = sub.process_cf32(fir.process_cf32(base[k, i * N:(i + 1) * N]))
            assert y[k].shape == ref.shape and rel_err(y[k], ref) <= RTOL, (k, i)


def test_fftconv_config4_full_size(ctx, golden, orc):
    """BASELINE config 4 (ii) at 1024 channels x 65536: overlap-save L = 16384, 4097 taps, against the time-domain
    oracle FIR on 3 of the 8 patterns (2 calls, history carried) and batching invariance over all channels."""
    a = golden.load("g2_firlp_alpha4097")
    taps = np.stack([a[::-1], np.zeros_like(a)], axis=1).astype(np.float32)
    rng = np.random.default_rng(23)
    C, N = 1024, 65536
    base = (rng.standard_normal((8, 2 * N, 2)) * 0.3).astype(np.float32)
    x = _tiled(base, C)
    node = sa.FFTConv(ctx, sa.FFTCONV_OLS, 16384, taps, channels=C, max_in=N)
    ys = [node.process(x[:, :N]), node.process(x[:, N:])]
    del x
    for y in ys:
        for k in range(8):
            assert np.array_equal(y[k::8], np.broadcast_to(y[k], y[k::8].shape))
    for k in (0, 3, 7):
        f = orc.FIR(a)
        for i, y in enumerate(ys):
            assert rel_err(y[k], f.process_cf32(base[k, i * N:(i + 1) * N])) <= RTOL, (k, i)


def test_fftconv_reference_mode_full_size(ctx, orc):
    """BASELINE config 4 (i) at 1024 channels: the reference's overlap-add mode (block 8192, FFT 16384, 8192-tap
    FilterSource kernel 50..150 kHz), 8 blocks per call, against the oracle's FilterSink/FilterSource restatement."""
    Nb, C, N = 8192, 1024, 65536
    h = sa.design_fftfilt_kernel(Nb, 50e3, 150e3, FS)
    K = sa.design_fftfilt_spectrum(h)
    rng = np.random.default_rng(24)
    base = (rng.standard_normal((8, 2 * N, 2)) * 0.3).astype(np.float32)
    x = _tiled(base, C)
    node = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * Nb, K, channels=C, max_in=N)
    ys = [node.process(x[:, :N]), node.process(x[:, N:])]
    del x
    for y in ys:
        for k in range(8):
            assert np.array_equal(y[k::8], np.broadcast_to(y[k], y[k::8].shape))
    for k in (1, 6):
        f = orc.FFTFilter(orc.fftfilt_design_K(h))
        for i, y in enumerate(ys):
            ref = np.concatenate([f.process(base[k, i * N + b * Nb:i * N + (b + 1) * Nb]) for b in range(N // Nb)])
            assert rel_err(y[k], ref) <= RTOL, (k, i)


def test_iqbb_usb_full_size(ctx, orc):
    """BASELINE config 5's per-GPU shard: 1024 channels, IQBaseBand<int16>(127, /8) -> USBDemod, 65536 per call."""
    C, N, D = 1024, 65536, 8
    taps = sa.design_iqbb_taps(100e3, 50e3, FS, 127)
    lut = sa.design_freqshift_lut_i16()
    base = synth_channels(orc, 8, 2 * N)
    x = _tiled(base, C)
    node = sa.IQBaseBandI16(ctx, taps, lut, 1365, False, D, channels=C, max_in=N, epilogue=sa.EPI_USB)
    ys = [node.process(x[:, :N]), node.process(x[:, N:])]
    for k in range(8):
        bb = orc.IQBaseBandI16(taps, lut, 1365, False, D)
        for i, y in enumerate(ys):
            assert (y[k::8] == y[k]).all()
            assert np.array_equal(y[k], orc.usb_i16(bb.process(base[k, i * N:(i + 1) * N]))), (k, i)


# ---- error behaviour -----------------------------------------------------------------------------------------

def test_process_dev_rejects_overlapping_ranges(ctx, golden):
    """sdrhip.h: *_process_dev output must not overlap the input (tile-parallel kernels would race): E_INVALID."""
    taps, lut = golden.load("g3_iqbb127d8_taps"), golden.load("g3_iqbb127d8_lut")
    n = 4096
    buf = ctx.malloc(2 * n * 8)
    try:
        bb = sa.IQBaseBandI16(ctx, taps, lut, 1365, 0, 8, max_in=n)
        fir = sa.FIR(ctx, sa.FIR_CS16_EXACT, golden.load("g2_firlp_alpha127"), max_in=n)
        dem = sa.Demod(ctx, sa.EPI_USB, sa.T_CS16, max_in=n)
        for call in (lambda: bb.process_dev(buf, n, n, buf, n), lambda: fir.process_dev(buf, n, n, buf + 4 * (n - 1), n),
                     lambda: dem.process_dev(buf, n, n, buf, n)):
            with pytest.raises(sa.SdrHipError) as e:
                call()
            assert e.value.code == sa.abi.E_INVALID and "overlap" in str(e.value)
        assert bb.process_dev(buf, n, n, buf + 4 * n, n) == 511      # adjacent, disjoint ranges are fine
        ctx.synchronize()
    finally:
        ctx.free(buf)


def test_error_codes(ctx, golden):
    taps, lut = golden.load("g3_iqbb127d8_taps"), golden.load("g3_iqbb127d8_lut")
    with pytest.raises(sa.SdrHipError) as e:
        sa.IQBaseBandI16(ctx, taps, lut, 1365, 0, 0)
    assert e.value.code == sa.abi.E_INVALID
    with pytest.raises(sa.SdrHipError) as e:
        sa.IQBaseBandI16(ctx, taps, lut, 1365, 0, 40000)   # (beyond 32768: D * D wraps in the box average's division)
    assert e.value.code == sa.abi.E_UNSUPPORTED
    node = sa.IQBaseBandI16(ctx, taps, lut, 1365, 0, 8, max_in=128)
    with pytest.raises(sa.SdrHipError) as e:
        node.process(np.zeros((1, 129, 2), np.int16))
    assert e.value.code == sa.abi.E_SIZE
    assert node.process(np.zeros((1, 0, 2), np.int16)).shape == (1, 0, 2)
    with pytest.raises(sa.SdrHipError) as e:
        sa.Demod(ctx, sa.EPI_FM, sa.T_CF32)
    assert e.value.code == sa.abi.E_UNSUPPORTED
    with pytest.raises(sa.SdrHipError) as e:
        sa.Context(99)
    assert e.value.code == sa.abi.E_NODEVICE


# ---- fast_atan2 over its whole domain ---------------------------------------------------------------------

def test_fm_angle_every_int16_pair(ctx):
    """fm_phi (libsdr_amd/csrc/fm_phi.hpp: one float-estimated division + an exact remainder test) against the
    reference's formula (src/math.hh:31-40, src/demod.hh:246) in exact 64-bit integer arithmetic, for EVERY (a, b) in
    int16 x int16 — 2^32 pairs through the stand-alone FMDemod<int16_t> kernel: every second sample is (0, 0), whose
    angle is 0, so output 2k+1 is -angle(pair k)."""
    dev = torch.device("cuda:0")
    NA = 1024                                   # a-values per call: 2^26 pairs, 2^27 samples
    n = 2 * NA * 65536
    node = sa.Demod(ctx, sa.EPI_FM, sa.T_CS16, max_in=n, inplace_fm0=False)
    b = (torch.arange(65536, device=dev, dtype=torch.int64) - 32768).repeat(NA)
    x = torch.zeros(n, dtype=torch.int32, device=dev)
    y = torch.empty(n, dtype=torch.int16, device=dev)
    for a0 in range(-32768, 32768, NA):
        a = (torch.arange(NA, device=dev, dtype=torch.int64) + a0).repeat_interleave(65536)
        x[1::2] = ((a & 0xffff) | ((b & 0xffff) << 16)).to(torch.int32)   # (int64 -> int32 wraps: the packed dword)
        torch.cuda.synchronize()
        node.reset()
        node.process_dev(x.data_ptr(), n, n, y.data_ptr(), n)
        ctx.synchronize()
        aabs = a.abs()
        num = 4096 * torch.where(b >= 0, b - aabs, b + aabs)
        den = torch.where(b >= 0, b + aabs, aabs - b)
        q = torch.div(num, den.clamp(min=1), rounding_mode="trunc")
        angle = torch.where(b >= 0, 4096, 12288) - q
        angle = torch.where((a == 0) & (b == 0), torch.zeros_like(angle), angle)
        angle = torch.where(a >= 0, angle, -angle)
        phi = torch.div(angle, 2, rounding_mode="trunc")
        got = y[1::2].to(torch.int64)
        bad = (got != -phi).nonzero()
        assert bad.numel() == 0, (a0, int(a[bad[0, 0]]), int(b[bad[0, 0]]), int(got[bad[0, 0]]), int(-phi[bad[0, 0]]))


# ---- the red-zoned arena itself (tests/conftest.py `redzone`, tests/redzone.py) ------------------------------

def test_redzone_arena_is_live_and_catches_stray_writes(ctx, golden, redzone):
    from redzone import RedZone
    m, node = iqbb_from_case(ctx, golden, "g3_iqbb127d8", "_out", sa.EPI_NONE)
    before = RedZone.calls
    node.process(golden.load("g1_iq_cs16")[:4096])
    assert RedZone.active == (redzone == "redzone") and RedZone.calls == before + (1 if redzone == "redzone" else 0)
    x, out = np.zeros((2, 64, 2), np.int16), np.zeros((2, 8), np.int16)
    eb = 2
    # one element past the end of row 0, one before row 0, one past the last row, and a write into the input arena
    for where in ("after_row0", "before_row0", "after_last", "input"):
        def stray(i, si, o, so, where=where):
            if where == "input":
                ctx.memset(i + 3 * 4, 0x11, 4)
            else:
                off = {"after_row0": 8 * eb, "before_row0": -eb, "after_last": (so + 8) * eb}[where]
                ctx.memset(o + off, 0x11, eb)
        with pytest.raises(AssertionError, match="red zone"):
            RedZone.run(ctx, x, out, stray)
    RedZone.run(ctx, x, out, lambda i, si, o, so: ctx.memset(o, 0, 8 * eb))   # writing the rows themselves is fine
