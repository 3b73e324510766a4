"""Pins the CPU restatement (oracle/sdr_oracle.cc) to golden vectors cut from the compiled,
unmodified reference (oracle/ref_driver.cc -> tests/golden/).  CPU only.

Bars: bit-exact for every int16/int32 quantity and for double taps designed with the same libm;
float paths bit-exact where the reference arithmetic is restated operation by operation.
"""
import numpy as np
import pytest

FS = 2.4e6


def split(x, lens):
    out, off = [], 0
    for n in lens:
        out.append(x[off:off + n])
        off += n
    assert off == len(x)
    return out


# ---- generators -------------------------------------------------------------------------------

def test_iqsiggen_cs16(golden, orc):
    g = orc.IQSigGen(FS, [(100e3, 8000, 0.0), (-300e3, 6000, 0.3)])
    x = np.concatenate([g.next_cs16(4096) for _ in range(4)])
    assert np.array_equal(x, golden.load("g1_iq_cs16"))
    # SURVEY Appendix B literals
    assert x[0].tolist() == [6866, 886] and x[3].tolist() == [1428, 174]
    assert x[4095].tolist() == [-1428, -174] and x[4096].tolist() == [866, -2577]


def test_iqsiggen_cf32(golden, orc):
    g = orc.IQSigGen(FS, [(100e3, 0.5, 0.0), (-300e3, 0.3, 0.3)])
    x = np.concatenate([g.next_cf32(4096) for _ in range(3)])
    assert np.array_equal(x, golden.load("g1_iq_cf32"))


def test_iqsiggen_single_tones(golden, orc):
    g = orc.IQSigGen(FS, [(100e3, 8000, 0.0)])
    assert np.array_equal(np.concatenate([g.next_cs16(4096) for _ in range(2)]),
                          golden.load("g1_iq_cs16_tone_p100k"))
    g = orc.IQSigGen(FS, [(-100e3, 8000, 0.0)])
    assert np.array_equal(g.next_cs16(4096), golden.load("g1_iq_cs16_tone_m100k"))


# ---- designers --------------------------------------------------------------------------------

IQBB_CASES = ["g3_iqbb127d8", "g8_neg_o16_d1", "g8_o21_d3", "g8_o33_d5", "g8_o16_d4_even",
              "g8_o255_d8", "g8_noshift_o21_d8", "g8_ofs_d300"]


@pytest.mark.parametrize("case", IQBB_CASES)
def test_iqbb_design(golden, orc, case):
    m = golden.meta(case + "_taps")
    taps = orc.iqbb_design(m["Ff"], m["width"], m["Fs"], m["order"])
    assert np.array_equal(taps.ravel(), golden.load(case + "_taps"))
    assert np.array_equal(orc.freqshift_lut_i16().ravel(), golden.load(case + "_lut"))
    assert orc.freqshift_inc(m["Fc"], m["Fs"]) == m["lut_inc"]
    assert orc.iqbb_decim(m["Fs"], m["sub"], m["oFs"]) == m["decim"]


def test_iqbb_design_known_answers(golden, orc):
    t = orc.iqbb_design(100e3, 50e3, FS, 127)
    assert t[0].tolist() == [0, 0] and t[126].tolist() == [0, 0]
    assert t[62].tolist() == [-345, 199] and t[63].tolist() == [-283, 283] and t[64].tolist() == [-200, 346]
    assert t.sum(0).tolist() == [2, 1]
    lut = orc.freqshift_lut_i16()
    assert lut[0].tolist() == [65536, 0] and lut[1].tolist() == [65457, -3215]
    assert lut[32].tolist() == [0, -65536] and lut[127].tolist() == [65457, 3215]
    assert orc.freqshift_inc(100e3, FS) == 1365


@pytest.mark.parametrize("N", [127, 255, 4097])
def test_fir_lowpass_design(golden, orc, N):
    a = orc.fir_lowpass_design(N, 100e3, FS)
    assert np.array_equal(a, golden.load("g2_firlp_alpha%d" % N))
    if N == 127:
        assert a[0] == 3.8459023977995984e-20 and a[63] == 0.057890032145963485


@pytest.mark.parametrize("N", [1024, 8192])
def test_fftfilt_kernel_h(golden, orc, N):
    h = orc.fftfilt_design_h(N, 50e3, 150e3, FS)
    assert np.array_equal(h, golden.load("g7_fftfilt_h%d" % N))


def test_fftfilt_kernel_h_any_block_size_and_double(golden, orc):
    """g15: sinc_flt_kernel<float> at N = 1000 (FilterNode takes any block size) and sinc_flt_kernel<double>."""
    assert np.array_equal(orc.fftfilt_design_h(1000, -350e3, -250e3, FS), golden.load("g15_fftfilt_h1000"))
    for N in (1000, 1024):
        assert np.array_equal(orc.fftfilt_design_h_f64(N, -350e3, -250e3, FS).ravel(), golden.load("g15_fftfilt_h%d_f64" % N))


# ---- IQBaseBand<int16_t> ------------------------------------------------------------------------

def run_iqbb(golden, orc, case, inp, suffix):
    m = golden.meta(case + suffix)
    tcase = case if (case + "_taps") in golden.manifest else "g3_iqbb127d8"
    taps = golden.load(tcase + "_taps")
    lut = golden.load(tcase + "_lut")
    bb = orc.IQBaseBandI16(taps, lut, m["lut_inc"], m["negative"], m["decim"])
    x = golden.load(inp)
    chunks = split(x, m["in_lens"])
    return m, [bb.process(c) for c in chunks]


@pytest.mark.parametrize("case,inp", [
    ("g3_iqbb127d8", "g1_iq_cs16"), ("g8_neg_o16_d1", "g1_iq_cs16_tone_m100k"),
    ("g8_o21_d3", "g1_iq_cs16"), ("g8_o33_d5", "g1_iq_cs16"), ("g8_o16_d4_even", "g1_iq_cs16"),
    ("g8_o255_d8", "g1_iq_cs16"), ("g8_noshift_o21_d8", "g1_iq_cs16"), ("g8_ofs_d300", "g1_iq_cs16"),
    ("g8_irregular", "g1_iq_cs16")])
def test_iqbb_i16(golden, orc, case, inp):
    m, outs = run_iqbb(golden, orc, case, inp, "_out")
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load(case + "_out"))


def test_iqbb_i16_known_answers(golden, orc):
    m, outs = run_iqbb(golden, orc, "g3_iqbb127d8", "g1_iq_cs16", "_out")
    assert [len(o) for o in outs] == [511, 512, 512, 512]
    y0, y1 = outs[0].astype(np.int64), outs[1].astype(np.int64)
    assert y0[15].tolist() == [121, -3978] and y0[510].tolist() == [1084, -3829]
    i = np.arange(len(y0))
    assert int((y0[:, 0] * (i + 1) + y0[:, 1] * (i + 7)).sum()) == -418559318
    i = np.arange(len(y1))
    assert int((y1[:, 0] * (i + 1) + y1[:, 1] * (i + 7)).sum()) == -246805328
    _, o = run_iqbb(golden, orc, "g8_neg_o16_d1", "g1_iq_cs16_tone_m100k", "_out")
    assert o[0][100].tolist() == [-5466, -5837] and o[0][101].tolist() == [-5560, -5747]


@pytest.mark.parametrize("case,inp,demod", [
    ("g4_iqbb127d8", "g1_iq_cs16", "fm"), ("g4_iqbb127d8", "g1_iq_cs16", "am"),
    ("g4_iqbb127d8", "g1_iq_cs16", "usb"), ("g8_o33_d5", "g1_iq_cs16", "fm"),
    ("g8_irregular", "g1_iq_cs16", "fm"), ("g8_irregular", "g1_iq_cs16", "usb"),
    ("g8_loud_iqbb127d8", "g8_iq_cs16_loud", "fm"), ("g8_loud_iqbb127d8", "g8_iq_cs16_loud", "am")])
def test_iqbb_demod_chain(golden, orc, case, inp, demod):
    m, outs = run_iqbb(golden, orc, case, inp, "_" + demod)
    fm = orc.FMDemodI16()
    res = []
    for y in outs:
        if demod == "fm":
            if len(y):               # FMDemod::process returns without send on an empty buffer
                res.append(fm.process(y, inplace=True))
        elif demod == "am":
            res.append(orc.am_i16(y))
        else:
            res.append(orc.usb_i16(y))
    assert [len(r) for r in res] == m["out_lens"]
    assert np.array_equal(np.concatenate(res), golden.load(case + "_" + demod))


def test_fm_known_answers(golden, orc):
    f = golden.load("g4_iqbb127d8_fm").astype(np.int64)
    b0, b1 = f[:511], f[511:1023]
    assert b0[0] == 0 and b0[16] == 10 and b0[17] == 8 and b0[510] == -13
    assert int((b0[1:] * (np.arange(1, 511) + 1)).sum()) == 97546
    assert b1[0] == 1093 and b1[1] == 45 and int((b1[1:] * (np.arange(1, 512) + 1)).sum()) == 143983


# ---- FIRFilter ----------------------------------------------------------------------------------

@pytest.mark.parametrize("case,order,inp", [
    ("g5_fir127", 127, "g1_iq_cs16"), ("g5_fir255", 255, "g1_iq_cs16"),
    ("g8_irregular_fir127", 127, "g1_iq_cs16"), ("g8_loud_fir127", 127, "g8_iq_cs16_loud")])
def test_fir_cs16(golden, orc, case, order, inp):
    m = golden.meta(case + "_out")
    fir = orc.FIR(golden.load("g2_firlp_alpha%d" % order))
    outs = [fir.process_cs16(c) for c in split(golden.load(inp), m["in_lens"])]
    assert np.array_equal(np.concatenate(outs), golden.load(case + "_out"))


@pytest.mark.parametrize("case,order", [("g5_fir127", 127), ("g5_fir255", 255), ("g8_irregular_fir127", 127)])
def test_fir_cs16_fm_inplace(golden, orc, case, order):
    m = golden.meta(case + "_fm")
    fir = orc.FIR(golden.load("g2_firlp_alpha%d" % order))
    fm = orc.FMDemodI16()
    res = []
    for c in split(golden.load("g1_iq_cs16"), m["in_lens"]):
        y = fir.process_cs16(c)
        if len(y):
            res.append(fm.process(y, inplace=True))
    assert [len(r) for r in res] == m["out_lens"]
    assert np.array_equal(np.concatenate(res), golden.load(case + "_fm"))


def test_fir_tone_known_answer(golden, orc):
    x = golden.load("g1_iq_cs16_tone_p100k")[4096:]
    fir = orc.FIR(golden.load("g2_firlp_alpha127"))
    y = fir.process_cs16(x)
    assert np.array_equal(y, golden.load("g5_fir127_tone_buf2_out"))
    assert y[200].tolist() == [-2207, 1646] and y[201].tolist() == [-2566, 1008]
    f = orc.FMDemodI16().process(y)
    assert np.array_equal(f, golden.load("g5_fir127_tone_buf2_fm")) and f[200] == 543


@pytest.mark.parametrize("order", [127, 4097])
def test_fir_cf32(golden, orc, order):
    fir = orc.FIR(golden.load("g2_firlp_alpha%d" % order))
    x = golden.load("g1_iq_cf32")
    y = np.concatenate([fir.process_cf32(x[i * 4096:(i + 1) * 4096]) for i in range(3)])
    assert np.array_equal(y, golden.load("g6_fir%d_cf32_out" % order))


# ---- SubSample / standalone demods --------------------------------------------------------------

@pytest.mark.parametrize("n", [8, 3])
def test_subsample(golden, orc, n):
    s = orc.SubSample(n)
    x = golden.load("g1_iq_cs16")
    outs = [s.process_cs16(x[i * 4096:(i + 1) * 4096]) for i in range(4)]
    assert [len(o) for o in outs] == golden.meta("g6_subsample_cs16_n%d" % n)["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load("g6_subsample_cs16_n%d" % n))
    s = orc.SubSample(n)
    xf = golden.load("g6_fir127_cf32_out")
    outs = [s.process_cf32(xf[i * 4096:(i + 1) * 4096]) for i in range(3)]
    assert [len(o) for o in outs] == golden.meta("g6_fir127_cf32_sub%d" % n)["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load("g6_fir127_cf32_sub%d" % n))


def test_demods_raw(golden, orc):
    x = golden.load("g1_iq_cs16")
    assert np.array_equal(orc.am_i16(x), golden.load("g4_raw_am"))
    assert np.array_equal(orc.usb_i16(x), golden.load("g4_raw_usb"))
    fm = orc.FMDemodI16()
    f = np.concatenate([fm.process(x[i * 4096:(i + 1) * 4096], inplace=False) for i in range(4)])
    assert np.array_equal(f, golden.load("g4_raw_fm_masked0"))
    xf = golden.load("g6_fir127_cf32_out")
    assert np.array_equal(orc.am_f32(xf), golden.load("g6_fir127_cf32_am"))
    assert np.array_equal(orc.usb_f32(xf), golden.load("g6_fir127_cf32_usb"))


def test_fast_atan2(golden, orc):
    t = golden.load("g8_fast_atan2_triples").reshape(-1, 3)
    for a, b, r in t.tolist():
        assert orc.fast_atan2_i16(a, b) == r
    assert orc.fast_atan2_i16(1000, 3) == 8167 and orc.fast_atan2_i16(12345, -6789) == 11099


# ---- FFT convolution: unpinned at the FFTW boundary; checked against its closed form ------------

@pytest.mark.parametrize("N", [1024])
def test_fftfilt_closed_form(golden, orc, N):
    """y[n] = sum_k h[k] x[n-k] / (sqrt(2N) ||h||_2)  (SURVEY fact 7), float64 direct convolution."""
    h = golden.load("g7_fftfilt_h%d" % N)
    K = orc.fftfilt_design_K(h)
    flt = orc.FFTFilter(K)
    x = golden.load("g1_iq_cf32")[:4 * N]
    y = np.concatenate([flt.process(x[i * N:(i + 1) * N]) for i in range(4)])
    hc = h[:, 0].astype(np.float64) + 1j * h[:, 1]
    xc = x[:, 0].astype(np.float64) + 1j * x[:, 1]
    ref = np.convolve(xc, hc)[:4 * N] / (np.sqrt(2 * N) * np.sqrt((np.abs(hc) ** 2).sum()))
    yc = y[:, 0].astype(np.float64) + 1j * y[:, 1]
    assert np.abs(yc - ref).max() / np.abs(ref).max() < 1e-5


@pytest.mark.parametrize("n", [1, 2, 3, 256, 1000, 2000, 2002, 97, 3000])
def test_dft_helper(orc, n):
    """the oracle's own DFT (any length: the reference hands every size to FFTW) against numpy, both directions"""
    import ctypes as C
    rng = np.random.default_rng(n)
    x = rng.standard_normal(2 * n)
    o = np.zeros(2 * n)
    for sign, ref in ((-1, np.fft.fft(x[0::2] + 1j * x[1::2])), (+1, np.fft.ifft(x[0::2] + 1j * x[1::2]) * n)):
        orc.lib().orc_dft_f64(n, sign, x.ctypes.data_as(C.POINTER(C.c_double)), o.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.abs((o[0::2] + 1j * o[1::2]) - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())


def test_fftfilt_closed_form_block_1000_and_double(golden, orc):
    """FilterNode(1000) and FilterNode<double>: the overlap-add filter equals y = h (*) x / (sqrt(2N) ||h||_2) for a block
    size that is not a power of two (float) and in double (1e-12)."""
    N = 1000
    h = golden.load("g15_fftfilt_h1000")
    flt = orc.FFTFilter(orc.fftfilt_design_K(h))
    x = golden.load("g1_iq_cf32")[:4 * N]
    y = np.concatenate([flt.process(x[i * N:(i + 1) * N]) for i in range(4)])
    hc = h[:, 0].astype(np.float64) + 1j * h[:, 1]
    xc = x[:, 0].astype(np.float64) + 1j * x[:, 1]
    ref = np.convolve(xc, hc)[:4 * N] / (np.sqrt(2 * N) * np.sqrt((np.abs(hc) ** 2).sum()))
    assert np.abs(y[:, 0].astype(np.float64) + 1j * y[:, 1] - ref).max() / np.abs(ref).max() < 1e-5
    for N in (1000, 1024):
        hd = golden.load("g15_fftfilt_h%d_f64" % N).reshape(-1, 2)
        fl = orc.FFTFilterF64(orc.fftfilt_design_K_f64(hd))
        xd = x[:3 * N].astype(np.float64)
        y = np.concatenate([fl.process(xd[i * N:(i + 1) * N]) for i in range(3)])
        hc, xc = hd[:, 0] + 1j * hd[:, 1], xd[:, 0] + 1j * xd[:, 1]
        ref = np.convolve(xc, hc)[:3 * N] / (np.sqrt(2 * N) * np.sqrt((np.abs(hc) ** 2).sum()))
        assert np.abs(y[:, 0] + 1j * y[:, 1] - ref).max() / np.abs(ref).max() < 1e-12


# ---- "next" rows: AutoCast cu8 -> cs16 in front, FMDeemph behind ---------------------------

def test_autocast_cu8(golden, orc):
    u = golden.load("g9_iq_cu8").reshape(-1, 2)
    y = orc.autocast_cu8_cs16(u)
    assert np.array_equal(y, golden.load("g9_autocast_cs16"))
    # SURVEY §8f literals: (0,127)->(-32512,0), (128,200)->(256,18688), (255,1)->(-32768,-32256)
    assert y[0].tolist() == [-32512, 0] and y[1].tolist() == [256, 18688] and y[2].tolist() == [-32768, -32256]


@pytest.mark.parametrize("rate", [125000, 48000])
def test_fmdeemph(golden, orc, rate):
    x = golden.load("g9_deemph_in")
    de = orc.FMDeemphI16(float(rate))
    assert de.alpha == (4 if rate == 48000 else 10)
    y = np.concatenate([de.process(x[i * 512:(i + 1) * 512]) for i in range(3)])
    assert np.array_equal(y, golden.load("g9_deemph_out_%d" % rate))


@pytest.mark.parametrize("order", [21, 127])
def test_sdr_fm_chain_cu8(golden, orc, order):
    """cu8 -> AutoCast -> IQBaseBand(order, /8) -> FMDemod (in place) -> FMDeemph: the DSP of examples/sdr_fm.cc."""
    name = "g9_cu8_iqbb%dd8" % order
    m = golden.meta(name + "_taps")
    u = golden.load("g9_iq_cu8").reshape(-1, 2)
    bb = orc.IQBaseBandI16(golden.load(name + "_taps"), orc.freqshift_lut_i16(), m["lut_inc"], 0, 8)
    fm, de = orc.FMDemodI16(), orc.FMDeemphI16(1e6 / 8)
    f_all, d_all = [], []
    for b in range(3):
        f = fm.process(bb.process(orc.autocast_cu8_cs16(u[b * 4096:(b + 1) * 4096])))
        f_all.append(f); d_all.append(de.process(f))
    assert np.array_equal(np.concatenate(f_all), golden.load(name + "_fm"))
    assert np.array_equal(np.concatenate(d_all), golden.load(name + "_fm_deemph"))


# ---- "next" row 3: BaseBand<int16_t>, real input (src/baseband.hh:305-529) -----------------------------------

BB_REAL_CASES = [("g10_bb21d8", "g10_real_in"), ("g10_bb127d8_neg_ragged", "g10_real_in"), ("g10_bb64d5", "g10_real_in"),
                 ("g10_bb16d1_noshift", "g10_real_in"), ("g10_bb1d3", "g10_real_in"), ("g10_bb127d8_loud", "g10_real_loud_in")]


@pytest.mark.parametrize("case,inp", BB_REAL_CASES)
def test_bb_real_design(golden, orc, case, inp):
    m = golden.meta(case + "_taps")
    assert np.array_equal(orc.bb_design(m["Ff"], m["width"], m["Fs"], m["order"]), golden.load(case + "_taps").reshape(-1, 2))
    assert orc.freqshift_inc(m["Fc"], m["Fs"]) == m["lut_inc"]


@pytest.mark.parametrize("case,inp", BB_REAL_CASES)
def test_bb_real_i16(golden, orc, case, inp):
    m = golden.meta(case + "_out")
    bb = orc.BaseBandI16(golden.load(case + "_taps"), orc.freqshift_lut_i16(), m["lut_inc"], m["negative"], m["decim"])
    x = golden.load(inp)
    outs, off = [], 0
    for n in m["in_lens"]:
        outs.append(bb.process(x[off:off + n])); off += n
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load(case + "_out"))


def replay_bb_real_retune(m, x, make_node):
    """g16's event list on a real-input node made by make_node(lut_inc, negative, bufsize): "feed n", "shift F"
    (setFrequencyShift: node.set_shift), "bufsize n" (a new source Config: node = node.reconfigured(bufsize))."""
    Fs, F = float(m["Fs"]), float(m["Fc"])
    node, outs, off = None, [], 0
    for ev in m["events"]:
        if ev[0] == "feed":
            if node is None:
                node = make_node(F)
            outs.append(node.process(x[off:off + ev[1]])); off += ev[1]
        elif ev[0] == "shift":
            F = float(ev[1])
            node.set_shift_hz(F)
        elif ev[0] == "bufsize":
            node = node.reconfigured(int(ev[1]), F)
    return outs


def test_bb_real_retune_midstream(golden, orc):
    """g16: BaseBand<int16_t> through setFrequencyShift and a new source Config between buffers."""
    m = golden.meta("g16_bb_real_retune_out")
    Fs = float(m["Fs"])
    taps, lut = orc.bb_design(m["Ff"], m["width"], Fs, m["order"]), orc.freqshift_lut_i16()

    class Node:
        def __init__(self, F):
            self.bb = orc.BaseBandI16(taps, lut, orc.freqshift_inc(F, Fs), F < 0, m["decim"])

        def process(self, x):
            return self.bb.process(x)

        def set_shift_hz(self, F):
            self.bb.set_shift(orc.freqshift_inc(F, Fs), F < 0)

        def reconfigured(self, bufsize, F):   # config(): LUT phase and counters restart, the ring stays
            self.bb.set_shift(orc.freqshift_inc(F, Fs), F < 0); self.bb.reset()
            return self

    outs = replay_bb_real_retune(m, golden.load("g10_real_in"), Node)
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load("g16_bb_real_retune_out"))


# ---- mid-stream retuning (src/baseband.hh:82-112): the reference node's setters between buffers -------------------

def replay_retune(m, x, make_node, demod=None, demod_reset=None):
    """Replays the manifest's event list on a node made by make_node(Ff, width, Fc): "feed n" / "center Fc" /
    "filter Ff width" / "reconfigure" (set_taps / set_shift / reset) and, g14, the events that change what a device plan
    is made for: "subsample D" / "orate Fs_out" / "bufsize n" (each runs _reconfigure: node.regeometry(D, bufsize), and
    the FMDemod behind the node is reset because the Config it receives changes) and "order n" (node.set_order(n):
    kernel and ring only). Returns the outputs per feed."""
    Fs, Fc, Ff, width = float(m["Fs"]), 100e3, 100e3, 50e3
    D, bufsize = m["decim"], 4096
    node, outs, off = None, [], 0
    for ev in m["events"]:
        if ev[0] == "feed":
            if node is None:
                node = make_node(Ff, width, Fc)
            y = node.process(x[off:off + ev[1]])
            outs.append(demod(y) if demod else y)
            off += ev[1]
        elif ev[0] == "center":
            Fc = float(int(ev[1]))
            node.set_shift_hz(Fc)
        elif ev[0] == "filter":
            Ff, width = float(int(ev[1])), float(int(ev[2]))
            node.set_filter(Ff, width)
        elif ev[0] == "reconfigure":
            node.set_filter(Ff, width); node.set_shift_hz(Fc); node.reconfigure()
        elif ev[0] in ("subsample", "orate", "bufsize"):
            if ev[0] == "subsample":
                D = int(ev[1])
            elif ev[0] == "orate":
                D = max(1, int(int(Fs) / float(ev[1])))   # src/baseband.hh:159-162
            else:
                bufsize = int(ev[1])
            node.regeometry(D, bufsize, Ff, width, Fc)
            if demod_reset:
                demod_reset()
        elif ev[0] == "order":
            node.set_order(int(ev[1]), Ff, width, Fc)
    return outs


def defined_mask(m):
    """True for every output of a g14 fixture that the reference defines (setOrder's new ring is uninitialised memory:
    the first `undefined_head` outputs of the feed behind it are not)."""
    lens, ok = m["out_lens"], np.ones(sum(m["out_lens"]), bool)
    feed = -1
    for ev in m["events"]:
        if ev[0] == "feed":
            feed += 1
        elif ev[0] == "order":
            a = sum(lens[:feed + 1])
            ok[a:a + m["undefined_head"]] = False
    return ok


class _OrcRetune:
    def __init__(self, orc, Ff, width, Fc, order=127, D=8):
        self.orc, self.order = orc, order
        self.bb = orc.IQBaseBandI16(orc.iqbb_design(Ff, width, FS, order), orc.freqshift_lut_i16(), orc.freqshift_inc(Fc, FS), Fc < 0, D)

    def process(self, x):
        return self.bb.process(x)

    def set_shift_hz(self, Fc):
        self.bb.set_shift(self.orc.freqshift_inc(Fc, FS), Fc < 0)

    def set_filter(self, Ff, width):
        self.bb.set_taps(self.orc.iqbb_design(Ff, width, FS, self.order))

    def reconfigure(self):
        self.bb.reset()

    def regeometry(self, D, bufsize, Ff, width, Fc):   # _reconfigure: kernel, LUT increment, counters; the ring stays
        self.bb.set_decim(D)
        self.set_filter(Ff, width); self.set_shift_hz(Fc); self.bb.reset()

    def set_order(self, order, Ff, width, Fc):
        self.order = order
        self.bb.set_order(self.orc.iqbb_design(Ff, width, FS, order))


def test_iqbb_regeometry_midstream(golden, orc):
    """g14: setSubsample / setOutputSampleRate / a new source Config (all _reconfigure, ring kept) and setOrder."""
    x = golden.load("g1_iq_cs16")
    m = golden.meta("g14_regeom_out")
    ok = defined_mask(m)
    assert ok.sum() == len(ok) - m["undefined_head"]
    outs = replay_retune(m, x, lambda Ff, w, Fc: _OrcRetune(orc, Ff, w, Fc))
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs)[ok], golden.load("g14_regeom_out").reshape(-1, 2)[ok])
    fm = [orc.FMDemodI16()]
    def fm_reset():
        fm[0] = orc.FMDemodI16()
    m = golden.meta("g14_regeom_fm")
    outs = replay_retune(m, x, lambda Ff, w, Fc: _OrcRetune(orc, Ff, w, Fc), demod=lambda y: fm[0].process(y), demod_reset=fm_reset)
    assert np.array_equal(np.concatenate(outs)[ok], golden.load("g14_regeom_fm")[ok])


def test_iqbb_retune_midstream(golden, orc):
    m = golden.meta("g12_retune_out")
    outs = replay_retune(m, golden.load("g1_iq_cs16"), lambda Ff, w, Fc: _OrcRetune(orc, Ff, w, Fc))
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs), golden.load("g12_retune_out"))
    fm = orc.FMDemodI16()
    outs = replay_retune(golden.meta("g12_retune_fm"), golden.load("g1_iq_cs16"), lambda Ff, w, Fc: _OrcRetune(orc, Ff, w, Fc),
                         demod=lambda y: fm.process(y))
    assert np.array_equal(np.concatenate(outs), golden.load("g12_retune_fm"))


# ---- the int8 chain (src/sdr.hh:225-240): IQBaseBand<int8_t> -> FMDemod<int8_t,int16_t> -------------

I8_CASES = ["g13_i8_o21_d8", "g13_i8_doc_o16", "g13_i8_neg_o33_d5"]


@pytest.mark.parametrize("case", I8_CASES)
def test_iqbb_i8_chain(golden, orc, case):
    m = golden.meta(case + "_out")
    x = golden.load("g13_iq_cs8").reshape(-1, 2)
    Fs = float(int(m["Fs"]))   # (the node stores Fs, Fc, Ff, width as int32)
    mk = lambda: orc.IQBaseBandI8(orc.iqbb_design(m["Ff"], m["width"], Fs, m["order"]), orc.freqshift_lut_i8(),
                                  orc.freqshift_inc(m["Fc"], Fs), m["Fc"] < 0, m["decim"])
    bb = mk()
    outs = [bb.process(c) for c in split(x, m["in_lens"])]
    assert [len(o) for o in outs] == m["out_lens"]
    assert np.array_equal(np.concatenate(outs).ravel(), golden.load(case + "_out"))
    bb, fm = mk(), orc.FMDemodI8()
    outs = [fm.process(bb.process(c)) for c in split(x, m["in_lens"])]
    assert np.array_equal(np.concatenate(outs), golden.load(case + "_fm"))


def test_fir_setfreq_midstream(golden, orc):
    """FIRLowPass::setFreq between buffers only recomputes the coefficients; the ring goes on (src/firfilter.hh:165-170).
    The oracle has no setter: a fresh filter with the new coefficients, primed with the `order` samples in front of the
    switch, continues the reference's stream (g17) bit for bit — the FIR's output depends on the last `order` samples only."""
    a100, a40 = orc.fir_lowpass_design(127, 100e3, 2.4e6), orc.fir_lowpass_design(127, 40e3, 2.4e6)
    x, ref = golden.load("g1_iq_cs16"), golden.load("g17_fir127_setfreq_cs16")
    sw = golden.meta("g17_fir127_setfreq_cs16")["switch_after_buffers"] * 4096
    f1, f2 = orc.FIR(a100), orc.FIR(a40)
    y1 = f1.process_cs16(x[:sw])
    f2.process_cs16(x[sw - 127:sw])
    assert np.array_equal(np.concatenate([y1, f2.process_cs16(x[sw:])]), ref)
    xf, reff = golden.load("g1_iq_cf32"), golden.load("g17_fir127_setfreq_cf32")
    sw = golden.meta("g17_fir127_setfreq_cf32")["switch_after_buffers"] * 4096
    f1, f2 = orc.FIR(a100), orc.FIR(a40)
    y1 = f1.process_cf32(xf[:sw])
    f2.process_cf32(xf[sw - 127:sw])
    assert np.array_equal(np.concatenate([y1, f2.process_cf32(xf[sw:])]), reff)
