"""ctypes loader for oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (libsdr_amd/) must never import it (tests/test_no_oracle_in_product.py checks).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile the restatement (g++, seconds). Also builds oracle/_ref when /root/reference exists."""
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ("sdr_oracle.cc", "sdr_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i32p, i16p, f32p, f64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int16), C.POINTER(C.c_float), C.POINTER(C.c_double)
        sig = {
            "orc_iqbb_design": (None, [C.c_double, C.c_double, C.c_double, C.c_int, i32p]),
            "orc_iqbb_decim": (C.c_int, [C.c_double, C.c_int, C.c_double]),
            "orc_freqshift_lut_i16": (None, [i32p]),
            "orc_freqshift_inc": (C.c_uint32, [C.c_double, C.c_double]),
            "orc_fir_lowpass_design": (None, [C.c_int, C.c_double, C.c_double, f64p]),
            "orc_fftfilt_design_h": (None, [C.c_int, C.c_double, C.c_double, C.c_double, f32p]),
            "orc_fftfilt_design_K": (None, [C.c_int, f32p, f32p]),
            "orc_iqsiggen_create": (vp, [C.c_double]),
            "orc_iqsiggen_add_sine": (None, [vp, C.c_double, C.c_double, C.c_double]),
            "orc_iqsiggen_next_cs16": (None, [vp, C.c_size_t, i16p]),
            "orc_iqsiggen_next_cf32": (None, [vp, C.c_size_t, f32p]),
            "orc_iqsiggen_destroy": (None, [vp]),
            "orc_iqbb_i16_create": (vp, [i32p, C.c_int, i32p, C.c_uint32, C.c_int, C.c_int]),
            "orc_iqbb_i16_process": (C.c_size_t, [vp, i16p, C.c_size_t, i16p]),
            "orc_iqbb_i16_reset": (None, [vp]),
            "orc_iqbb_i16_seek": (C.c_int, [vp, C.c_uint64]),
            "orc_bb_i16_seek": (C.c_int, [vp, C.c_uint64]),
            "orc_freqshift_lut_i8": (None, [C.POINTER(C.c_int32)]),
            "orc_iqbb_i8_create": (vp, [C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.c_uint32, C.c_int, C.c_int]),
            "orc_iqbb_i8_process": (C.c_size_t, [vp, C.POINTER(C.c_int8), C.c_size_t, C.POINTER(C.c_int8)]),
            "orc_iqbb_i8_destroy": (None, [vp]),
            "orc_fm_i8": (None, [C.POINTER(C.c_int8), C.c_size_t, C.POINTER(C.c_int16), C.POINTER(C.c_int16)]),
            "orc_iqbb_i16_set_taps": (None, [vp, C.POINTER(C.c_int32)]),
            "orc_iqbb_i16_set_decim": (None, [vp, C.c_int]),
            "orc_iqbb_i16_set_order": (None, [vp, C.POINTER(C.c_int32), C.c_int]),
            "orc_iqbb_i16_set_shift": (None, [vp, C.c_uint32, C.c_int]),
            "orc_iqbb_i16_destroy": (None, [vp]),
            "orc_bb_design": (None, [C.c_double, C.c_double, C.c_double, C.c_int, i32p]),
            "orc_bb_i16_create": (vp, [i32p, C.c_int, i32p, C.c_uint32, C.c_int, C.c_int]),
            "orc_bb_i16_process": (C.c_size_t, [vp, i16p, C.c_size_t, i16p]),
            "orc_bb_i16_reset": (None, [vp]),
            "orc_bb_i16_set_shift": (None, [vp, C.c_uint32, C.c_int]),
            "orc_bb_i16_destroy": (None, [vp]),
            "orc_fir_create": (vp, [f64p, C.c_int]),
            "orc_fir_cs16_process": (None, [vp, i16p, C.c_size_t, i16p]),
            "orc_fir_cf32_process": (None, [vp, f32p, C.c_size_t, f32p]),
            "orc_fir_reset": (None, [vp]),
            "orc_fir_destroy": (None, [vp]),
            "orc_fast_atan2_i16": (C.c_int16, [C.c_int16, C.c_int16]),
            "orc_fm_i16": (None, [i16p, C.c_size_t, i16p, i16p]),
            "orc_am_i16": (None, [i16p, C.c_size_t, i16p]),
            "orc_am_f32": (None, [f32p, C.c_size_t, f32p]),
            "orc_usb_i16": (None, [i16p, C.c_size_t, i16p]),
            "orc_usb_f32": (None, [f32p, C.c_size_t, f32p]),
            "orc_subsample_create": (vp, [C.c_size_t]),
            "orc_subsample_cs16_process": (C.c_size_t, [vp, i16p, C.c_size_t, i16p]),
            "orc_subsample_cf32_process": (C.c_size_t, [vp, f32p, C.c_size_t, f32p]),
            "orc_subsample_destroy": (None, [vp]),
            "orc_fftfilt_create": (vp, [C.c_int, f32p]),
            "orc_fftfilt_design_h_f64": (None, [C.c_int, C.c_double, C.c_double, C.c_double, f64p]),
            "orc_fftfilt_design_K_f64": (None, [C.c_int, f64p, f64p]),
            "orc_fftfilt_process_f64": (None, [C.c_int, f64p, f64p, f64p, f64p]),
            "orc_fftfilt_process": (None, [vp, f32p, f32p]),
            "orc_fftfilt_destroy": (None, [vp]),
            "orc_dft_f64": (None, [C.c_int, C.c_int, f64p, f64p]),
            "orc_freqshift_cf32": (None, [f32p, C.c_size_t, C.c_uint64, C.c_double, C.c_double, f32p]),
            "orc_autocast_cu8_cs16": (None, [C.POINTER(C.c_uint8), C.c_size_t, i16p]),
            "orc_fmdeemph_alpha": (C.c_int, [C.c_double]),
            "orc_fmdeemph_i16": (None, [i16p, C.c_size_t, C.c_int, i16p, i16p]),
            "orc_bench_iqbb_fm": (C.c_double, [i32p, C.c_int, i32p, C.c_uint32, C.c_int, C.c_int, i16p,
                                               C.c_size_t, C.c_size_t, C.POINTER(C.c_long)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


# ---- numpy-level helpers ---------------------------------------------------------------------

def iqbb_design(Ff, width, Fs, order):
    t = np.zeros(2 * order, np.int32)
    lib().orc_iqbb_design(Ff, width, Fs, order, _p(t, C.c_int32))
    return t.reshape(order, 2)


def freqshift_lut_i16():
    t = np.zeros(256, np.int32)
    lib().orc_freqshift_lut_i16(_p(t, C.c_int32))
    return t.reshape(128, 2)


def freqshift_inc(F, Fs):
    return int(lib().orc_freqshift_inc(F, Fs))


def iqbb_decim(Fs, sub, oFs=0.0):
    return int(lib().orc_iqbb_decim(Fs, sub, oFs))


def fir_lowpass_design(N, Fu, Fs):
    a = np.zeros(N, np.float64)
    lib().orc_fir_lowpass_design(N, Fu, Fs, _p(a, C.c_double))
    return a


def fftfilt_design_h(N, fmin, fmax, Fs):
    h = np.zeros(2 * N, np.float32)
    lib().orc_fftfilt_design_h(N, fmin, fmax, Fs, _p(h, C.c_float))
    return h.reshape(N, 2)


def fftfilt_design_h_f64(N, fmin, fmax, Fs):
    """sinc_flt_kernel<double>."""
    h = np.zeros(2 * N, np.float64)
    lib().orc_fftfilt_design_h_f64(N, fmin, fmax, Fs, _p(h, C.c_double))
    return h.reshape(N, 2)


def fftfilt_design_K_f64(h):
    h = np.ascontiguousarray(h, np.float64).reshape(-1, 2)
    K = np.zeros(4 * h.shape[0], np.float64)
    lib().orc_fftfilt_design_K_f64(h.shape[0], _p(h, C.c_double), _p(K, C.c_double))
    return K.reshape(2 * h.shape[0], 2)


class FFTFilterF64:
    """FilterSink<double> + FilterSource<double> (overlap-add, FFT size 2N). PARITY UNPINNED (FFTW)."""

    def __init__(self, K):
        self.K = np.ascontiguousarray(K, np.float64).reshape(-1, 2)
        self.N = self.K.shape[0] // 2
        self.last = np.zeros((self.N, 2), np.float64)

    def process(self, x):
        x = np.ascontiguousarray(x, np.float64).reshape(self.N, 2)
        o = np.zeros_like(x)
        lib().orc_fftfilt_process_f64(self.N, _p(self.K, C.c_double), _p(self.last, C.c_double), _p(x, C.c_double), _p(o, C.c_double))
        return o


def fftfilt_design_K(h):
    h = np.ascontiguousarray(h, np.float32).reshape(-1, 2)
    N = h.shape[0]
    K = np.zeros(4 * N, np.float32)
    lib().orc_fftfilt_design_K(N, _p(h, C.c_float), _p(K, C.c_float))
    return K.reshape(2 * N, 2)


class IQSigGen:
    """IQSigGen<T> restatement (src/siggen.hh:90-157)."""

    def __init__(self, Fs, tones):
        self._h = lib().orc_iqsiggen_create(Fs)
        for f, a, p in tones:
            lib().orc_iqsiggen_add_sine(self._h, f, a, p)

    def next_cs16(self, n):
        o = np.zeros((n, 2), np.int16)
        lib().orc_iqsiggen_next_cs16(self._h, n, _p(o, C.c_int16))
        return o

    def next_cf32(self, n):
        o = np.zeros((n, 2), np.float32)
        lib().orc_iqsiggen_next_cf32(self._h, n, _p(o, C.c_float))
        return o

    def __del__(self):
        if lib is not None and self._h:
            lib().orc_iqsiggen_destroy(self._h)
            self._h = None


class IQBaseBandI16:
    def __init__(self, taps, lut, lut_inc, negative, decim):
        taps = np.ascontiguousarray(taps, np.int32).reshape(-1, 2)
        lut = np.ascontiguousarray(lut, np.int32).reshape(128, 2)
        self.order, self.decim = taps.shape[0], decim
        self._h = lib().orc_iqbb_i16_create(_p(taps, C.c_int32), self.order, _p(lut, C.c_int32),
                                            lut_inc, int(negative), decim)

    def process(self, x):
        x = np.ascontiguousarray(x, np.int16).reshape(-1, 2)
        out = np.zeros((x.shape[0] // max(self.decim, 1) + 2, 2), np.int16)
        n = lib().orc_iqbb_i16_process(self._h, _p(x, C.c_int16), x.shape[0], _p(out, C.c_int16))
        return out[:n].copy()

    def reset(self):
        lib().orc_iqbb_i16_reset(self._h)

    def seek(self, abs_index):
        """Test-bench helper: decimator and LUT phase as they stand in front of absolute sample `abs_index` = g*D + 1 (right
        behind an emission); the ring is kept — prime it with the `order` samples before that index first."""
        if lib().orc_iqbb_i16_seek(self._h, int(abs_index)) != 0:
            raise ValueError("seek: %d is not right behind an emission (g*D + 1, g >= 1)" % abs_index)

    def set_decim(self, decim):
        """setSubsample's new decimation; follow with reset() (= _reconfigure)."""
        self.decim = int(decim)
        lib().orc_iqbb_i16_set_decim(self._h, self.decim)

    def set_order(self, taps):
        """setOrder: new kernel and new (zeroed) ring; decimator, counters and LUT phase go on."""
        taps = np.ascontiguousarray(taps, np.int32).reshape(-1, 2)
        self.order = taps.shape[0]
        lib().orc_iqbb_i16_set_order(self._h, _p(taps, C.c_int32), self.order)

    def set_taps(self, taps):
        taps = np.ascontiguousarray(taps, np.int32).reshape(self.order, 2)
        lib().orc_iqbb_i16_set_taps(self._h, _p(taps, C.c_int32))

    def set_shift(self, lut_inc, negative):
        lib().orc_iqbb_i16_set_shift(self._h, lut_inc, int(negative))

    def __del__(self):
        if self._h:
            lib().orc_iqbb_i16_destroy(self._h)
            self._h = None


def freqshift_lut_i8():
    t = np.zeros(256, np.int32)
    lib().orc_freqshift_lut_i8(_p(t, C.c_int32))
    return t.reshape(128, 2)


class IQBaseBandI8:
    """IQBaseBand<int8_t> (compute type int16): complex<int8> in, complex<int8> out."""

    def __init__(self, taps, lut, lut_inc, negative, decim):
        taps = np.ascontiguousarray(taps, np.int32).reshape(-1, 2)
        lut = np.ascontiguousarray(lut, np.int32).reshape(128, 2)
        self.order, self.decim = taps.shape[0], decim
        self._h = lib().orc_iqbb_i8_create(_p(taps, C.c_int32), self.order, _p(lut, C.c_int32), lut_inc, int(negative), decim)

    def process(self, x):
        x = np.ascontiguousarray(x, np.int8).reshape(-1, 2)
        out = np.zeros((x.shape[0] // max(self.decim, 1) + 2, 2), np.int8)
        n = lib().orc_iqbb_i8_process(self._h, _p(x, C.c_int8), x.shape[0], _p(out, C.c_int8))
        return out[:n].copy()

    def __del__(self):
        if self._h:
            lib().orc_iqbb_i8_destroy(self._h)
            self._h = None


class FMDemodI8:
    """FMDemod<int8_t,int16_t> run in place: out[0] = the two bytes of in[0]."""

    def __init__(self):
        self.last = C.c_int16(0)

    def process(self, y):
        y = np.ascontiguousarray(y, np.int8).reshape(-1, 2)
        out = y.copy().view(np.int16).reshape(-1)
        if len(y):
            lib().orc_fm_i8(_p(y, C.c_int8), y.shape[0], _p(out, C.c_int16), C.byref(self.last))
        return out


def bb_design(Ff, width, Fs, order):
    t = np.zeros(2 * order, np.int32)
    lib().orc_bb_design(Ff, width, Fs, order, _p(t, C.c_int32))
    return t.reshape(-1, 2)


class BaseBandI16:
    """BaseBand<int16_t>, the real-input variant: int16 samples in, cs16 out."""

    def __init__(self, taps, lut, lut_inc, negative, decim):
        taps = np.ascontiguousarray(taps, np.int32).reshape(-1, 2)
        lut = np.ascontiguousarray(lut, np.int32).reshape(128, 2)
        self.order, self.decim = taps.shape[0], decim
        self._h = lib().orc_bb_i16_create(_p(taps, C.c_int32), self.order, _p(lut, C.c_int32),
                                          lut_inc, int(negative), decim)

    def process(self, x):
        x = np.ascontiguousarray(x, np.int16).reshape(-1)
        out = np.zeros((x.shape[0] // max(self.decim, 1) + 2, 2), np.int16)
        n = lib().orc_bb_i16_process(self._h, _p(x, C.c_int16), x.shape[0], _p(out, C.c_int16))
        return out[:n].copy()

    def reset(self):
        lib().orc_bb_i16_reset(self._h)

    def seek(self, abs_index):
        """Test-bench helper: decimator and LUT phase as they stand in front of absolute sample `abs_index` = g*D; the ring
        is kept — prime it with the `order` samples before that index first."""
        if lib().orc_bb_i16_seek(self._h, int(abs_index)) != 0:
            raise ValueError("seek: %d is not a group boundary (g*D)" % abs_index)

    def set_shift(self, lut_inc, negative):
        lib().orc_bb_i16_set_shift(self._h, lut_inc, int(negative))

    def __del__(self):
        if self._h:
            lib().orc_bb_i16_destroy(self._h)
            self._h = None


class FIR:
    def __init__(self, alpha):
        alpha = np.ascontiguousarray(alpha, np.float64)
        self._h = lib().orc_fir_create(_p(alpha, C.c_double), alpha.shape[0])

    def process_cs16(self, x):
        x = np.ascontiguousarray(x, np.int16).reshape(-1, 2)
        o = np.zeros_like(x)
        lib().orc_fir_cs16_process(self._h, _p(x, C.c_int16), x.shape[0], _p(o, C.c_int16))
        return o

    def process_cf32(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 2)
        o = np.zeros_like(x)
        lib().orc_fir_cf32_process(self._h, _p(x, C.c_float), x.shape[0], _p(o, C.c_float))
        return o

    def reset(self):
        lib().orc_fir_reset(self._h)

    def __del__(self):
        if self._h:
            lib().orc_fir_destroy(self._h)
            self._h = None


class FMDemodI16:
    """FMDemod<int16_t> with the in-place convention of the north-star chain: out[0] of every
    buffer is the low int16 of in[0], i.e. in[0].real()."""

    def __init__(self):
        self.last = np.zeros(1, np.int16)

    def process(self, y, inplace=True):
        y = np.ascontiguousarray(y, np.int16).reshape(-1, 2)
        n = y.shape[0]
        if n == 0:
            return np.zeros(0, np.int16)
        out = np.zeros(n, np.int16)
        lib().orc_fm_i16(_p(y, C.c_int16), n, _p(out, C.c_int16), _p(self.last, C.c_int16))
        out[0] = y[0, 0] if inplace else 0
        return out


def am_i16(y):
    y = np.ascontiguousarray(y, np.int16).reshape(-1, 2)
    o = np.zeros(y.shape[0], np.int16)
    lib().orc_am_i16(_p(y, C.c_int16), y.shape[0], _p(o, C.c_int16))
    return o


def usb_i16(y):
    y = np.ascontiguousarray(y, np.int16).reshape(-1, 2)
    o = np.zeros(y.shape[0], np.int16)
    lib().orc_usb_i16(_p(y, C.c_int16), y.shape[0], _p(o, C.c_int16))
    return o


def am_f32(y):
    y = np.ascontiguousarray(y, np.float32).reshape(-1, 2)
    o = np.zeros(y.shape[0], np.float32)
    lib().orc_am_f32(_p(y, C.c_float), y.shape[0], _p(o, C.c_float))
    return o


def usb_f32(y):
    y = np.ascontiguousarray(y, np.float32).reshape(-1, 2)
    o = np.zeros(y.shape[0], np.float32)
    lib().orc_usb_f32(_p(y, C.c_float), y.shape[0], _p(o, C.c_float))
    return o


def fast_atan2_i16(a, b):
    return int(lib().orc_fast_atan2_i16(int(a), int(b)))


class SubSample:
    def __init__(self, n):
        self.n = n
        self._h = lib().orc_subsample_create(n)

    def process_cs16(self, x):
        x = np.ascontiguousarray(x, np.int16).reshape(-1, 2)
        o = np.zeros((x.shape[0] // self.n + 2, 2), np.int16)
        m = lib().orc_subsample_cs16_process(self._h, _p(x, C.c_int16), x.shape[0], _p(o, C.c_int16))
        return o[:m].copy()

    def process_cf32(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 2)
        o = np.zeros((x.shape[0] // self.n + 2, 2), np.float32)
        m = lib().orc_subsample_cf32_process(self._h, _p(x, C.c_float), x.shape[0], _p(o, C.c_float))
        return o[:m].copy()

    def __del__(self):
        if self._h:
            lib().orc_subsample_destroy(self._h)
            self._h = None


class FFTFilter:
    """FilterSink + FilterSource (overlap-add, FFT size 2N, N taps). PARITY UNPINNED (FFTW)."""

    def __init__(self, K):
        K = np.ascontiguousarray(K, np.float32).reshape(-1, 2)
        self.N = K.shape[0] // 2
        self._h = lib().orc_fftfilt_create(self.N, _p(K, C.c_float))

    def process(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(self.N, 2)
        o = np.zeros_like(x)
        lib().orc_fftfilt_process(self._h, _p(x, C.c_float), _p(o, C.c_float))
        return o

    def __del__(self):
        if self._h:
            lib().orc_fftfilt_destroy(self._h)
            self._h = None


def freqshift_cf32(x, n0, Fc, Fs):
    x = np.ascontiguousarray(x, np.float32).reshape(-1, 2)
    o = np.zeros_like(x)
    lib().orc_freqshift_cf32(_p(x, C.c_float), x.shape[0], n0, Fc, Fs, _p(o, C.c_float))
    return o


def autocast_cu8_cs16(x):
    """AutoCast<complex<int16>> on complex<uint8> input; x: [..., 2] uint8 -> [..., 2] int16."""
    x = np.ascontiguousarray(x, np.uint8)
    o = np.zeros(x.shape, np.int16)
    lib().orc_autocast_cu8_cs16(_p(x, C.c_uint8), x.size, _p(o, C.c_int16))
    return o


class FMDeemphI16:
    def __init__(self, sample_rate):
        self.alpha = int(lib().orc_fmdeemph_alpha(sample_rate))
        self.avg = np.zeros(1, np.int16)

    def process(self, x):
        x = np.ascontiguousarray(x, np.int16)
        o = np.zeros_like(x)
        lib().orc_fmdeemph_i16(_p(x, C.c_int16), x.size, self.alpha, _p(self.avg, C.c_int16), _p(o, C.c_int16))
        return o


def bench_iqbb_fm(taps, lut, lut_inc, negative, decim, x, nbuf):
    taps = np.ascontiguousarray(taps, np.int32).reshape(-1, 2)
    lut = np.ascontiguousarray(lut, np.int32).reshape(128, 2)
    x = np.ascontiguousarray(x, np.int16).reshape(-1, 2)
    cs = C.c_long(0)
    sec = lib().orc_bench_iqbb_fm(_p(taps, C.c_int32), taps.shape[0], _p(lut, C.c_int32), lut_inc,
                                  int(negative), decim, _p(x, C.c_int16), x.shape[0], nbuf, C.byref(cs))
    return sec
