"""Thin numpy-facing wrappers over the C ABI (libsdr_amd.abi) — the Python mirror of the C++ nodes in
include/sdr/gpu/*.hh, used by tests/ and bench.py.

Every class maps 1:1 to a handle type of include/sdrhip.h; `process(x)` takes/returns numpy arrays in
the channel-major layout [channels, n, 2] (complex as (re, im)), `process_dev(ptr, ...)` takes raw
device pointers (e.g. torch tensors' data_ptr()). No CPU fallback exists: constructing a Context
without a HIP device raises SdrHipError(E_NODEVICE).
"""
import ctypes as C

import numpy as np

from . import abi
from .abi import (EPI_NONE, EPI_FM, EPI_AM, EPI_USB, FIR_CS16_EXACT, FIR_CF32, T_CS16, T_CF32,
                  FFTCONV_OLA, FFTCONV_OLS, check)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- designers (host only; identical code to include/sdr/gpu/design.hh) -----------------------

def design_iqbb_taps(Ff, width, Fs, order):
    t = np.zeros((order, 2), np.int32)
    check(abi.lib().sdrhip_design_iqbb_taps(Ff, width, Fs, order, t.ctypes.data_as(C.POINTER(C.c_int32))))
    return t


def design_bb_taps(Ff, width, Fs, order):
    """Q16 taps of the real-input BaseBand<int16_t> (reference src/baseband.hh:464-491)."""
    t = np.zeros((order, 2), np.int32)
    check(abi.lib().sdrhip_design_bb_taps(Ff, width, Fs, order, t.ctypes.data_as(C.POINTER(C.c_int32))))
    return t


def design_iqbb_decim(Fs, sub, oFs=0.0):
    d = C.c_int(0)
    check(abi.lib().sdrhip_design_iqbb_decim(Fs, sub, oFs, C.byref(d)))
    return d.value


def design_freqshift_lut_i16():
    t = np.zeros((128, 2), np.int32)
    check(abi.lib().sdrhip_design_freqshift_lut_i16(t.ctypes.data_as(C.POINTER(C.c_int32))))
    return t


def design_freqshift_inc(F, Fs):
    v = C.c_uint32(0)
    check(abi.lib().sdrhip_design_freqshift_inc(F, Fs, C.byref(v)))
    return v.value


def design_fir_lowpass(order, Fu, Fs):
    a = np.zeros(order, np.float64)
    check(abi.lib().sdrhip_design_fir_lowpass(order, Fu, Fs, a.ctypes.data_as(C.POINTER(C.c_double))))
    return a


def design_fmdeemph_alpha(Fs):
    v = C.c_int(0)
    check(abi.lib().sdrhip_design_fmdeemph_alpha(Fs, C.byref(v)))
    return v.value


def design_fftfilt_kernel(N, fmin, fmax, Fs, dtype=np.float32):
    """sinc_flt_kernel<float> (default) or <double> (dtype=np.float64): N x (re, im)."""
    h = np.zeros((N, 2), dtype)
    if np.dtype(dtype) == np.float64:
        check(abi.lib().sdrhip_design_fftfilt_kernel_f64(N, fmin, fmax, Fs, h.ctypes.data_as(C.POINTER(C.c_double))))
    else:
        check(abi.lib().sdrhip_design_fftfilt_kernel(N, fmin, fmax, Fs, h.ctypes.data_as(C.POINTER(C.c_float))))
    return h


def design_fftfilt_spectrum(h):
    """FilterSource::_updateFilter: K = DFT_2N([h, 0]) / ||K||; the dtype follows h (float32 unless h is float64)."""
    f64 = np.asarray(h).dtype == np.float64
    h = np.ascontiguousarray(h, np.float64 if f64 else np.float32).reshape(-1, 2)
    K = np.zeros((2 * h.shape[0], 2), h.dtype)
    if f64:
        check(abi.lib().sdrhip_design_fftfilt_spectrum_f64(h.shape[0], h.ctypes.data_as(C.POINTER(C.c_double)),
                                                           K.ctypes.data_as(C.POINTER(C.c_double))))
    else:
        check(abi.lib().sdrhip_design_fftfilt_spectrum(h.shape[0], h.ctypes.data_as(C.POINTER(C.c_float)),
                                                       K.ctypes.data_as(C.POINTER(C.c_float))))
    return K


# ---- context ------------------------------------------------------------------------------------

def device_count():
    n = C.c_int(0)
    check(abi.lib().sdrhip_device_count(C.byref(n)))
    return n.value


class Context:
    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        check(abi.lib().sdrhip_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.device = device

    @classmethod
    def borrowed(cls, handle, device):
        """A context some other object owns (sdrhip_comm_ctx: rank r's context lives as long as the comm)."""
        self = cls.__new__(cls)
        self._h, self.device, self._borrowed = C.c_void_p(handle.value if isinstance(handle, C.c_void_p) else handle), device, True
        return self

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        check(abi.lib().sdrhip_ctx_synchronize(self._h))

    def device_name(self):
        b = C.create_string_buffer(256)
        check(abi.lib().sdrhip_ctx_device_name(self._h, b, 256))
        return b.value.decode()

    def malloc(self, nbytes):
        p = C.c_void_p()
        check(abi.lib().sdrhip_malloc(self._h, nbytes, C.byref(p)))
        return p.value

    def free(self, p):
        check(abi.lib().sdrhip_free(self._h, C.c_void_p(p)))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        check(abi.lib().sdrhip_memcpy_h2d(self._h, C.c_void_p(dptr), _ptr(arr), arr.nbytes))

    def d2h(self, arr, dptr):
        assert arr.flags["C_CONTIGUOUS"]
        check(abi.lib().sdrhip_memcpy_d2h(self._h, _ptr(arr), C.c_void_p(dptr), arr.nbytes))

    def memset(self, dptr, value, nbytes):
        check(abi.lib().sdrhip_memset(self._h, C.c_void_p(dptr), value, nbytes))

    def close(self):
        if self._h:
            for hook in close_hooks:   # (buffers a device_router holds on this context)
                hook(self)
            if not getattr(self, "_borrowed", False):
                abi.lib().sdrhip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Timer:
    """HIP events on the context's stream."""

    def __init__(self, ctx):
        self._h = C.c_void_p()
        check(abi.lib().sdrhip_timer_create(ctx.handle, C.byref(self._h)))

    def start(self):
        check(abi.lib().sdrhip_timer_start(self._h))

    def stop(self):
        check(abi.lib().sdrhip_timer_stop(self._h))

    def elapsed_ms(self):
        ms = C.c_float(0)
        check(abi.lib().sdrhip_timer_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def __del__(self):
        if self._h:
            abi.lib().sdrhip_timer_destroy(self._h)
            self._h = C.c_void_p()


class _Node:
    _destroy = None

    def __init__(self):
        self._h = C.c_void_p()

    def close(self):
        if self._h:
            getattr(abi.lib(), self._destroy)(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _as3(x, dtype, comps=2):
    x = np.ascontiguousarray(x, dtype)
    if x.ndim == 2:
        x = x[None]
    assert x.ndim == 3 and x.shape[2] == comps, x.shape
    return x


# Seam for callers that bring their own device buffers to process(): when `device_router` is set, process(x) hands
# (ctx, x, out, call) to it instead of using the library's host-pointer entry point; the router stages x / out in device
# memory of its choosing and runs `call(in_ptr, in_stride, out_ptr, out_stride)` (the node's *_process_dev entry point,
# strides in row elements). None: the host-pointer path. `close_hooks` run before a context is destroyed.
# (tests/redzone.py installs a red-zoned arena here: guard bands around every row, checked after every call.)
device_router = None
close_hooks = []


class IQBaseBandI16(_Node):
    """K1 — IQBaseBand<int16_t> (+ fused FM/AM/USB). Mirrors sdr::gpu::IQBaseBand<int16_t>."""
    _destroy = "sdrhip_iqbb_i16_destroy"

    def __init__(self, ctx, taps, lut, lut_inc, negative, decim, channels=1, max_in=65536, epilogue=EPI_NONE):
        super().__init__()
        taps = np.ascontiguousarray(taps, np.int32).reshape(-1, 2)
        lut = np.ascontiguousarray(lut, np.int32).reshape(128, 2)
        self.ctx, self.channels, self.decim, self.epilogue, self.max_in = ctx, channels, decim, epilogue, max_in
        check(abi.lib().sdrhip_iqbb_i16_create(ctx.handle, taps.ctypes.data_as(C.POINTER(C.c_int32)), taps.shape[0],
                                               lut.ctypes.data_as(C.POINTER(C.c_int32)), lut_inc, int(bool(negative)),
                                               decim, channels, max_in, epilogue, C.byref(self._h)))

    @property
    def path(self):
        """0 = VALU dot2 kernel, 1 = int8-MFMA 32x32x32 (decim 8), 2 = int8-MFMA 16x16x64 (decim 8), 3 = int8-MFMA, any decim."""
        v = C.c_int(0)
        check(abi.lib().sdrhip_iqbb_i16_path(self._h, C.byref(v)))
        return v.value

    @property
    def kernel_names(self):
        """Kernels a call launches, dominant first (what to look for in a rocprofv3 kernel trace)."""
        b = C.create_string_buffer(256)
        check(abi.lib().sdrhip_iqbb_i16_kernel_names(self._h, b, 256))
        return b.value.decode().split(",")

    @property
    def plan_info(self):
        """{path, S, S0, NH, NW, kind, OP, HH} of the plan (sdrhip.h: sdrhip_iqbb_i16_plan_info)."""
        v = (C.c_int * 9)()
        check(abi.lib().sdrhip_iqbb_i16_plan_info(self._h, v, 9))
        return dict(zip(("path", "S", "S0", "NH", "NW", "kind", "OP", "HH", "multi_left"), list(v)))

    def out_count(self, n_in):
        n = C.c_size_t(0)
        check(abi.lib().sdrhip_iqbb_i16_out_count(self._h, n_in, C.byref(n)))
        return n.value

    def process(self, x):
        x = _as3(x, np.uint8 if getattr(self, "_cu8", False) else np.int16)
        assert x.shape[0] == self.channels
        n_in = x.shape[1]
        no = self.out_count(n_in)
        if self.epilogue == EPI_NONE:
            out = np.zeros((self.channels, no, 2), np.int16)
        else:
            out = np.zeros((self.channels, no), np.int16)
        if device_router is not None and n_in and no:
            return device_router(self.ctx, x, out, lambda i, si, o, so: self._dev_checked(i, n_in, si, o, so, no))
        got = C.c_size_t(0)
        check(abi.lib().sdrhip_iqbb_i16_process(self._h, _ptr(x), n_in, n_in, _ptr(out), no, C.byref(got)))
        assert got.value == no
        return out

    def _dev_checked(self, i, n_in, si, o, so, no):
        assert self.process_dev(i, n_in, si, o, so) == no

    def process_dev(self, in_ptr, n_in, in_stride, out_ptr, out_stride):
        got = C.c_size_t(0)
        check(abi.lib().sdrhip_iqbb_i16_process_dev(self._h, C.c_void_p(in_ptr), n_in, in_stride, C.c_void_p(out_ptr),
                                                    out_stride, C.byref(got)))
        return got.value

    def process_dev_multi(self, in_ptr, n_buffers, n_per_buffer, in_stride, out_ptr, out_stride):
        """n_buffers consecutive buffers per channel in ONE launch, buffer boundaries kept (sdrhip.h); returns the output
        counts per buffer (their outputs follow one another in each channel's row)."""
        counts = (C.c_size_t * max(1, n_buffers))()
        total = C.c_size_t(0)
        check(abi.lib().sdrhip_iqbb_i16_process_dev_multi(self._h, C.c_void_p(in_ptr), n_buffers, n_per_buffer, in_stride, C.c_void_p(out_ptr),
                                                          out_stride, counts, C.byref(total)))
        assert sum(counts[:n_buffers]) == total.value
        return list(counts[:n_buffers])

    def process_multi(self, x, n_buffers):
        """Host arrays through process_dev_multi: x = [channels, n_buffers * n_per_buffer(, 2)]; returns (rows, counts) — the
        concatenated outputs of the buffers per channel and the output count of each buffer."""
        real = x.ndim == 2 and not getattr(self, "_cu8", False) and isinstance(self, BaseBandI16)
        x = np.ascontiguousarray(x, np.int16) if real else _as3(x, np.uint8 if getattr(self, "_cu8", False) else np.int16)
        assert x.shape[0] == self.channels and x.shape[1] % n_buffers == 0
        n_in, nb = x.shape[1], x.shape[1] // n_buffers
        no = self.out_count(n_in)
        out = np.zeros((self.channels, no, 2) if self.epilogue == EPI_NONE else (self.channels, no), np.int16)
        counts = []
        run = lambda i, si, o, so: counts.extend(self.process_dev_multi(i, n_buffers, nb, si, o, so))
        if device_router is not None:
            device_router(self.ctx, x, out, run)
        else:
            din, dout = self.ctx.malloc(max(x.nbytes, 16)), self.ctx.malloc(max(out.nbytes, 16))
            try:
                self.ctx.h2d(din, x); self.ctx.h2d(dout, out)
                run(din, n_in, dout, no)
                self.ctx.synchronize()
                self.ctx.d2h(out, dout)
            finally:
                self.ctx.free(din); self.ctx.free(dout)
        return out, counts

    def reset(self, keep_history=False, keep_fm=False):
        check(abi.lib().sdrhip_iqbb_i16_reset(self._h, int(bool(keep_history)) | (2 if keep_fm else 0)))

    def adopt_state(self, other, what):
        """Streaming state of `other` carried into this FRESH plan (abi.KEEP_RING | KEEP_FM | KEEP_COUNTERS): what the
        reference node keeps when a setter changes the geometry a device plan is made for."""
        check(abi.lib().sdrhip_iqbb_i16_adopt_state(self._h, other._h, int(what)))

    def set_taps(self, taps):
        """setFilterFrequency / setFilterWidth of the reference node: the kernel only."""
        taps = np.ascontiguousarray(taps, np.int32).reshape(-1, 2)
        check(abi.lib().sdrhip_iqbb_i16_set_taps(self._h, taps.ctypes.data_as(C.POINTER(C.c_int32))))

    def set_shift(self, lut_inc, negative):
        """setCenterFrequency of the reference node: increment, sign, LUT phase restarts."""
        check(abi.lib().sdrhip_iqbb_i16_set_shift(self._h, lut_inc, int(bool(negative))))

    def set_input_format(self, fmt):
        """abi.IN_CS16 (default) or abi.IN_CU8 (complex<uint8> buffers, AutoCast<cs16> fused into the load)."""
        check(abi.lib().sdrhip_iqbb_i16_set_input_format(self._h, fmt))
        self._cu8 = fmt == abi.IN_CU8


class BaseBandI16(IQBaseBandI16):
    """BaseBand<int16_t>, the real-input node (reference src/baseband.hh:305-529): int16 samples in, cs16 (or the
    demodulated int16) out. Shares the handle type and every call except create with IQBaseBandI16."""

    def __init__(self, ctx, taps, lut, lut_inc, negative, decim, channels=1, max_in=65536, epilogue=EPI_NONE):
        _Node.__init__(self)
        taps = np.ascontiguousarray(taps, np.int32).reshape(-1, 2)
        lut = np.ascontiguousarray(lut, np.int32).reshape(128, 2)
        self.ctx, self.channels, self.decim, self.epilogue, self.max_in = ctx, channels, decim, epilogue, max_in
        check(abi.lib().sdrhip_bb_i16_create(ctx.handle, taps.ctypes.data_as(C.POINTER(C.c_int32)), taps.shape[0],
                                             lut.ctypes.data_as(C.POINTER(C.c_int32)), lut_inc, int(bool(negative)),
                                             decim, channels, max_in, epilogue, C.byref(self._h)))

    def process(self, x):
        x = np.ascontiguousarray(x, np.int16)
        if x.ndim == 1:
            x = x[None, :]
        assert x.ndim == 2 and x.shape[0] == self.channels
        n_in = x.shape[1]
        no = self.out_count(n_in)
        out = np.zeros((self.channels, no, 2) if self.epilogue == EPI_NONE else (self.channels, no), np.int16)
        if device_router is not None and n_in and no:
            return device_router(self.ctx, x, out, lambda i, si, o, so: self._dev_checked(i, n_in, si, o, so, no))
        got = C.c_size_t(0)
        check(abi.lib().sdrhip_iqbb_i16_process(self._h, _ptr(x), n_in, n_in, _ptr(out), no, C.byref(got)))
        assert got.value == no
        return out


def design_freqshift_lut_i8():
    t = np.zeros((128, 2), np.int32)
    check(abi.lib().sdrhip_design_freqshift_lut_i8(t.ctypes.data_as(C.POINTER(C.c_int32))))
    return t


class IQBaseBandI8(IQBaseBandI16):
    """IQBaseBand<int8_t> (the documentation example's baseband, reference src/sdr.hh:225-240): complex<int8> in,
    complex<int8> out — or, with EPI_FM, FMDemod<int8_t,int16_t>'s int16. Same handle type as IQBaseBandI16."""

    def __init__(self, ctx, taps, lut, lut_inc, negative, decim, channels=1, max_in=65536, epilogue=EPI_NONE):
        _Node.__init__(self)
        taps = np.ascontiguousarray(taps, np.int32).reshape(-1, 2)
        lut = np.ascontiguousarray(lut, np.int32).reshape(128, 2)
        self.ctx, self.channels, self.decim, self.epilogue, self.max_in = ctx, channels, decim, epilogue, max_in
        check(abi.lib().sdrhip_iqbb_i8_create(ctx.handle, taps.ctypes.data_as(C.POINTER(C.c_int32)), taps.shape[0],
                                              lut.ctypes.data_as(C.POINTER(C.c_int32)), lut_inc, int(bool(negative)),
                                              decim, channels, max_in, epilogue, C.byref(self._h)))

    def process(self, x):
        x = _as3(x, np.int8)
        assert x.shape[0] == self.channels
        n_in = x.shape[1]
        no = self.out_count(n_in)
        out = np.zeros((self.channels, no, 2), np.int8) if self.epilogue == EPI_NONE else np.zeros((self.channels, no), np.int16)
        if device_router is not None and n_in and no:
            return device_router(self.ctx, x, out, lambda i, si, o, so: self._dev_checked(i, n_in, si, o, so, no))
        got = C.c_size_t(0)
        check(abi.lib().sdrhip_iqbb_i16_process(self._h, _ptr(x), n_in, n_in, _ptr(out), no, C.byref(got)))
        assert got.value == no
        return out


class FIR(_Node):
    """K2/K3 — FIRFilter<complex<int16>> exact / FIRFilter<complex<float>> (+ folded SubSample, + demod)."""
    _destroy = "sdrhip_fir_destroy"

    def __init__(self, ctx, kind, alpha, decim=1, channels=1, max_in=65536, epilogue=EPI_NONE):
        super().__init__()
        alpha = np.ascontiguousarray(alpha, np.float64)
        self.ctx, self.kind, self.channels, self.decim, self.epilogue = ctx, kind, channels, decim, epilogue
        self.order = alpha.shape[0]
        check(abi.lib().sdrhip_fir_create(ctx.handle, kind, alpha.ctypes.data_as(C.POINTER(C.c_double)), alpha.shape[0],
                                          decim, channels, max_in, epilogue, C.byref(self._h)))

    def kernel_names(self, n_in=0):
        """The kernel a call of n_in samples per channel runs (0: max_in) — what to look for in a rocprofv3 kernel trace."""
        b = C.create_string_buffer(256)
        check(abi.lib().sdrhip_fir_kernel_names(self._h, n_in, b, 256))
        return b.value.decode().split(",")

    def out_count(self, n_in):
        n = C.c_size_t(0)
        check(abi.lib().sdrhip_fir_out_count(self._h, n_in, C.byref(n)))
        return n.value

    def process(self, x):
        it = np.int16 if self.kind == FIR_CS16_EXACT else np.float32
        x = _as3(x, it)
        n_in = x.shape[1]
        no = self.out_count(n_in)
        out = np.zeros((self.channels, no, 2) if self.epilogue == EPI_NONE else (self.channels, no), it)
        if device_router is not None and n_in and no:
            def call(i, si, o, so):
                assert self.process_dev(i, n_in, si, o, so) == no
            return device_router(self.ctx, x, out, call)
        got = C.c_size_t(0)
        check(abi.lib().sdrhip_fir_process(self._h, _ptr(x), n_in, n_in, _ptr(out), no, C.byref(got)))
        assert got.value == no
        return out

    def process_dev(self, in_ptr, n_in, in_stride, out_ptr, out_stride):
        got = C.c_size_t(0)
        check(abi.lib().sdrhip_fir_process_dev(self._h, C.c_void_p(in_ptr), n_in, in_stride, C.c_void_p(out_ptr),
                                               out_stride, C.byref(got)))
        return got.value

    def reset(self):
        check(abi.lib().sdrhip_fir_reset(self._h))

    def set_taps(self, alpha):
        """New coefficients, same order: the ring (the stream) goes on (FIRFilter::setUpperFreq, src/firfilter.hh:165-170)."""
        alpha = np.ascontiguousarray(alpha, np.float64)
        assert alpha.shape == (self.order,)
        check(abi.lib().sdrhip_fir_set_taps(self._h, alpha.ctypes.data_as(C.POINTER(C.c_double))))


class Demod(_Node):
    """K4/K5 — stand-alone FMDemod<int16_t> / AMDemod / USBDemod."""
    _destroy = "sdrhip_demod_destroy"

    def __init__(self, ctx, kind, dtype=T_CS16, channels=1, max_in=65536, inplace_fm0=True):
        super().__init__()
        self.ctx, self.kind, self.dtype, self.channels = ctx, kind, dtype, channels
        check(abi.lib().sdrhip_demod_create(ctx.handle, kind, dtype, channels, max_in, int(inplace_fm0), C.byref(self._h)))

    def process(self, x, out=None):
        it = np.int16 if self.dtype == T_CS16 else np.int8 if self.dtype == abi.T_CS8 else np.float32
        x = _as3(x, it)
        n = x.shape[1]
        if out is None:
            out = np.zeros((self.channels, n), np.float32 if self.dtype == T_CF32 else np.int16)
            if device_router is not None and n:
                return device_router(self.ctx, x, out, lambda i, si, o, so: self.process_dev(i, n, si, o, so))
        check(abi.lib().sdrhip_demod_process(self._h, _ptr(x), n, n, _ptr(out), n))
        return out

    def process_dev(self, in_ptr, n, in_stride, out_ptr, out_stride):
        check(abi.lib().sdrhip_demod_process_dev(self._h, C.c_void_p(in_ptr), n, in_stride, C.c_void_p(out_ptr), out_stride))

    def reset(self):
        check(abi.lib().sdrhip_demod_reset(self._h))


class FMDeemphI16(_Node):
    """FMDeemph<int16_t>: sequential integer IIR per channel."""
    _destroy = "sdrhip_deemph_i16_destroy"

    def __init__(self, ctx, alpha, channels=1, max_in=65536):
        super().__init__()
        self.ctx, self.channels = ctx, channels
        check(abi.lib().sdrhip_deemph_i16_create(ctx.handle, alpha, channels, max_in, C.byref(self._h)))

    def process(self, x):
        x = np.ascontiguousarray(x, np.int16)
        if x.ndim == 1:
            x = x[None]
        n = x.shape[1]
        out = np.zeros_like(x)
        if device_router is not None and n:
            return device_router(self.ctx, x, out, lambda i, si, o, so: self.process_dev(i, n, si, o, so))
        check(abi.lib().sdrhip_deemph_i16_process(self._h, _ptr(x), n, n, _ptr(out), n))
        return out

    def kernel_names(self, n=0):
        """The kernel a call of n samples per channel runs (0: max_in)."""
        b = C.create_string_buffer(256)
        check(abi.lib().sdrhip_deemph_i16_kernel_names(self._h, n, b, 256))
        return b.value.decode().split(",")

    def process_dev(self, in_ptr, n, in_stride, out_ptr, out_stride):
        check(abi.lib().sdrhip_deemph_i16_process_dev(self._h, C.c_void_p(in_ptr), n, in_stride, C.c_void_p(out_ptr), out_stride))

    def reset(self):
        check(abi.lib().sdrhip_deemph_i16_reset(self._h))


class SubSample(_Node):
    """K6 — SubSample<complex<int16>|complex<float>>."""
    _destroy = "sdrhip_subsample_destroy"

    def __init__(self, ctx, dtype, n, channels=1, max_in=65536):
        super().__init__()
        self.ctx, self.dtype, self.n, self.channels = ctx, dtype, n, channels
        check(abi.lib().sdrhip_subsample_create(ctx.handle, dtype, n, channels, max_in, C.byref(self._h)))

    def out_count(self, n_in):
        n = C.c_size_t(0)
        check(abi.lib().sdrhip_subsample_out_count(self._h, n_in, C.byref(n)))
        return n.value

    def process(self, x):
        it = np.int16 if self.dtype == T_CS16 else np.float32
        x = _as3(x, it)
        n_in = x.shape[1]
        no = self.out_count(n_in)
        out = np.zeros((self.channels, no, 2), it)
        if device_router is not None and n_in and no:
            def call(i, si, o, so):
                assert self.process_dev(i, n_in, si, o, so) == no
            return device_router(self.ctx, x, out, call)
        got = C.c_size_t(0)
        check(abi.lib().sdrhip_subsample_process(self._h, _ptr(x), n_in, n_in, _ptr(out), no, C.byref(got)))
        assert got.value == no
        return out

    def process_dev(self, in_ptr, n_in, in_stride, out_ptr, out_stride):
        got = C.c_size_t(0)
        check(abi.lib().sdrhip_subsample_process_dev(self._h, C.c_void_p(in_ptr), n_in, in_stride, C.c_void_p(out_ptr),
                                                     out_stride, C.byref(got)))
        return got.value

    def reset(self):
        check(abi.lib().sdrhip_subsample_reset(self._h))


class FFTConv(_Node):
    """K7 — FilterSink+FilterSource (mode OLA, kernel = 2N spectrum) or overlap-save with taps (mode OLS).
    `kernels` may be a list of equally sized kernels: a filter bank behind one forward transform per block
    (FilterNode); process() then returns [bands, channels, n, 2]. dtype=np.float64: FilterNode<double>'s plan
    (sdrhip_fftconv_f64_*)."""
    _destroy = "sdrhip_fftconv_destroy"

    def __init__(self, ctx, mode, fft_size, kernel, channels=1, max_in=65536, dtype=np.float32):
        super().__init__()
        bank = isinstance(kernel, (list, tuple))
        self.dtype = np.dtype(dtype)
        self.f64 = self.dtype == np.float64
        self._ct = C.c_double if self.f64 else C.c_float
        ks = [np.ascontiguousarray(k, self.dtype).reshape(-1, 2) for k in (kernel if bank else [kernel])]
        assert all(k.shape == ks[0].shape for k in ks)
        self.ctx, self.mode, self.fft_size, self.channels, self.bands, self._bank = ctx, mode, fft_size, channels, len(ks), bank
        allk = np.ascontiguousarray(np.stack(ks))
        create = abi.lib().sdrhip_fftconv_f64_create_bank if self.f64 else abi.lib().sdrhip_fftconv_create_bank
        check(create(ctx.handle, mode, fft_size, allk.ctypes.data_as(C.POINTER(self._ct)), ks[0].shape[0], len(ks), channels, max_in,
                     C.byref(self._h)))

    def process(self, x):
        x = _as3(x, self.dtype)
        n = x.shape[1]
        out = np.zeros((self.bands,) + x.shape, self.dtype)
        if device_router is not None and n:
            flat = out.reshape((self.bands * x.shape[0],) + x.shape[1:])   # band-major rows, as the C ABI lays them out
            device_router(self.ctx, x, flat, lambda i, si, o, so: self.process_dev(i, n, si, o, so))
            return out if self._bank else out[0]
        fn = abi.lib().sdrhip_fftconv_f64_process if self.f64 else abi.lib().sdrhip_fftconv_process
        check(fn(self._h, _ptr(x), n, n, _ptr(out), n))
        return out if self._bank else out[0]

    def set_kernel(self, band, kernel):
        kernel = np.ascontiguousarray(kernel, self.dtype).reshape(-1, 2)
        fn = abi.lib().sdrhip_fftconv_f64_set_kernel if self.f64 else abi.lib().sdrhip_fftconv_set_kernel
        check(fn(self._h, band, kernel.ctypes.data_as(C.POINTER(self._ct))))

    def process_dev(self, in_ptr, n, in_stride, out_ptr, out_stride):
        fn = abi.lib().sdrhip_fftconv_f64_process_dev if self.f64 else abi.lib().sdrhip_fftconv_process_dev
        check(fn(self._h, C.c_void_p(in_ptr), n, in_stride, C.c_void_p(out_ptr), out_stride))

    def reset(self):
        check(abi.lib().sdrhip_fftconv_reset(self._h))


class FloatBaseBand(_Node):
    """Build-defined float baseband (BASELINE config 2): shift -> FIR(cf32) -> /D."""
    _destroy = "sdrhip_fbb_f32_destroy"

    def __init__(self, ctx, Fc, Fs, alpha, decim, channels=1, max_in=65536):
        super().__init__()
        alpha = np.ascontiguousarray(alpha, np.float64)
        self.ctx, self.channels, self.decim, self.order = ctx, channels, decim, alpha.shape[0]
        check(abi.lib().sdrhip_fbb_f32_create(ctx.handle, Fc, Fs, alpha.ctypes.data_as(C.POINTER(C.c_double)),
                                              alpha.shape[0], decim, channels, max_in, C.byref(self._h)))

    def kernel_names(self, n_in=0):
        """The kernel a call of n_in samples per channel runs (0: max_in)."""
        b = C.create_string_buffer(256)
        check(abi.lib().sdrhip_fbb_f32_kernel_names(self._h, n_in, b, 256))
        return b.value.decode().split(",")

    def out_count(self, n_in):
        n = C.c_size_t(0)
        check(abi.lib().sdrhip_fbb_f32_out_count(self._h, n_in, C.byref(n)))
        return n.value

    def process(self, x):
        x = _as3(x, np.float32)
        n_in = x.shape[1]
        no = self.out_count(n_in)
        out = np.zeros((self.channels, no, 2), np.float32)
        if device_router is not None and n_in and no:
            def call(i, si, o, so):
                assert self.process_dev(i, n_in, si, o, so) == no
            return device_router(self.ctx, x, out, call)
        got = C.c_size_t(0)
        check(abi.lib().sdrhip_fbb_f32_process(self._h, _ptr(x), n_in, n_in, _ptr(out), no, C.byref(got)))
        assert got.value == no
        return out

    def process_dev(self, in_ptr, n_in, in_stride, out_ptr, out_stride):
        got = C.c_size_t(0)
        check(abi.lib().sdrhip_fbb_f32_process_dev(self._h, C.c_void_p(in_ptr), n_in, in_stride, C.c_void_p(out_ptr),
                                                   out_stride, C.byref(got)))
        return got.value

    def reset(self):
        check(abi.lib().sdrhip_fbb_f32_reset(self._h))

    def set_taps(self, alpha):
        alpha = np.ascontiguousarray(alpha, np.float64)
        assert alpha.shape == (self.order,)   # (the C side reads `order` doubles)
        check(abi.lib().sdrhip_fbb_f32_set_taps(self._h, alpha.ctypes.data_as(C.POINTER(C.c_double))))

    def set_shift(self, Fc):
        check(abi.lib().sdrhip_fbb_f32_set_shift(self._h, float(Fc)))


def fft_c2c(ctx, x, sign):
    """Batched DFT with the library's own in-LDS FFT (test hook)."""
    x = np.ascontiguousarray(x, np.float32)
    batch, n = x.shape[0], x.shape[1]
    din, dout = ctx.malloc(x.nbytes), ctx.malloc(x.nbytes)
    try:
        ctx.h2d(din, x)
        check(abi.lib().sdrhip_fft_c2c(ctx.handle, n, sign, batch, C.c_void_p(din), C.c_void_p(dout)))
        out = np.zeros_like(x)
        ctx.d2h(out, dout)
    finally:
        ctx.free(din)
        ctx.free(dout)
    return out


def fft_c2c_f64(ctx, x, sign):
    """Batched DFT on complex<double> (FFTPlan<double>): x is (batch, n, 2) float64."""
    x = np.ascontiguousarray(x, np.float64)
    batch, n = x.shape[0], x.shape[1]
    din, dout = ctx.malloc(x.nbytes), ctx.malloc(x.nbytes)
    try:
        ctx.h2d(din, x)
        check(abi.lib().sdrhip_fft_c2c_f64(ctx.handle, n, sign, batch, C.c_void_p(din), C.c_void_p(dout)))
        out = np.zeros_like(x)
        ctx.d2h(out, dout)
    finally:
        ctx.free(din)
        ctx.free(dout)
    return out


def fft_exec(ctx, x, sign):
    """FFT::exec on a host buffer: x complex64 or complex128, one transform."""
    x = np.ascontiguousarray(x)
    assert x.dtype in (np.complex64, np.complex128) and x.ndim == 1
    out = np.empty_like(x)
    check(abi.lib().sdrhip_fft_exec(ctx.handle, abi.T_CF64 if x.dtype == np.complex128 else abi.T_CF32, x.shape[0], sign,
                                    _ptr(x), _ptr(out)))
    return out


class Comm:
    """sdrhip_comm_*: one process, one rank context per device (RCCL between distinct devices, same-device copies when
    every rank sits on one device). Mirrors what sdr::gpu::ChannelBank(devices) does on the C++ side."""

    def __init__(self, devices):
        devs = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        check(abi.lib().sdrhip_comm_create(devs, len(devices), C.byref(self._h)))
        self.devices = list(devices)
        self.ctx = []
        for r, d in enumerate(devices):
            h = C.c_void_p()
            check(abi.lib().sdrhip_comm_ctx(self._h, r, C.byref(h)))
            self.ctx.append(Context.borrowed(h, d))

    @property
    def transport(self):
        s = C.c_char_p()
        check(abi.lib().sdrhip_comm_transport(self._h, C.byref(s)))
        return s.value.decode()

    def broadcast(self, ptrs, nbytes, root=0):
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        check(abi.lib().sdrhip_comm_broadcast(self._h, arr, nbytes, root))

    def gather(self, send_ptrs, nbytes, recv_ptr, root=0):
        sp = (C.c_void_p * len(send_ptrs))(*send_ptrs)
        nb = (C.c_size_t * len(nbytes))(*nbytes)
        check(abi.lib().sdrhip_comm_gather(self._h, sp, nb, C.c_void_p(recv_ptr), root))

    def gather_begin(self, slot, send_ptrs, nbytes, recv_ptr, root=0):
        """The gather on the comm's own streams (overlaps the ranks' next kernels); gather_wait(slot) orders the ranks' streams behind it."""
        sp = (C.c_void_p * len(send_ptrs))(*send_ptrs)
        nb = (C.c_size_t * len(nbytes))(*nbytes)
        check(abi.lib().sdrhip_comm_gather_begin(self._h, slot, sp, nb, C.c_void_p(recv_ptr), root))

    def gather_wait(self, slot):
        check(abi.lib().sdrhip_comm_gather_wait(self._h, slot))

    def synchronize(self):
        check(abi.lib().sdrhip_comm_synchronize(self._h))

    def close(self):
        if self._h:
            for c in self.ctx:
                c.close()
            abi.lib().sdrhip_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FFTPlan(_Node):
    """sdrhip_fft_plan_*: FFTPlan<float|double> planned once (any size), executed many times."""
    _destroy = "sdrhip_fft_plan_destroy"

    def __init__(self, ctx, n, dtype=np.complex64):
        super().__init__()
        self.ctx, self.n, self.dtype = ctx, int(n), np.dtype(dtype)
        assert self.dtype in (np.dtype(np.complex64), np.dtype(np.complex128))
        check(abi.lib().sdrhip_fft_plan_create(ctx.handle, abi.T_CF64 if self.dtype == np.complex128 else abi.T_CF32, self.n, C.byref(self._h)))

    @property
    def form(self):
        s = C.c_char_p()
        check(abi.lib().sdrhip_fft_plan_form(self._h, C.byref(s)))
        return s.value.decode()

    def exec(self, x, sign):
        """One transform on host buffers (FFTPlan::operator())."""
        x = np.ascontiguousarray(x, self.dtype)
        assert x.shape == (self.n,)
        out = np.empty_like(x)
        check(abi.lib().sdrhip_fft_plan_exec(self._h, sign, _ptr(x), _ptr(out)))
        return out

    def exec_batch(self, x, sign):
        """x: [batch, n] -> [batch, n] through device memory (exec_dev)."""
        x = np.ascontiguousarray(x, self.dtype)
        assert x.ndim == 2 and x.shape[1] == self.n
        din, dout = self.ctx.malloc(x.nbytes), self.ctx.malloc(x.nbytes)
        try:
            self.ctx.h2d(din, x)
            check(abi.lib().sdrhip_fft_plan_exec_dev(self._h, sign, x.shape[0], C.c_void_p(din), C.c_void_p(dout)))
            out = np.empty_like(x)
            self.ctx.d2h(out, dout)
        finally:
            self.ctx.free(din)
            self.ctx.free(dout)
        return out
