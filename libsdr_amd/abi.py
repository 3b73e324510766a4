"""ctypes binding of the C ABI declared in include/sdrhip.h (libsdr_amd/libsdrhip.so).

This is plumbing for tests and bench.py; the product is the shared library and the C++ nodes in
include/sdr/gpu/.  There is NO CPU fallback here: if the HIP library is missing or no GPU is
present, loading / context creation raises.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SO_PATH = os.path.join(_HERE, "libsdrhip.so")
if os.environ.get("SDRHIP_LIB"):   # tuning hook: A/B an earlier build of the library on the same box (tools/k1_variants.sh)
    SO_PATH = os.path.abspath(os.environ["SDRHIP_LIB"])
HEADER = os.path.join(ROOT, "include", "sdrhip.h")

OK, E_INVALID, E_NODEVICE, E_HIP, E_NOMEM, E_UNSUPPORTED, E_SIZE = 0, -1, -2, -3, -4, -5, -6
EPI_NONE, EPI_FM, EPI_AM, EPI_USB = 0, 1, 2, 3
FIR_CS16_EXACT, FIR_CF32 = 0, 1
T_CS16, T_CF32, T_CS8, T_CF64 = 0, 1, 2, 3
IN_CS16, IN_CU8 = 0, 1
KEEP_RING, KEEP_FM, KEEP_COUNTERS = 1, 2, 4
FFTCONV_OLA, FFTCONV_OLS = 0, 1


class SdrHipError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("sdrhip error %d: %s" % (code, text))
        self.code = code


def header_functions():
    """Names of every function include/sdrhip.h declares (used by the symbol-export test)."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdrhip_[a-z0-9_]+)\s*\(", src)))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("libsdrhip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C libsdr_amd/csrc); there is no CPU fallback")
        L = C.CDLL(SO_PATH)
        vp, sz = C.c_void_p, C.c_size_t
        pvp, psz = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)
        i32p, f32p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)
        sig = {
            "sdrhip_version": (C.c_int, []),
            "sdrhip_strerror": (C.c_char_p, [C.c_int]),
            "sdrhip_last_error": (C.c_char_p, []),
            "sdrhip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
            "sdrhip_ctx_create": (C.c_int, [C.c_int, vp, pvp]),
            "sdrhip_ctx_destroy": (C.c_int, [vp]),
            "sdrhip_ctx_synchronize": (C.c_int, [vp]),
            "sdrhip_ctx_device_name": (C.c_int, [vp, C.c_char_p, sz]),
            "sdrhip_malloc": (C.c_int, [vp, sz, pvp]),
            "sdrhip_free": (C.c_int, [vp, vp]),
            "sdrhip_memcpy_h2d": (C.c_int, [vp, vp, vp, sz]),
            "sdrhip_memcpy_d2h": (C.c_int, [vp, vp, vp, sz]),
            "sdrhip_memset": (C.c_int, [vp, vp, C.c_int, sz]),
            "sdrhip_timer_create": (C.c_int, [vp, pvp]),
            "sdrhip_timer_start": (C.c_int, [vp]),
            "sdrhip_timer_stop": (C.c_int, [vp]),
            "sdrhip_timer_elapsed_ms": (C.c_int, [vp, f32p]),
            "sdrhip_timer_destroy": (C.c_int, [vp]),
            "sdrhip_bench_stream_read": (C.c_int, [vp, vp, sz, C.c_int, C.POINTER(C.c_double)]),
            "sdrhip_design_iqbb_taps": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_int, i32p]),
            "sdrhip_design_bb_taps": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_int, i32p]),
            "sdrhip_design_iqbb_decim": (C.c_int, [C.c_double, C.c_int, C.c_double, C.POINTER(C.c_int)]),
            "sdrhip_design_freqshift_lut_i16": (C.c_int, [i32p]),
            "sdrhip_design_freqshift_inc": (C.c_int, [C.c_double, C.c_double, C.POINTER(C.c_uint32)]),
            "sdrhip_design_fir_lowpass": (C.c_int, [C.c_int, C.c_double, C.c_double, f64p]),
            "sdrhip_design_fftfilt_kernel": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_double, f32p]),
            "sdrhip_design_fftfilt_spectrum": (C.c_int, [C.c_int, f32p, f32p]),
            "sdrhip_iqbb_i16_create": (C.c_int, [vp, i32p, C.c_int, i32p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                                 sz, C.c_int, pvp]),
            "sdrhip_bb_i16_create": (C.c_int, [vp, i32p, C.c_int, i32p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                               sz, C.c_int, pvp]),
            "sdrhip_iqbb_i16_path": (C.c_int, [vp, C.POINTER(C.c_int)]),
            "sdrhip_design_freqshift_lut_i8": (C.c_int, [i32p]),
            "sdrhip_iqbb_i8_create": (C.c_int, [vp, i32p, C.c_int, i32p, C.c_uint32, C.c_int, C.c_int, C.c_int, sz, C.c_int, pvp]),
            "sdrhip_iqbb_i16_kernel_names": (C.c_int, [vp, C.c_char_p, sz]),
            "sdrhip_iqbb_i16_set_taps": (C.c_int, [vp, i32p]),
            "sdrhip_iqbb_i16_set_shift": (C.c_int, [vp, C.c_uint32, C.c_int]),
            "sdrhip_iqbb_i16_out_count": (C.c_int, [vp, sz, psz]),
            "sdrhip_iqbb_i16_process": (C.c_int, [vp, vp, sz, sz, vp, sz, psz]),
            "sdrhip_iqbb_i16_process_dev": (C.c_int, [vp, vp, sz, sz, vp, sz, psz]),
            "sdrhip_iqbb_i16_plan_info": (C.c_int, [vp, C.POINTER(C.c_int), C.c_int]),
            "sdrhip_iqbb_i16_process_dev_multi": (C.c_int, [vp, vp, sz, sz, sz, vp, sz, psz, psz]),
            "sdrhip_iqbb_i16_reset": (C.c_int, [vp, C.c_int]),
            "sdrhip_iqbb_i16_adopt_state": (C.c_int, [vp, vp, C.c_int]),
            "sdrhip_iqbb_i16_destroy": (C.c_int, [vp]),
            "sdrhip_fir_create": (C.c_int, [vp, C.c_int, f64p, C.c_int, C.c_int, C.c_int, sz, C.c_int, pvp]),
            "sdrhip_fir_out_count": (C.c_int, [vp, sz, psz]),
            "sdrhip_fir_kernel_names": (C.c_int, [vp, sz, C.c_char_p, sz]),
            "sdrhip_fir_process": (C.c_int, [vp, vp, sz, sz, vp, sz, psz]),
            "sdrhip_fir_process_dev": (C.c_int, [vp, vp, sz, sz, vp, sz, psz]),
            "sdrhip_fir_reset": (C.c_int, [vp]),
            "sdrhip_fir_set_taps": (C.c_int, [vp, f64p]),
            "sdrhip_fir_destroy": (C.c_int, [vp]),
            "sdrhip_demod_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, sz, C.c_int, pvp]),
            "sdrhip_demod_process": (C.c_int, [vp, vp, sz, sz, vp, sz]),
            "sdrhip_demod_process_dev": (C.c_int, [vp, vp, sz, sz, vp, sz]),
            "sdrhip_demod_reset": (C.c_int, [vp]),
            "sdrhip_demod_destroy": (C.c_int, [vp]),
            "sdrhip_design_fmdeemph_alpha": (C.c_int, [C.c_double, C.POINTER(C.c_int)]),
            "sdrhip_deemph_i16_create": (C.c_int, [vp, C.c_int, C.c_int, sz, pvp]),
            "sdrhip_deemph_i16_process": (C.c_int, [vp, vp, sz, sz, vp, sz]),
            "sdrhip_deemph_i16_process_dev": (C.c_int, [vp, vp, sz, sz, vp, sz]),
            "sdrhip_deemph_i16_kernel_names": (C.c_int, [vp, sz, C.c_char_p, sz]),
            "sdrhip_deemph_i16_reset": (C.c_int, [vp]),
            "sdrhip_deemph_i16_destroy": (C.c_int, [vp]),
            "sdrhip_iqbb_i16_set_input_format": (C.c_int, [vp, C.c_int]),
            "sdrhip_subsample_create": (C.c_int, [vp, C.c_int, sz, C.c_int, sz, pvp]),
            "sdrhip_subsample_out_count": (C.c_int, [vp, sz, psz]),
            "sdrhip_subsample_process": (C.c_int, [vp, vp, sz, sz, vp, sz, psz]),
            "sdrhip_subsample_process_dev": (C.c_int, [vp, vp, sz, sz, vp, sz, psz]),
            "sdrhip_subsample_reset": (C.c_int, [vp]),
            "sdrhip_subsample_destroy": (C.c_int, [vp]),
            "sdrhip_fftconv_create": (C.c_int, [vp, C.c_int, C.c_int, f32p, C.c_int, C.c_int, sz, pvp]),
            "sdrhip_fftconv_process": (C.c_int, [vp, vp, sz, sz, vp, sz]),
            "sdrhip_fftconv_process_dev": (C.c_int, [vp, vp, sz, sz, vp, sz]),
            "sdrhip_fftconv_reset": (C.c_int, [vp]),
            "sdrhip_fftconv_destroy": (C.c_int, [vp]),
            "sdrhip_fft_c2c": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
            "sdrhip_fft_c2c_f64": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
            "sdrhip_fft_exec": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
            "sdrhip_fbb_f32_create": (C.c_int, [vp, C.c_double, C.c_double, f64p, C.c_int, C.c_int, C.c_int, sz, pvp]),
            "sdrhip_fbb_f32_out_count": (C.c_int, [vp, sz, psz]),
            "sdrhip_fbb_f32_kernel_names": (C.c_int, [vp, sz, C.c_char_p, sz]),
            "sdrhip_fbb_f32_process": (C.c_int, [vp, vp, sz, sz, vp, sz, psz]),
            "sdrhip_fbb_f32_process_dev": (C.c_int, [vp, vp, sz, sz, vp, sz, psz]),
            "sdrhip_fbb_f32_reset": (C.c_int, [vp]),
            "sdrhip_fbb_f32_set_taps": (C.c_int, [vp, f64p]),
            "sdrhip_fbb_f32_set_shift": (C.c_int, [vp, C.c_double]),
            "sdrhip_fbb_f32_destroy": (C.c_int, [vp]),
            "sdrhip_fftconv_create_bank": (C.c_int, [vp, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_int, sz, pvp]),
            "sdrhip_fftconv_bands": (C.c_int, [vp, C.POINTER(C.c_int)]),
            "sdrhip_fftconv_set_kernel": (C.c_int, [vp, C.c_int, f32p]),
            "sdrhip_design_fftfilt_kernel_f64": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_double, f64p]),
            "sdrhip_design_fftfilt_spectrum_f64": (C.c_int, [C.c_int, f64p, f64p]),
            "sdrhip_fftconv_f64_create_bank": (C.c_int, [vp, C.c_int, C.c_int, f64p, C.c_int, C.c_int, C.c_int, sz, pvp]),
            "sdrhip_fftconv_f64_set_kernel": (C.c_int, [vp, C.c_int, f64p]),
            "sdrhip_fftconv_f64_process": (C.c_int, [vp, vp, sz, sz, vp, sz]),
            "sdrhip_fftconv_f64_process_dev": (C.c_int, [vp, vp, sz, sz, vp, sz]),
            "sdrhip_fft_plan_create": (C.c_int, [vp, C.c_int, C.c_int, pvp]),
            "sdrhip_fft_plan_form": (C.c_int, [vp, C.POINTER(C.c_char_p)]),
            "sdrhip_fft_plan_exec_dev": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
            "sdrhip_fft_plan_exec": (C.c_int, [vp, C.c_int, vp, vp]),
            "sdrhip_fft_plan_destroy": (C.c_int, [vp]),
            "sdrhip_comm_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, pvp]),
            "sdrhip_comm_size": (C.c_int, [vp, C.POINTER(C.c_int)]),
            "sdrhip_comm_ctx": (C.c_int, [vp, C.c_int, pvp]),
            "sdrhip_comm_transport": (C.c_int, [vp, C.POINTER(C.c_char_p)]),
            "sdrhip_comm_broadcast": (C.c_int, [vp, pvp, sz, C.c_int]),
            "sdrhip_comm_gather": (C.c_int, [vp, pvp, psz, vp, C.c_int]),
            "sdrhip_comm_gather_begin": (C.c_int, [vp, C.c_int, pvp, psz, vp, C.c_int]),
            "sdrhip_comm_gather_wait": (C.c_int, [vp, C.c_int]),
            "sdrhip_comm_synchronize": (C.c_int, [vp]),
            "sdrhip_comm_destroy": (C.c_int, [vp]),
            "sdrhip_host_alloc": (C.c_int, [sz, pvp]),
            "sdrhip_host_free": (C.c_int, [vp]),
            "sdrhip_host_register": (C.c_int, [vp, sz]),
            "sdrhip_host_unregister": (C.c_int, [vp]),
            "sdrhip_memcpy_h2d_async": (C.c_int, [vp, vp, vp, sz]),
            "sdrhip_memcpy_d2h_async": (C.c_int, [vp, vp, vp, sz]),
            "sdrhip_memcpy2d_d2h_async": (C.c_int, [vp, vp, sz, vp, sz, sz, sz]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        L._declared = sorted(sig)
        _lib = L
    return _lib


def check(code):
    if code != OK:
        L = lib()
        raise SdrHipError(code, "%s: %s" % (L.sdrhip_strerror(code).decode(), L.sdrhip_last_error().decode()))
