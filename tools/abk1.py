#!/usr/bin/env python3
"""In-process A/B of K1 builds on ONE device: every variant (a libsdrhip*.so build, optionally with environment
settings) gets its own plan over the SAME resident input batches; timing rounds are interleaved (variant A, B, C, A, B,
C, ...) so that clock drift and box-to-box differences cancel. Reports median / min ms per launch per variant.

usage: tools/abk1.py [--order 127] [--epi fm|usb|am|none] [--cu8|--real] [--decim 8] [--fc 100e3] [--channels 1024] [--samples 65536] [--rounds 7]
                     [--launches 200] name=libsdr_amd/libsdrhip_x.so[@ENV=VAL[,ENV=VAL]] ...
"""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402  (initialises HIP before the libraries do)

torch.cuda.init()
import libsdr_amd as sa   # noqa: E402  (designers only; the default library)


def load(path):
    L = C.CDLL(os.path.abspath(path))
    vp, sz = C.c_void_p, C.c_size_t
    i32p = C.POINTER(C.c_int32)
    L.sdrhip_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.sdrhip_iqbb_i16_create.argtypes = [vp, i32p, C.c_int, i32p, C.c_uint32, C.c_int, C.c_int, C.c_int, sz, C.c_int, C.POINTER(vp)]
    L.sdrhip_bb_i16_create.argtypes = L.sdrhip_iqbb_i16_create.argtypes
    L.sdrhip_iqbb_i16_set_input_format.argtypes = [vp, C.c_int]
    L.sdrhip_iqbb_i16_process_dev.argtypes = [vp, vp, sz, sz, vp, sz, C.POINTER(sz)]
    L.sdrhip_iqbb_i16_kernel_names.argtypes = [vp, C.c_char_p, sz]
    L.sdrhip_last_error.restype = C.c_char_p
    return L


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--order", type=int, default=127)
    p.add_argument("--epi", default="fm")
    p.add_argument("--cu8", action="store_true")
    p.add_argument("--decim", type=int, default=8)
    p.add_argument("--fc", type=float, default=100e3, help="centre frequency (0: no shift)")
    p.add_argument("--real", action="store_true", help="the real-input BaseBand<int16> (sdrhip_bb_i16_create)")
    p.add_argument("--channels", type=int, default=1024)
    p.add_argument("--samples", type=int, default=65536)
    p.add_argument("--rounds", type=int, default=7)
    p.add_argument("--launches", type=int, default=200)
    p.add_argument("variants", nargs="+")
    a = p.parse_args()
    FS = 2.4e6
    Cn, N = a.channels, a.samples
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    taps = sa.design_bb_taps(100e3, 50e3, FS, a.order) if a.real else sa.design_iqbb_taps(a.fc, 50e3, FS, a.order)
    lut = sa.design_freqshift_lut_i16()
    inc = sa.design_freqshift_inc(a.fc, FS)
    epi = {"none": 0, "fm": 1, "am": 2, "usb": 3}[a.epi]
    with torch.cuda.stream(stream):
        if a.real:
            xs = [torch.randint(-8000, 8000, (Cn, N), dtype=torch.int16, device=dev) for _ in range(3)]
        elif a.cu8:
            xs = [torch.randint(0, 256, (Cn, N, 2), dtype=torch.uint8, device=dev) for _ in range(3)]
        else:
            xs = [torch.randint(-8000, 8000, (Cn, N, 2), dtype=torch.int16, device=dev) for _ in range(3)]
        out = torch.zeros((Cn, N // min(a.decim, 8) + 2, 2), dtype=torch.int16, device=dev)
        plans = []
        for v in a.variants:
            name, rest = v.split("=", 1)
            path, _, envs = rest.partition("@")
            env = dict(e.split("=", 1) for e in envs.split(",") if e)
            os.environ.update(env)
            L = load(path)
            ctx, h = C.c_void_p(), C.c_void_p()
            assert L.sdrhip_ctx_create(0, C.c_void_p(stream.cuda_stream), C.byref(ctx)) == 0
            rc = (L.sdrhip_bb_i16_create if a.real else L.sdrhip_iqbb_i16_create)(ctx, taps.ctypes.data_as(C.POINTER(C.c_int32)), a.order, lut.ctypes.data_as(C.POINTER(C.c_int32)), inc, 0, a.decim,
                                          Cn, N, epi, C.byref(h))
            assert rc == 0, L.sdrhip_last_error()
            if a.cu8:
                assert L.sdrhip_iqbb_i16_set_input_format(h, 1) == 0
            b = C.create_string_buffer(256)
            L.sdrhip_iqbb_i16_kernel_names(h, b, 256)
            for k in env:
                del os.environ[k]
            plans.append((name, L, h, env, b.value.decode()))
        no = C.c_size_t(0)

        def run(pl, k):
            name, L, h, env, _ = pl
            os.environ.update(env)   # (launch-time hooks such as SDRHIP_IQBB_TPW)
            for i in range(k):
                rc = L.sdrhip_iqbb_i16_process_dev(h, C.c_void_p(xs[i % 3].data_ptr()), N, N, C.c_void_p(out.data_ptr()), out.shape[1], C.byref(no))
                assert rc == 0, L.sdrhip_last_error()
            for k_ in env:
                del os.environ[k_]
        for pl in plans:
            run(pl, 50)
        torch.cuda.synchronize()
        res = {pl[0]: [] for pl in plans}
        for r in range(a.rounds):
            for pl in (plans if r % 2 == 0 else plans[::-1]):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                run(pl, 20)
                e0.record(stream)
                run(pl, a.launches)
                e1.record(stream)
                e1.synchronize()
                res[pl[0]].append(e0.elapsed_time(e1) / a.launches)
    base = statistics.median(res[plans[0][0]])
    for name, L, h, env, kn in plans:
        v = res[name]
        med = statistics.median(v)
        print("%-14s median %.4f ms  min %.4f  max %.4f  (%+.1f %% vs %s)  %s %s" % (name, med, min(v), max(v), 100.0 * (med / base - 1.0), plans[0][0], kn, env or ""))


if __name__ == "__main__":
    main()
