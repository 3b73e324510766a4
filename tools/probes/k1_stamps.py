"""Diagnostic (tools/build_variant.sh stamps "-DK1_STAMPS"): per-phase shader-clock totals of the K1 hot kernel.
usage: python tools/probes/k1_stamps.py [variant-name=stamps] [order=127] [cu8=0] [epi=1] [decim=8] [fs=2.4e6] [width=50e3]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SDRHIP_LIB"] = os.path.join(ROOT, "libsdr_amd", "libsdrhip_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "stamps"))
order = int(sys.argv[2]) if len(sys.argv) > 2 else 127
cu8 = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 1
decim = int(sys.argv[5]) if len(sys.argv) > 5 else 8
import torch
import numpy as np
import libsdr_amd as sa
FS = float(sys.argv[6]) if len(sys.argv) > 6 else 2.4e6
WIDTH = float(sys.argv[7]) if len(sys.argv) > 7 else 50e3
C, N = 1024, 65536
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = sa.Context(0, stream=stream.cuda_stream)
    taps = sa.design_iqbb_taps(100e3, WIDTH, FS, order); lut = sa.design_freqshift_lut_i16()
    node = sa.IQBaseBandI16(ctx, taps, lut, sa.design_freqshift_inc(100e3, FS), False, decim, channels=C, max_in=N, epilogue=epi)
    if cu8:
        node.set_input_format(sa.abi.IN_CU8)
        x = [torch.randint(0, 256, (C, N, 2), dtype=torch.uint8, device=dev) for _ in range(3)]
    else:
        x = [torch.randint(-8000, 8000, (C, N, 2), dtype=torch.int16, device=dev) for _ in range(3)]
    out = torch.zeros((C, N // decim + 2, 2), dtype=torch.int16, device=dev)
    L = sa.abi.lib()
    L.sdrhip_debug_k1_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    W = 32768 * 16
    buf = (ctypes.c_ulonglong * W)()
    for i in range(20):
        node.process_dev(x[i % 3].data_ptr(), N, N, out.data_ptr(), out.shape[1])
    torch.cuda.synchronize()
    K = 2000   # ~0.2 s of back-to-back launches: the clock the chip holds
    t0 = time.perf_counter()
    for i in range(K):
        node.process_dev(x[i % 3].data_ptr(), N, N, out.data_ptr(), out.shape[1])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    assert L.sdrhip_debug_k1_stamps(node._h, buf, W) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 16)
a = a[a[:, 8] > 0]
waves = len(a)
hwid = a[:, 9].astype(np.int64)
tiles = float(a[:, 8].sum())
t0, t1 = a[:, 10].astype(np.int64), a[:, 11].astype(np.int64)
span = (t1.max() - t0.min()) / 100.0   # us (100 MHz)
life = (t1 - t0) / 100.0
print("order %d cu8 %d epi %d kernels %s" % (order, cu8, epi, node.kernel_names))
print("clock %.3f GHz;" % (float(a[:, :6].sum()) / float((t1 - t0).sum()) / 10.0), "kernel span %.1f us; wave life mean %.1f us (min %.1f max %.1f); sum(life)/span/1024 SIMDs = %.2f waves per SIMD" % (span, life.mean(), life.min(), life.max(), life.sum() / span / 1024))
v = [float(a[:, i].sum()) for i in range(6)]
tot = sum(v)
names = ["bookkeeping", "DMA issue + wait", "raw->planes", "K loop", "recombine/rotate/sum", "finish/demod/store"]
print("launch %.1f us; waves %d, wave-slices %.0f; cycles per wave %.0f, per wave-slice %.0f" % (dt * 1e6, waves, tiles, tot / waves, tot / tiles))
for n_, c_ in zip(names, v):
    print("  %-24s %6.1f %%   %8.0f cycles per wave-slice" % (n_, 100.0 * c_ / tot, c_ / tiles))
slot, simd = hwid & 15, (hwid >> 4) & 3
xcc, cu, se = a[:, 12].astype(np.int64) & 15, (hwid >> 8) & 15, (hwid >> 13) & 7
print("life quantiles (us): " + " ".join("%d%%:%.1f" % (q, np.percentile(life, q)) for q in (0, 5, 25, 50, 75, 95, 100)))
print("end-time quantiles (us after the first start): " + " ".join("%d%%:%.1f" % (q, np.percentile((t1 - t0.min()) / 100.0, q)) for q in (0, 5, 25, 50, 75, 95, 100)))
print("start-time quantiles (us): " + " ".join("%d%%:%.1f" % (q, np.percentile((t0 - t0.min()) / 100.0, q)) for q in (0, 50, 95, 100)))
for name, key in (("slot", slot), ("simd", simd), ("xcc", xcc), ("se", se), ("cu", cu)):
    print("life by %-4s:" % name, " ".join("%d:%.0f" % (v_, life[key == v_].mean()) for v_ in np.unique(key)))
