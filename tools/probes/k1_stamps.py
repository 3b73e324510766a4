"""Diagnostic (tools/build_variant.sh stamps "-DK1_STAMPS"): per-phase shader-clock totals of the K1 hot kernel."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SDRHIP_LIB"] = os.path.join(ROOT, "libsdr_amd", "libsdrhip_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "stamps"))
import torch
import libsdr_amd as sa
FS = 2.4e6
C, N = 1024, 65536
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = sa.Context(0, stream=stream.cuda_stream)
    taps = sa.design_iqbb_taps(100e3, 50e3, FS, 127); lut = sa.design_freqshift_lut_i16()
    node = sa.IQBaseBandI16(ctx, taps, lut, sa.design_freqshift_inc(100e3, FS), False, 8, channels=C, max_in=N, epilogue=sa.EPI_FM)
    x = [torch.randint(-8000, 8000, (C, N, 2), dtype=torch.int16, device=dev) for _ in range(3)]
    out = torch.zeros((C, N // 8 + 2), dtype=torch.int16, device=dev)
    L = sa.abi.lib()
    L.sdrhip_debug_k1_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    W = 32768 * 8
    buf = (ctypes.c_ulonglong * W)()
    for i in range(20):
        node.process_dev(x[i % 3].data_ptr(), N, N, out.data_ptr(), out.shape[1])
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for i in range(K):
        node.process_dev(x[i % 3].data_ptr(), N, N, out.data_ptr(), out.shape[1])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    L.sdrhip_debug_k1_stamps(buf, W)
import numpy as np
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 5] > 0]
waves = len(a)
hwid = (a[:, 5] >> np.uint64(32)).astype(np.int64)
a[:, 5] &= np.uint64(0xffffffff)
tiles = float(a[:, 5].sum())
t0, t1 = a[:, 6].astype(np.int64), a[:, 7].astype(np.int64)
span = (t1.max() - t0.min()) / 100.0   # us (100 MHz)
life = (t1 - t0) / 100.0
print("clock %.3f GHz;" % (float(a[:, :5].sum()) / float((t1 - t0).sum()) / 10.0), "kernel span %.1f us; wave life mean %.1f us (min %.1f max %.1f); sum(life)/span/1024 SIMDs = %.2f waves per SIMD" % (span, life.mean(), life.min(), life.max(), life.sum() / span / 1024))
print("wave-slot histogram (HW_ID[3:0]):", np.bincount(hwid & 15, minlength=8)[:8].tolist(), " simd:", np.bincount((hwid >> 4) & 3).tolist())
ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)]); ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1]); tt = (ev[:, 0] - ev[0, 0]) / 100.0
for q in (0.1, 0.3, 0.5, 0.7, 0.9):
    i = np.searchsorted(tt, q * span); print("  t=%5.1f us: %d waves resident (%.2f per SIMD)" % (q * span, conc[i], conc[i] / 1024.0))
v = [float(a[:, i].sum()) for i in range(5)]
tot = sum(v)
names = ["wait DMA", "raw->planes", "K loop", "recombine/rotate/sum", "finish/demod/store"]
print("launch %.1f us; waves %d, wave-tiles %.0f; cycles per wave %.0f, per wave-tile %.0f" % (dt * 1e6, waves, tiles, tot / waves, tot / tiles))
for n_, c_ in zip(names, v[:5]):
    print("  %-24s %6.1f %%   %8.0f cycles per wave-tile" % (n_, 100.0 * c_ / tot, c_ / tiles))

# where does the spread of wave lifetimes come from? by wave slot of the SIMD, by SIMD, by CU (HW_ID fields)
slot, simd, cu, se = hwid & 15, (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 13) & 7
for name, key in (("slot", slot), ("simd", simd), ("se", se), ("cu", cu)):
    print("life by %-4s:" % name, " ".join("%d:%.0f" % (v_, life[key == v_].mean()) for v_ in np.unique(key)))
cukey = se * 16 + cu
m = np.array([life[cukey == v_].mean() for v_ in np.unique(cukey)])
print("per-CU mean life: min %.1f max %.1f std %.1f us over %d (se,cu) groups; within-CU std %.1f us" %
      (m.min(), m.max(), m.std(), len(m), np.mean([life[cukey == v_].std() for v_ in np.unique(cukey)])))
