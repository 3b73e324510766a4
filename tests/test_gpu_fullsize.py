"""BASELINE config 5 at its G = 1 point: ALL 8192 channels on one MI355X (C = 8192 split over G in {1,2,4,8};
the per-GPU shards of G = 8 are tests/test_gpu_parity.py::test_iqbb_usb_full_size). One input batch is 8192 x 65536 x 4 B =
2^31 bytes — the place where a 32-bit byte offset breaks — and the rows are taken from a buffer of twice that stride, so the
last channel's row starts beyond 2^32 bytes. Everything stays on the device (the host holds only the 8 base patterns)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

FS = 2.4e6


def _base_patterns(orc, P, N, seed=0x5D2):
    x = np.zeros((P, N, 2), np.int16)
    for c in range(P):
        g = orc.IQSigGen(FS, [(50e3 + 97 * c, 7000, 0.1 * c), (-200e3 - 53 * c, 5000, 0.1 * c)])
        s = g.next_cs16(N).astype(np.int32)
        s += np.random.default_rng(seed + c).integers(-64, 65, size=s.shape)
        x[c] = s.astype(np.int16)
    return x


@pytest.mark.parametrize("epi_name", ["usb", "fm"])
def test_iqbb_8192_channels_one_gpu(orc, epi_name):
    """IQBaseBand<int16>(127 taps, /8) -> USBDemod (config 5's chain) and -> FMDemod (the north-star chain) on 8192 channels
    x 65536 samples x 2 calls: the 8 base patterns against the oracle, every other channel against its pattern (batching
    invariance) — compared on the device."""
    import torch
    import libsdr_amd as sa
    C, N, D, P = 8192, 65536, 8, 8
    if torch.cuda.get_device_properties(0).total_memory < 16 * 2**30:
        pytest.skip("needs 6 GiB of device memory")
    dev = torch.device("cuda:0")
    epi = sa.EPI_USB if epi_name == "usb" else sa.EPI_FM
    taps = sa.design_iqbb_taps(100e3, 50e3, FS, 127)
    lut = sa.design_freqshift_lut_i16()
    base = _base_patterns(orc, P, 2 * N)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = sa.Context(0, stream=stream.cuda_stream)
        xb = torch.from_numpy(base).to(dev)                       # [P, 2N, 2]
        x = xb.repeat(C // P, 1, 1).contiguous()                   # [C, 2N, 2]: channel c carries pattern c % P; 4 GiB
        assert x.numel() * 2 == 2 * C * N * 4 and x.data_ptr() % 4 == 0
        node = sa.IQBaseBandI16(ctx, taps, lut, 1365, False, D, channels=C, max_in=N, epilogue=epi)
        n_max = N // D + 1
        outs = [torch.full((C, n_max), -12345, dtype=torch.int16, device=dev) for _ in range(2)]
        got = []
        for i in range(2):   # call i reads columns [i*N, (i+1)*N) of every row: row stride 2N samples = 512 KiB
            k = node.process_dev(x.data_ptr() + i * N * 4, N, 2 * N, outs[i].data_ptr(), n_max)
            got.append(k)
        ctx.synchronize()
        torch.cuda.synchronize()
        assert got == [N // D - 1, N // D]
        for i in range(2):
            y = outs[i][:, :got[i]]
            assert bool((outs[i][:, got[i]:] == -12345).all())     # nothing written beyond the call's outputs
            yp = y.view(C // P, P, got[i])
            assert bool((yp == yp[0:1]).all()), "call %d: a channel differs from its pattern's first instance" % i
        heads = [outs[i][:P, :got[i]].cpu().numpy() for i in range(2)]
        tail = [outs[i][C - P:, :got[i]].cpu().numpy() for i in range(2)]   # the rows beyond 2^32 bytes of input
        del node
    for k in range(P):
        bb, fm = orc.IQBaseBandI16(taps, lut, 1365, False, D), orc.FMDemodI16()
        for i in range(2):
            r = bb.process(base[k, i * N:(i + 1) * N])
            r = orc.usb_i16(r) if epi == sa.EPI_USB else fm.process(r)
            assert np.array_equal(heads[i][k], r), (epi_name, k, i)
            assert np.array_equal(tail[i][k], r), (epi_name, "last rows", k, i)
