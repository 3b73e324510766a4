"""world_size-2 `gloo` test of the N>1 path (CPU): design broadcast + channel sharding + output gather
must reproduce the single-process result row for row. The per-rank compute is the oracle here (no GPU in
this container); on the GPU box the same shard.py helpers drive the HIP nodes under RCCL (bench.py)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS, C, N = 2.4e6, 6, 2048


def _inputs(orc):
    return np.stack([orc.IQSigGen(FS, [(50e3 + 970 * c, 7000, 0.1 * c), (-200e3 - 530 * c, 5000, 0.0)]).next_cs16(2 * N)
                     for c in range(C)])


def _run_channels(orc, taps, lut, inc, x):
    outs = []
    for c in range(x.shape[0]):
        bb, fm = orc.IQBaseBandI16(taps, lut, inc, False, 8), orc.FMDemodI16()
        outs.append(np.concatenate([fm.process(bb.process(x[c, k * N:(k + 1) * N])) for k in range(2)]))
    return np.stack(outs)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libsdr_amd import shard
    from oracle import pyoracle as orc
    # rank 0 designs; the others start from garbage and must end up with rank 0's numbers
    if rank == 0:
        taps = torch.from_numpy(orc.iqbb_design(100e3, 50e3, FS, 127).copy())
        lut = torch.from_numpy(orc.freqshift_lut_i16().copy())
    else:
        taps = torch.full((127, 2), -7, dtype=torch.int32)
        lut = torch.zeros((128, 2), dtype=torch.int32)
    shard.broadcast_design([taps, lut], src=0)
    lo, hi = shard.shard_range(C, world, rank)
    x = _inputs(orc)[lo:hi]
    y = torch.from_numpy(_run_channels(orc, taps.numpy(), lut.numpy(), 1365, x))
    full, _ = shard.gather_output(y, C, dst=0)
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges():
    from libsdr_amd import shard
    for total, world in ((8192, 8), (1024, 3), (5, 8), (6, 2)):
        r = [shard.shard_range(total, world, k) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == total
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1


def test_two_rank_gloo_matches_single_process(orc):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _run_channels(orc, orc.iqbb_design(100e3, 50e3, FS, 127), orc.freqshift_lut_i16(), 1365, _inputs(orc))
    assert got.shape == ref.shape and np.array_equal(got, ref)
