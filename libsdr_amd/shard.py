"""Channel sharding across the GPUs of one node: independent IQ channels are split into
contiguous blocks, one process per GPU; the only exchanges are a broadcast of the read-only design
(taps / LUT / FFT kernel, a few KB, at config time) and an optional gather of demodulated output on a
root. `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm; the same code runs on "gloo" (CPU)
in tests/test_dist_gloo.py. No compute lives here."""
import torch
import torch.distributed as dist


def shard_range(total_channels, world, rank):
    """Contiguous block [lo, hi) of channels owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total_channels, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _host_staged(t):
    """gloo (CPU dry runs, tests) moves host memory only: device tensors are staged through the host there.
    On RCCL ("nccl") device tensors travel as they are, over xGMI."""
    return t.is_cuda and dist.get_backend() != "nccl"


def broadcast_design(tensors, src=0):
    """Broadcasts design tensors (designed on `src`) so that every rank filters with bit-identical taps."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            if _host_staged(t):
                h = t.cpu()
                dist.broadcast(h, src)
                t.copy_(h)
            else:
                dist.broadcast(t, src)
    return tensors


def gather_output(local, total_channels, dst=0, out=None):
    """Gathers per-rank output rows [channels_local, n] onto `dst` as [total_channels, n] (row order =
    global channel order). Ranks may own different numbers of channels. `out` (on `dst`): a preallocated
    [total_channels, n] tensor the rows land in. Returns (tensor_or_None, None). This is the blocking form (tests, gloo
    dry runs); bench.py's RCCL path issues `dist.gather(..., async_op=True)` on a side stream straight into row views
    of its preallocated tensor."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        if out is not None:
            out.copy_(local)
            return out, None
        return local, None
    rank = dist.get_rank()
    dtype = local.dtype
    dev = local.device
    if _host_staged(local):
        local = local.cpu()
    if dtype == torch.int16:   # gloo has no int16 collectives; bytes travel the same on RCCL
        local = local.contiguous().view(torch.uint8)
    sizes = [shard_range(total_channels, world, r) for r in range(world)]
    if rank == dst:
        parts = [torch.empty((hi - lo,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device) for lo, hi in sizes]
        dist.gather(local, parts, dst=dst)
        full = torch.cat(parts, 0).view(dtype)
        if out is not None:
            out.copy_(full.to(out.device))
            return out, None
        return full.to(dev), None
    dist.gather(local, None, dst=dst)
    return None, None
