"""Red-zoned device arena for the GPU tests (guard-band canaries around device buffers) — TEST CODE: it plugs
into libsdr_amd.nodes.device_router, the seam process() offers to callers that bring their own device buffers."""
import ctypes as C

import numpy as np

from libsdr_amd import abi, nodes


class RedZone:
    """Test aid (guard-band canaries around device buffers). While `RedZone.active` is set, every node's
    process(x) runs through the DEVICE-pointer entry point (`*_process_dev`) on buffers cut from a red-zoned arena:
    a pattern-filled guard band before the first row, between any two rows (the row stride is the row length plus the
    band) and after the last one, for the input and for the output. After the call both arenas are read back whole:
    every guard byte must still hold the pattern and the input rows must be untouched — an out-of-bounds WRITE that
    lands in a neighbour row's not-yet-compared region, which an equality test of the outputs cannot see, fails here.
    The odd band width also walks the rows over every alignment the 16-byte loads and LDS-DMA windows can meet."""
    active = False
    band = 37            # samples (elements of the row type)
    calls = 0            # guarded calls made (tests assert that the fixture was live)
    IN_PAT, OUT_PAT = 0xA5, 0x5A

    _cache = {}          # (context handle, which) -> (device pointer, bytes): grow-only, one pair per context

    @staticmethod
    def _buf(ctx, which, nbytes):
        key = (ctx.handle.value, which)
        p, cap = RedZone._cache.get(key, (0, 0))
        if cap < nbytes:
            if p:
                ctx.free(p)
            cap = max(nbytes, 2 * cap, 1 << 16)
            p = ctx.malloc(cap)
            RedZone._cache[key] = (p, cap)
        return p

    @staticmethod
    def run(ctx, x, out, call):
        """x: [rows_in, n_in, ...] host array, out: [rows_out, n_out, ...] host array to fill;
        call(in_ptr, in_stride, out_ptr, out_stride) -> launches on device pointers, strides in row elements."""
        g = RedZone.band

        def arena(a, pat):
            rows, n = a.shape[0], a.shape[1]
            eb = a.dtype.itemsize * int(np.prod(a.shape[2:], dtype=np.int64))   # bytes per row element
            h = np.full(((rows * (n + g) + g) * eb,), pat, np.uint8)
            return h, rows, n, eb

        hin, ri, ni, ebi = arena(x, RedZone.IN_PAT)
        rows_in = hin[g * ebi:].reshape(-1)[: ri * (ni + g) * ebi].reshape(ri, (ni + g) * ebi)
        rows_in[:, : ni * ebi] = np.ascontiguousarray(x).view(np.uint8).reshape(ri, ni * ebi)
        hout, ro, no, ebo = arena(out, RedZone.OUT_PAT)
        # (the output rows start out as the caller's array does: a node that leaves an element alone — FMDemod never
        # writes index 0 — leaves the caller's value there)
        hout[g * ebo:].reshape(-1)[: ro * (no + g) * ebo].reshape(ro, (no + g) * ebo)[:, : no * ebo] = np.ascontiguousarray(out).view(np.uint8).reshape(ro, no * ebo)
        din, dout = RedZone._buf(ctx, 0, hin.nbytes), RedZone._buf(ctx, 1, hout.nbytes)
        ctx.h2d(din, hin)
        ctx.h2d(dout, hout)
        call(din + g * ebi, ni + g, dout + g * ebo, no + g)
        ctx.synchronize()
        bin_, bout = np.empty_like(hin), np.empty_like(hout)
        ctx.d2h(bin_, din)
        ctx.d2h(bout, dout)
        RedZone.calls += 1
        if not np.array_equal(bin_, hin):
            bad = np.flatnonzero(bin_ != hin)
            raise AssertionError("red zone: the call wrote into its INPUT arena (%d bytes, first at byte %d of %d)" % (bad.size, bad[0], hin.size))
        rows_out = bout[g * ebo:].reshape(-1)[: ro * (no + g) * ebo].reshape(ro, (no + g) * ebo)
        guard = np.concatenate([bout[: g * ebo], rows_out[:, no * ebo:].reshape(-1)])
        if np.any(guard != RedZone.OUT_PAT):
            bad = np.flatnonzero(guard != RedZone.OUT_PAT)
            raise AssertionError("red zone: %d guard bytes of the OUTPUT arena were overwritten (rows of %d elements, band %d; first at guard byte %d)"
                                 % (bad.size, no, g, bad[0]))
        out.view(np.uint8).reshape(ro, no * ebo)[...] = rows_out[:, : no * ebo]
        return out

    @classmethod
    def install(cls, on):
        """Route every node's process() through the arena (on) or back through the host-pointer entry points (off)."""
        cls.active = bool(on)
        nodes.device_router = cls.run if on else None
        if cls._close not in nodes.close_hooks:
            nodes.close_hooks.append(cls._close)

    @staticmethod
    def _close(ctx):
        for key in [k for k in RedZone._cache if k[0] == ctx.handle.value]:
            abi.lib().sdrhip_free(ctx.handle, C.c_void_p(RedZone._cache.pop(key)[0]))
