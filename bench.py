#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native libsdr hot path.

A "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
per GPU `--channels` (default 1024) independent complex<int16> IQ channels x `--samples` (65536)
samples through IQBaseBand<int16>(127-tap Q14 FIR -> LUT shift -> /8) -> FMDemod, i.e. the
north-star chain of BASELINE.json on the per-GPU shard of its config 5 (8192 channels over 8 GPUs).
Channels are independent, so ranks shard them with no data-path collective (weak scaling);
taps/LUT are designed on rank 0 and broadcast over RCCL at config time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload iqbb_fm|iqbb_usb|iqbb_fm_cu8|bb_real_fm|fir255_fm|fbb_f32|fftconv|fftbank|fm_demod|subsample8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     achieved algorithmic HBM bytes/s of the dominant kernel vs 8 TB/s, from HIP events
                  recorded on the stream the kernel runs on;
  "cpu_baseline": the reference CPU path (oracle/_ref/ref_driver, the unmodified reference compiled
                  here) or the oracle port, timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
FS = 2.4e6


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=400)    # ~60 ms timed at the headline workload (a few-ms burst shows a clock the chip does not hold)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--channels", type=int, default=1024, help="channels per GPU")
    p.add_argument("--samples", type=int, default=65536, help="samples per channel per step")
    p.add_argument("--workload", default="iqbb_fm")
    p.add_argument("--decim", type=int, default=8, help="iqbb_* workloads: decimation D (8 = the BASELINE configs)")
    p.add_argument("--order", type=int, default=127, help="iqbb_* workloads: FIR order (127 = the BASELINE configs)")
    p.add_argument("--batches", type=int, default=3, help="distinct input batches rotated through (defeats the 256 MiB L3)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for single-GPU dry runs)")
    p.add_argument("--force-device", type=int, default=-1, help="dry runs only: put every rank on this device")
    p.add_argument("--gather", action="store_true", help="also gather the demodulated output on rank 0 every step (RCCL)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    p.add_argument("--sustain-seconds", type=float, default=2.0,
                   help="after the timed steps: this many seconds of back-to-back launches for the sustained-clock figure (0 = skip)")
    p.add_argument("--dump-output", default="", help="rank 0 saves the last step's (gathered) output rows as .npy (tests)")
    return p.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD `torch.distributed.run`
    (one process per GPU, rendezvous on 127.0.0.1) and hand back its exit code. This runs before torch or
    HIP is touched in this process, and it is a child process, never an exec."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env, cwd=ROOT)


def synth_cs16(torch, C, N, dev, seed, chan0=0):
    """Two tones per channel + small integer noise (SURVEY §8d config 3/5 recipe), generated on the GPU."""
    n = torch.arange(N, device=dev, dtype=torch.float64) / FS
    ni = torch.arange(N, device=dev, dtype=torch.int64)[None, :]
    M32 = 0xFFFFFFFF
    out = torch.empty((C, N, 2), dtype=torch.int16, device=dev)
    step = 64
    for c0 in range(0, C, step):
        c = torch.arange(c0, min(C, c0 + step), device=dev, dtype=torch.float64) + chan0
        f1 = (50e3 + 97.0 * c)[:, None]
        f2 = (-200e3 - 53.0 * c)[:, None]
        ph = (0.1 * c)[:, None]
        a1 = 2 * torch.pi * f1 * n[None, :] + ph
        a2 = 2 * torch.pi * f2 * n[None, :] + ph
        re = torch.trunc(3500.0 * torch.cos(a1)) + torch.trunc(2500.0 * torch.cos(a2))
        im = torch.trunc(3500.0 * torch.sin(a1)) + torch.trunc(2500.0 * torch.sin(a2))
        # integer noise in [-64, 64]: a hash of (global channel, sample, seed), so a channel's stream does not
        # depend on which rank generates it (the 2-rank test compares with the single-process run bit for bit)
        h = (c.to(torch.int64)[:, None] * 0x9E3779B1 + ni * 0x85EBCA77 + seed * 0xC2B2AE3D) & M32
        h = ((h ^ (h >> 15)) * 0x2C1B3C6D) & M32
        h = ((h ^ (h >> 12)) * 0x297A2D39) & M32
        h = h ^ (h >> 15)
        out[c0:c0 + re.shape[0], :, 0] = (re.to(torch.int64) + (h & 0xFFFF) % 129 - 64).to(torch.int16)
        out[c0:c0 + re.shape[0], :, 1] = (im.to(torch.int64) + (h >> 16) % 129 - 64).to(torch.int16)
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, target_s):
    """Reference CPU path on this box: the compiled, unmodified reference if oracle/_ref travelled here,
    else the oracle port. One thread = the reference's real execution model (one Queue worker)."""
    chain = {"iqbb_fm": "iqbb_fm", "iqbb_usb": "iqbb_usb", "fir255_fm": "fir255_fm", "fir127_fm": "fir127_fm",
             "fbb_f32": "fir_cf32_sub8"}.get(workload)
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    cores_avail = os.cpu_count()
    if chain and os.path.exists(ref):
        try:
            probe = json.loads(subprocess.run([ref, "bench", chain, "8"], capture_output=True, text=True, timeout=120).stdout)
            nbuf = max(8, int(target_s * probe["msps"] * 1e6 / 65536))
            r = json.loads(subprocess.run([ref, "bench", chain, str(nbuf)], capture_output=True, text=True, timeout=600).stdout)
            res = {"value": round(r["msps"], 4), "unit": "Msamples/s", "cores": 1, "kind": "reference",
                   "sample": "%d buffers x 65536 cs16 samples, 1 channel, chain %s (reference nodes compiled -O3, "
                             "%.1f s)" % (nbuf, chain, r["seconds"]), "host_cores_available": cores_avail, "cpu_model": cpu_model()}
            # one channel (= one reference graph) per host core, all cores at once, ~5 s
            try:
                try:
                    ncore = len(os.sched_getaffinity(0))
                except AttributeError:
                    ncore = cores_avail or 1
                ncore = max(1, min(ncore, 32))   # (the GPU boxes expose 256 CPUs under a much smaller CPU quota)
                nb = max(8, int(3.0 * probe["msps"] * 1e6 / 65536))
                t0 = time.perf_counter()
                procs = [subprocess.Popen([ref, "bench", chain, str(nb)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                         for _ in range(ncore)]
                outs = [pr.communicate(timeout=600)[0] for pr in procs]
                wall = time.perf_counter() - t0
                done = sum(json.loads(o)["samples"] for o in outs if o.strip())
                res["all_cores"] = {"value": round(done / wall / 1e6, 2), "unit": "Msamples/s", "cores": ncore,
                                    "sample": "%d independent reference graphs (processes) x %d buffers, wall %.1f s" % (ncore, nb, wall)}
            except Exception as e:
                res["all_cores"] = {"error": str(e)[:80]}
            return res
        except Exception as e:   # fall through to the port
            sys.stderr.write("cpu_baseline: reference binary failed (%s), using the port\n" % e)
    if workload not in ("iqbb_fm",):
        return None
    from oracle import pyoracle as orc   # bench.py's cpu_baseline leg may use the oracle
    taps = orc.iqbb_design(100e3, 50e3, FS, 127)
    lut = orc.freqshift_lut_i16()
    x = orc.IQSigGen(FS, [(100e3, 8000, 0.0), (-300e3, 6000, 0.3)]).next_cs16(65536)
    sec = orc.bench_iqbb_fm(taps, lut, 1365, False, 8, x, 8)
    nbuf = max(8, int(target_s / (sec / 8)))
    sec = orc.bench_iqbb_fm(taps, lut, 1365, False, 8, x, nbuf)
    return {"value": round(nbuf * 65536 / sec / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d buffers x 65536 cs16 samples, 1 channel, IQBaseBand(127,/8)->FM oracle port (%.1f s)" % (nbuf, sec),
            "host_cores_available": cores_avail}


def measured_traffic(kernels):
    """Per-step HBM bytes of the step's kernels (one step = one launch of each) from the committed rocprofv3 PMC
    passes (profiles/*_pmc.json, newest round first; collected in separate --pmc runs of this same command and
    corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 on gfx950, KiB units). None if no profile holding
    every one of these kernels is committed."""
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            js = json.load(open(fn))
            ds = [js.get(k, {}).get("derived") for k in kernels]
            if all(ds):
                return {"bytes": sum(d["hbm_traffic_bytes_per_launch"] for d in ds), "source": os.path.basename(fn)}
        except Exception:
            pass
    return None


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a))
    import torch
    import torch.distributed as dist
    import libsdr_amd as sa
    from libsdr_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if a.force_device >= 0:
        local = a.force_device
    if world != a.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; reporting the ranks that run\n" % (a.gpus, world))
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants device %d but only %d visible (dry runs: --backend gloo --force-device 0)"
                         % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(a.backend)
    C, N, W, K = a.channels, a.samples, a.warmup, a.steps

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = sa.Context(local, stream=stream.cuda_stream)

        # ---- config(): design on rank 0, broadcast over RCCL (KBs; outside the timed region) ----
        order, D = a.order, a.decim
        wl = a.workload
        if wl in ("iqbb_fm", "iqbb_usb", "iqbb_fm_cu8"):
            taps = torch.from_numpy(sa.design_iqbb_taps(100e3, 50e3, FS, order)).to(dev)
            lut = torch.from_numpy(sa.design_freqshift_lut_i16()).to(dev)
            shard.broadcast_design([taps, lut], src=0)
            node = sa.IQBaseBandI16(ctx, taps.cpu().numpy(), lut.cpu().numpy(), sa.design_freqshift_inc(100e3, FS), False, D,
                                    channels=C, max_in=N, epilogue=sa.EPI_USB if wl == "iqbb_usb" else sa.EPI_FM)
            in_bytes, alg_bytes = 4.0, 4.0 + 2.0 / D
            n_out = node.out_count(N) + 1
            outs = torch.zeros((C, n_out), dtype=torch.int16, device=dev)
            ins = [synth_cs16(torch, C, N, dev, 1234 + b, chan0=rank * C) for b in range(a.batches)]
            if wl == "iqbb_fm_cu8":   # RTL-SDR bytes: the same signal as offset-binary complex<uint8>, AutoCast fused into the load
                node.set_input_format(sa.abi.IN_CU8)
                ins = [((x.to(torch.int32) >> 6) + 127).clamp_(0, 255).to(torch.uint8) for x in ins]
                in_bytes, alg_bytes = 2.0, 2.0 + 2.0 / D
            run = lambda b: node.process_dev(ins[b].data_ptr(), N, N, outs.data_ptr(), n_out)
            dtype = "i16"
            kernels = node.kernel_names   # dominant first; path 1 on cs16 input = hot kernel + the small border launch
            kernel = kernels[0]
            desc = "IQBaseBand<int16>(%d-tap Q14 FIR, LUT shift 100 kHz, /%d) -> %s" % (order, D, "USBDemod" if wl == "iqbb_usb" else "FMDemod")
            if wl == "iqbb_fm_cu8":
                desc = "complex<uint8> -> AutoCast + " + desc
        elif wl == "bb_real_fm":   # the real-input BaseBand<int16_t> (2 bytes per sample in)
            taps = sa.design_bb_taps(100e3, 50e3, FS, order)
            node = sa.BaseBandI16(ctx, taps, sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(100e3, FS), False, D,
                                  channels=C, max_in=N, epilogue=sa.EPI_FM)
            in_bytes, alg_bytes = 2.0, 2.0 + 2.0 / D
            n_out = node.out_count(N) + 1
            outs = torch.zeros((C, n_out), dtype=torch.int16, device=dev)
            ins = [synth_cs16(torch, C, N, dev, 1234 + b, chan0=rank * C)[..., 0].contiguous() for b in range(a.batches)]
            run = lambda b: node.process_dev(ins[b].data_ptr(), N, N, outs.data_ptr(), n_out)
            dtype = "i16"
            kernels = node.kernel_names
            kernel = kernels[0]
            desc = "BaseBand<int16> real input (%d-tap Q16 FIR, LUT shift 100 kHz, /%d) -> FMDemod" % (order, D)
        elif wl in ("fir255_fm", "fir127_fm"):
            order = 255 if wl == "fir255_fm" else 127
            alpha = torch.from_numpy(sa.design_fir_lowpass(order, 100e3, FS)).to(dev)
            shard.broadcast_design([alpha], src=0)
            node = sa.FIR(ctx, sa.FIR_CS16_EXACT, alpha.cpu().numpy(), channels=C, max_in=N, epilogue=sa.EPI_FM)
            in_bytes, alg_bytes = 4.0, 6.0
            outs = torch.zeros((C, N), dtype=torch.int16, device=dev)
            ins = [synth_cs16(torch, C, N, dev, 1234 + b, chan0=rank * C) for b in range(a.batches)]
            run = lambda b: node.process_dev(ins[b].data_ptr(), N, N, outs.data_ptr(), N)
            dtype, kernel = "f64", "fir_cs16_exact_kernel"
            kernels = [kernel]
            desc = "FIRLowPass<complex<int16>>(%d taps, exact per-tap truncation) -> FMDemod" % order
        elif wl == "fbb_f32":
            alpha = sa.design_fir_lowpass(127, 100e3, FS)
            node = sa.FloatBaseBand(ctx, 100e3, FS, alpha, 8, channels=C, max_in=N)
            in_bytes, alg_bytes = 8.0, 9.0
            n_out = N // 8 + 1
            outs = torch.zeros((C, n_out, 2), dtype=torch.float32, device=dev)
            ins = [torch.randn((C, N, 2), dtype=torch.float32, device=dev) * 0.3 for b in range(a.batches)]
            run = lambda b: node.process_dev(ins[b].data_ptr(), N, N, outs.data_ptr(), n_out)
            dtype, kernel = "f32", "fir_cf32_rt_kernel"
            kernels = [kernel]
            desc = "float baseband: shift 100 kHz -> FIRLowPass<cf32>(127) -> /8"
        elif wl == "fftbank":   # FilterNode<float>: 4 bands behind ONE forward transform per block (2048-point, overlap-add)
            import numpy as np
            bands = [(50e3, 150e3), (-350e3, -250e3), (200e3, 300e3), (-120e3, -20e3)]
            Ks = [sa.design_fftfilt_spectrum(sa.design_fftfilt_kernel(1024, lo_, hi_, FS)) for lo_, hi_ in bands]
            node = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2048, Ks, channels=C, max_in=N)
            in_bytes, alg_bytes = 8.0, 8.0 + 8.0 * len(bands)
            outs = torch.zeros((len(bands), C, N, 2), dtype=torch.float32, device=dev)
            ins = [torch.randn((C, N, 2), dtype=torch.float32, device=dev) * 0.3 for b in range(a.batches)]
            run = lambda b: node.process_dev(ins[b].data_ptr(), N, N, outs.data_ptr(), N)
            dtype, kernel = "f32", "fftconv_fused_kernel"
            kernels = [kernel]
            desc = "FFT filter bank: 2048-point overlap-add, 1024-sample blocks, %d bands behind one forward transform" % len(bands)
        elif wl == "fftconv":
            alpha = sa.design_fir_lowpass(4097, 100e3, FS)
            import numpy as np
            tapsf = np.stack([alpha[::-1], np.zeros_like(alpha)], 1).astype(np.float32)   # h[k] = alpha[order-1-k]
            if N == 65536:
                N = 6 * 12288   # whole blocks per call (hop = 16384 - 4096): a ragged last block is a full transform for a third of a hop
            node = sa.FFTConv(ctx, sa.FFTCONV_OLS, 16384, tapsf, channels=C, max_in=N)
            in_bytes, alg_bytes = 8.0, 16.0
            outs = torch.zeros((C, N, 2), dtype=torch.float32, device=dev)
            ins = [torch.randn((C, N, 2), dtype=torch.float32, device=dev) * 0.3 for b in range(a.batches)]
            run = lambda b: node.process_dev(ins[b].data_ptr(), N, N, outs.data_ptr(), N)
            dtype, kernel = "f32", "fftconv_fused_kernel"
            kernels = [kernel]
            desc = "FFT convolution, overlap-save L=16384, 4097 taps (hop 12288)"
        elif wl in ("fm_demod", "subsample8"):
            ins = [synth_cs16(torch, C, N, dev, 1234 + b, chan0=rank * C) for b in range(a.batches)]
            if wl == "fm_demod":
                node = sa.Demod(ctx, sa.EPI_FM, sa.T_CS16, channels=C, max_in=N)
                in_bytes, alg_bytes = 4.0, 6.0
                outs = torch.zeros((C, N), dtype=torch.int16, device=dev)
                run = lambda b: node.process_dev(ins[b].data_ptr(), N, N, outs.data_ptr(), N)
                kernel, desc = "demod_cs16_kernel", "FMDemod<int16> alone (complex<int16> -> int16)"
            else:
                node = sa.SubSample(ctx, sa.T_CS16, 8, channels=C, max_in=N)
                in_bytes, alg_bytes = 4.0, 4.5
                outs = torch.zeros((C, N // 8 + 1, 2), dtype=torch.int16, device=dev)
                run = lambda b: node.process_dev(ins[b].data_ptr(), N, N, outs.data_ptr(), N // 8 + 1)
                kernel, desc = "subsample8_cs16_kernel", "SubSample<complex<int16>>(8) alone"
            dtype = "i16"
            kernels = [kernel]
        else:
            raise SystemExit("unknown workload " + wl)

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for i in range(W):
            run(i % a.batches)
        barrier()
        timer = sa.Timer(ctx)
        gathered = None
        t0 = time.perf_counter()
        timer.start()
        for i in range(K):
            run(i % a.batches)
            if a.gather and world > 1:
                gathered, _ = shard.gather_output(outs, C * world, dst=0)
        timer.stop()
        barrier()
        wall = time.perf_counter() - t0
        dev_ms = timer.elapsed_ms()
        if a.dump_output and rank == 0:
            import numpy as np
            torch.cuda.synchronize()
            np.save(a.dump_output, (gathered if gathered is not None else outs).cpu().numpy())

        # ---- sustained figure: >= --sustain-seconds of back-to-back launches on the same stream, so that the
        # clock the chip HOLDS under this load (DVFS) is what is measured, not a few-ms burst (every rank runs it;
        # rank 0 reports its own) ----
        sustained = None
        if a.sustain_seconds > 0:
            chunk = max(K, 20)
            tot_ms, launches, t1 = 0.0, 0, time.perf_counter()
            while time.perf_counter() - t1 < a.sustain_seconds:
                timer.start()
                for i in range(chunk):
                    run(i % a.batches)
                timer.stop()
                tot_ms += timer.elapsed_ms()   # waits for the chunk; the next one follows within microseconds
                launches += chunk
            last_ms = timer.elapsed_ms() / chunk
            sustained = {"ms_per_launch": tot_ms / launches, "launches": launches, "last_chunk_ms_per_launch": last_ms}
            barrier()

    host_coll = world > 1 and a.backend != "nccl"
    wall_t = torch.tensor([wall], dtype=torch.float64, device="cpu" if host_coll else dev)
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())

    if rank == 0:
        total_samples = float(C) * N * K * world
        value = total_samples / wall / 1e6
        per_launch_s = dev_ms / 1e3 / K
        achieved = C * N * alg_bytes / per_launch_s / 1e9
        res = {
            "metric": "Msamples/s through baseband->FIR->demod chain",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(wall / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": desc, "channels_per_gpu": C, "samples_per_channel_per_step": N,
                       "global_channels": C * world, "input": "int16 (real)" if wl == "bb_real_fm" else {2.0: "complex<uint8>", 4.0: "complex<int16>"}.get(in_bytes, "complex<float>"),
                       "parallelism": "channel-sharded x%d, %s" % (world, "output gathered on rank 0 per step (RCCL)" if a.gather else "no data-path collective")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "kernel": kernel, "kernels_per_step": kernels,
                         "algorithmic_bytes_per_launch": C * N * alg_bytes,
                         "algorithmic_bytes_per_sample": alg_bytes, "avg_launch_ms": round(per_launch_s * 1e3, 4),
                         "hbm_read_frac": round(C * N * in_bytes / per_launch_s / 1e9 / HBM_PEAK_GBS, 5),
                         "per_gpu_msamples_s": round(C * N / per_launch_s / 1e6, 2),
                         "ranks_seen": dist.get_world_size() if world > 1 else 1},
        }
        if sustained:
            rf = res["roofline"]
            rf["sustained_ms_per_launch"] = round(sustained["ms_per_launch"], 4)
            rf["sustained_frac"] = round(C * N * alg_bytes / (sustained["ms_per_launch"] / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
            rf["sustained_launches"] = sustained["launches"]
            rf["sustained_last_chunk_ms_per_launch"] = round(sustained["last_chunk_ms_per_launch"], 4)
            rf["sustained_per_gpu_msamples_s"] = round(C * N / (sustained["ms_per_launch"] / 1e3) / 1e6, 2)
        if rank == 0:   # what this box's HBM delivers to a pure read of the same buffers, beside the nominal peak
            try:
                import ctypes
                gbs = ctypes.c_double(0.0)
                buf = torch.ones(1 << 28, dtype=torch.int32, device=dev)   # 1 GiB: four times the 256 MiB Infinity Cache
                rc = sa.abi.lib().sdrhip_bench_stream_read(ctx.handle, ctypes.c_void_p(buf.data_ptr()), buf.numel() * buf.element_size(),
                                                           5, ctypes.byref(gbs))
                del buf
                if rc == 0 and gbs.value > 0:
                    res["roofline"]["stream_read_gbs"] = round(gbs.value, 1)
                    res["roofline"]["frac_of_stream_read"] = round(achieved / gbs.value, 5)
            except Exception as e:   # measurement aid only
                res["roofline"]["stream_read_error"] = str(e)[:80]
        tr = measured_traffic(kernels) if (C, N) == (1024, 65536) else None
        if tr:
            res["roofline"]["traffic"] = tr["bytes"]
            res["roofline"]["traffic_source"] = "profiles/" + tr["source"]
        if world == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline(wl, a.cpu_seconds)
            if cb:
                res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
