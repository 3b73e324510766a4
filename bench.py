#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native libsdr hot path.

A "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
per GPU `--channels` (default 1024) independent complex<int16> IQ channels x `--samples` (65536)
samples through IQBaseBand<int16>(127-tap Q14 FIR -> LUT shift -> /8) -> FMDemod, i.e. the
north-star chain of BASELINE.json on the per-GPU shard of its config 5 (8192 channels over 8 GPUs).
Channels are independent, so ranks shard them with no data-path collective (weak scaling);
taps/LUT are designed on rank 0 and broadcast over RCCL at config time.

With more than one GPU (`--gpus N`, N > 1, no --workload given) the line is BASELINE config 5 as SURVEY §8d
states it: the same baseband with the USB (SSB) demodulator, and the demodulated output of every rank GATHERED
on rank 0 inside the timed region — from a double-buffered output, on a side stream, straight into one
preallocated [N * channels, n_out] tensor (`roofline.without_gather_*` = the same steps without the gather).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload iqbb_fm|iqbb_usb|iqbb_fm_cu8|bb_real_fm|fir255_fm|
                     fir127_fm|fbb_f32|fftconv|fftbank|fm_demod|subsample8|iqbb_fm_cs8] [--order 127] [--no-verify]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     achieved algorithmic HBM bytes/s of the dominant kernel vs 8 TB/s, from HIP events
                  recorded on the stream the kernel runs on; `traffic` = PMC bytes of THIS workload's committed profile;
  "cpu_baseline": the reference CPU path (oracle/_ref/ref_driver, the unmodified reference compiled
                  here) or the oracle port, timed on this box's host cores on a bounded sample;
  "verified":     the LAST timed step's output of 32 randomly chosen channels compared with the CPU oracle (outside the
                  timed region): bit-exact for the int16 paths, max|y - ref| / max|ref| <= 1e-5 for float / FFT.
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
INT8_MFMA_PEAK_TOPS = 5000.0   # dense int8 MFMA (MI355X_MICROARCH.md: I8 = 2x the BF16 rate per clock, BF16 ~2.5 PF dense)
DP_ISSUE_PEAK_TOPS = 35.2      # non-FMA fp64 vector instructions: v_mul/add/trunc_f64 issue at 4.2 cycles per wave and SIMD and
                               # v_mul_f64 holds 2.10 GHz (tools/probes/dp_rates.hip) = 23 GS/s x 6 x 255 slots
FS = 2.4e6
RTOL = 1e-5


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=400)    # ~50 ms timed at the headline workload (a few-ms burst shows a clock the chip does not hold)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--channels", type=int, default=1024, help="channels per GPU")
    p.add_argument("--global-channels", type=int, default=0,
                   help="STRONG scaling: this many channels in all, a contiguous block of G / N per GPU (BASELINE config 5: 8192); "
                        "overrides --channels and the line says \"scaling\": \"strong\"")
    p.add_argument("--samples", type=int, default=65536, help="samples per channel per step")
    p.add_argument("--workload", default="", help="default: iqbb_fm on one GPU, iqbb_usb with the gather (BASELINE config 5) on several")
    p.add_argument("--decim", type=int, default=8, help="iqbb_* workloads: decimation D (8 = the BASELINE configs)")
    p.add_argument("--fc", type=float, default=100e3, help="iqbb_* workloads: centre and filter frequency in Hz (0: no frequency shift, as examples/sdr_rec.cc tunes)")
    p.add_argument("--deemph", action="store_true", help="iqbb_fm* workloads: FMDeemph<int16> behind the demodulator (examples/sdr_fm.cc:44-53), alpha from the output rate")
    p.add_argument("--order", type=int, default=127, help="iqbb_* workloads: FIR order (127 = the BASELINE configs)")
    p.add_argument("--fs", type=float, default=FS, help="iqbb_* workloads: input sample rate the filter is designed for")
    p.add_argument("--width", type=float, default=50e3, help="iqbb_* workloads: filter width in Hz")
    p.add_argument("--buffers", type=int, default=1,
                   help="iqbb_* / bb_real_fm workloads: this many reference-sized buffers of --samples per channel in ONE launch "
                        "(sdrhip_iqbb_i16_process_dev_multi: the buffer boundaries kept); a step is then --buffers buffers per channel")
    p.add_argument("--batches", type=int, default=3, help="distinct input batches rotated through (defeats the 256 MiB L3)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for single-GPU dry runs)")
    p.add_argument("--force-device", type=int, default=-1, help="dry runs only: put every rank on this device")
    p.add_argument("--gather", action="store_true", help="gather the demodulated output on rank 0 every step (default with --gpus > 1 and no --workload)")
    p.add_argument("--no-gather", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    p.add_argument("--sustain-seconds", type=float, default=2.0,
                   help="BEFORE the warm-up and the timed steps: this many seconds of back-to-back launches, so that the timed region "
                        "starts at the clock the chip holds under this load; also the sustained figure (0 = skip)")
    p.add_argument("--no-configs", action="store_true",
                   help="default one-GPU run: do not append the other BASELINE configs (2, 3, 4 i/ii, 5 at G=1, the sdr_fm plan) as \"configs\"")
    p.add_argument("--configs-max-channels", type=int, default=0, help="tests: cap the channel count of every entry of \"configs\"")
    p.add_argument("--config-sustain-seconds", type=float, default=0.4, help="pre-conditioning per entry of \"configs\"")
    p.add_argument("--config-cpu-seconds", type=float, default=2.5, help="CPU-baseline sample per entry of \"configs\"")
    p.add_argument("--comm", default="torch", choices=["torch", "sdrhip"],
                   help="sdrhip: ONE process drives --gpus N rank contexts through the C ABI's sdrhip_comm_* (RCCL opened by the library; "
                        "ranks that share a device use its same-device transport)")
    p.add_argument("--no-verify", action="store_true", help="skip the oracle check of the last timed step's output")
    p.add_argument("--verify-channels", type=int, default=32)
    p.add_argument("--fft-whole-blocks", action="store_true",
                   help="fftconv: round --samples up to whole overlap-save hops (12288) so that no ragged last block is transformed")
    p.add_argument("--dump-output", default="", help="rank 0 saves the last step's (gathered) output rows as .npy (tests)")
    p.add_argument("--force-dist", action="store_true",
                   help="tests: initialise the process group even with one rank, so that the RCCL gather path (side stream, async_op) runs on a one-GPU box")
    return p.parse_args()


def count_gpus_sysfs():
    """HIP devices of this node WITHOUT touching HIP or torch: KFD topology nodes that have SIMDs (CPUs have none).
    None if the topology is not readable (the ranks then report a shortage themselves)."""
    import glob
    n, seen = 0, False
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(f):
                if line.startswith("simd_count"):
                    seen = True
                    n += int(line.split()[1]) > 0
        except (OSError, ValueError):
            pass
    if seen:
        return n
    try:   # no readable topology: ask a short-lived child (this parent still never loads torch or HIP)
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        sys.stderr.write("bench.py: could not count devices (%s)\n" % e)
        return None


def spawn_ranks(a):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD `torch.distributed.run`
    (one process per GPU, rendezvous on 127.0.0.1) and hand back its exit code. Neither torch nor HIP is touched in this
    process, and it is a child process, never an exec. The child's stdout is passed through line by line; if it ends
    with a non-zero code and no JSON line was seen (a rank other than 0 died first, or the RCCL watchdog aborted the
    ranks), this parent prints the one error line the contract asks for."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True, bufsize=1)
    saw_json = False
    for line in child.stdout:
        saw_json = saw_json or line.lstrip().startswith("{")
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = child.wait()
    if rc != 0 and not saw_json:
        print(error_line(a, "the %d ranks exited with code %d before rank 0 printed its line (see stderr)" % (a.gpus, rc)), flush=True)
    return rc


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


def cpu_limits():
    """What the box lets this process use: CPUs in the affinity mask, and the cgroup CPU quota if there is one
    (cgroup v2 cpu.max "quota period", or v1 cfs_quota_us / cfs_period_us) as a number of cores."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota, src = None, None
    try:
        t = open("/sys/fs/cgroup/cpu.max").read().split()
        src = "cgroup v2 cpu.max = %s" % " ".join(t)
        if t and t[0] != "max":
            quota = float(t[0]) / float(t[1])
    except (OSError, ValueError, IndexError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            src = "cgroup v1 cfs_quota_us / cfs_period_us = %d / %d" % (q, pr)
            if q > 0:
                quota = q / float(pr)
        except (OSError, ValueError):
            pass
    return {"cpus_online": os.cpu_count(), "cpus_in_affinity_mask": aff, "cgroup_quota_cores": quota, "cgroup_source": src}


def _throttled():
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("nr_throttled"):
                return int(line.split()[1])
    except (OSError, ValueError):
        pass
    return None


CPU_CHAINS = {"iqbb_fm": "iqbb_fm", "iqbb_usb": "iqbb_usb", "fir255_fm": "fir255_fm", "fir127_fm": "fir127_fm",
              "fbb_f32": "fir_cf32_sub8", "fftconv": "fir_cf32_4097", "iqbb_fm_cu8/sdr_fm": "sdr_fm_cu8"}
CPU_CHAIN_NOTE = {"fir_cf32_sub8": "FIRLowPass<cf32>(127) -> SubSample(8): the reference has no float frequency-shift node (SURVEY fact 6)",
                  "fir_cf32_4097": "FIRLowPass<cf32>(4097 taps), the time-domain filter config 4 (ii) is set against (the reference's FFT filter needs FFTW3, absent here)",
                  "sdr_fm_cu8": "examples/sdr_fm.cc's plan: cu8 -> AutoCast -> IQBaseBand<int16>(21 taps, 1 MS/s -> 8 kS/s) -> FMDemod"}


def cpu_baseline(workload, target_s, all_cores=True, chain=None):
    """Reference CPU path on this box: the compiled, unmodified reference if oracle/_ref travelled here,
    else the oracle port. One thread = the reference's real execution model (one Queue worker)."""
    chain = chain or CPU_CHAINS.get(workload)
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    cores_avail = os.cpu_count()
    nsamp = 16384 if chain == "fir_cf32_4097" else 65536   # (4097 fp64 taps per sample: a 65536 buffer alone is seconds)
    if chain and os.path.exists(ref):
        try:
            probe = json.loads(subprocess.run([ref, "bench", chain, "1" if nsamp < 65536 else "8", str(nsamp)], capture_output=True, text=True, timeout=120).stdout)
            nbuf = max(1 if nsamp < 65536 else 8, int(target_s * probe["msps"] * 1e6 / nsamp))
            r = json.loads(subprocess.run([ref, "bench", chain, str(nbuf), str(nsamp)], capture_output=True, text=True, timeout=600).stdout)
            res = {"value": round(r["msps"], 4), "unit": "Msamples/s", "cores": 1, "kind": "reference",
                   "sample": "%d buffers x %d samples, 1 channel, chain %s (reference nodes compiled -O3, "
                             "%.1f s)" % (nbuf, nsamp, chain, r["seconds"]), "host_cores_available": cores_avail, "cpu_model": cpu_model()}
            if chain in CPU_CHAIN_NOTE:
                res["chain"] = CPU_CHAIN_NOTE[chain]
            if not all_cores:
                return res
            # one channel (= one reference graph) per host core the box lets us use, all at once, ~3 s
            try:
                lim = cpu_limits()
                ncore = lim["cpus_in_affinity_mask"]
                if lim["cgroup_quota_cores"]:
                    ncore = min(ncore, max(1, int(lim["cgroup_quota_cores"] + 0.5)))
                ncore = max(1, min(ncore, 256))
                nb = max(8, int(3.0 * probe["msps"] * 1e6 / nsamp))
                th0 = _throttled()
                t0 = time.perf_counter()
                procs = [subprocess.Popen([ref, "bench", chain, str(nb), str(nsamp)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                         for _ in range(ncore)]
                outs = [pr.communicate(timeout=600)[0] for pr in procs]
                wall = time.perf_counter() - t0
                rows = [json.loads(o) for o in outs if o.strip()]
                done = sum(x["samples"] for x in rows)
                th1 = _throttled()
                res["all_cores"] = {"value": round(done / wall / 1e6, 2), "unit": "Msamples/s", "cores": ncore,
                                    "per_process_msps": round(sum(x["msps"] for x in rows) / max(1, len(rows)), 3),
                                    "sample": "%d independent reference graphs (processes) x %d buffers, wall %.1f s" % (ncore, nb, wall),
                                    "limits": lim, "cgroup_throttled_periods": (th1 - th0) if th0 is not None and th1 is not None else None}
            except Exception as e:
                res["all_cores"] = {"error": str(e)[:80]}
            return res
        except Exception as e:   # fall through to the port
            sys.stderr.write("cpu_baseline: reference binary failed (%s), using the port\n" % e)
    from oracle import pyoracle as orc   # bench.py's cpu_baseline leg may use the oracle
    if workload == "fftconv_ola":
        # the reference's FilterSink/FilterSource need FFTW3 (not in /root/reference, not installed): the oracle's restatement
        # of the same overlap-add blocks (its own double-precision radix-2 transform) is what can be timed here
        import numpy as np
        f = orc.FFTFilter(orc.fftfilt_design_K(orc.fftfilt_design_h(8192, 50e3, 150e3, FS)))
        x = (np.random.default_rng(1).standard_normal((8192, 2)) * 0.3).astype(np.float32)
        f.process(x)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < target_s:
            f.process(x)
            n += 1
        sec = time.perf_counter() - t0
        return {"value": round(n * 8192 / sec / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
                "sample": "%d blocks x 8192 cf32 samples, 1 channel, the oracle's FilterSink -> FilterSource restatement (16384-point "
                          "double-precision radix-2 transforms; the reference's own needs FFTW3), %.1f s" % (n, sec),
                "host_cores_available": cores_avail, "cpu_model": cpu_model()}
    if workload not in ("iqbb_fm",):
        return None
    taps = orc.iqbb_design(100e3, 50e3, FS, 127)
    lut = orc.freqshift_lut_i16()
    x = orc.IQSigGen(FS, [(100e3, 8000, 0.0), (-300e3, 6000, 0.3)]).next_cs16(65536)
    sec = orc.bench_iqbb_fm(taps, lut, 1365, False, 8, x, 8)
    nbuf = max(8, int(target_s / (sec / 8)))
    sec = orc.bench_iqbb_fm(taps, lut, 1365, False, 8, x, nbuf)
    return {"value": round(nbuf * 65536 / sec / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d buffers x 65536 cs16 samples, 1 channel, IQBaseBand(127,/8)->FM oracle port (%.1f s)" % (nbuf, sec),
            "host_cores_available": cores_avail}


def measured_traffic(workload_key, kernels):
    """Per-step HBM bytes of the step's kernels (one step = one launch of each) from the committed rocprofv3 PMC
    passes (profiles/*_pmc.json, newest first; collected by tools/prof.sh in separate --pmc runs of this same command
    and corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 on gfx950, KiB units). A profile only counts for the
    workload it was cut on: its `_meta.workload_key` (written by tools/summarize_prof.py from this script's own
    `config.workload_key`) must equal ours — a kernel NAME is shared by many workloads. None if there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), key=os.path.getmtime, reverse=True)
    files.sort(key=lambda f: os.path.basename(f), reverse=True)
    for fn in files:
        try:
            js = json.load(open(fn))
            if js.get("_meta", {}).get("workload_key") != workload_key:
                continue
            ds = [js.get(k, {}).get("derived") for k in kernels]
            if all(ds):
                res = {"bytes": sum(d["hbm_traffic_bytes_per_launch"] for d in ds), "source": os.path.basename(fn)}
                k0 = js.get(kernels[0], {})
                cu = k0.get("SQ_BUSY_CU_CYCLES", {}).get("mean")
                if cu:   # (per launch: instruction-active cycles summed over the SIMDs / CU-busy cycles x 4 SIMDs)
                    if "SQ_ACTIVE_INST_VALU" in k0:
                        res["valu_busy"] = round(k0["SQ_ACTIVE_INST_VALU"]["mean"] / cu, 3)
                    if "SQ_VALU_MFMA_BUSY_CYCLES" in k0:
                        res["mfma_busy"] = round(k0["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (4.0 * cu), 3)
                return res
        except Exception:
            pass
    return None


def roofline_block(w, C, N, per_launch_s, sustained_ms=None):
    """The `roofline` object of one workload: algorithmic HBM bytes per launch / the launch's HIP-event time against 8 TB/s
    (`achieved`, `peak`, `frac`, as the contract asks), `bound` = what actually limits the kernel, `compute` = the limiting
    pipe's own rate against its peak (report both), `traffic` = PMC bytes of this workload's committed profile."""
    ach = C * N * w.alg_bytes / per_launch_s / 1e9
    rf = {"bound": w.bound, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
          "traffic": None, "kernel": w.kernels[0], "avg_launch_ms": round(per_launch_s * 1e3, 4)}
    if len(w.kernels) > 1:
        rf["kernels_per_step"] = w.kernels
    if sustained_ms:
        rf["sustained_ms_per_launch"] = round(sustained_ms, 4)
    sps = C * N / per_launch_s
    comp = None
    if getattr(w, "mfma", None):
        comp = w.mfma(sps)
    elif getattr(w, "dp_slots", None):
        t = w.dp_slots * sps / 1e12
        comp = {"unit": "T fp64 instr/s (mul+add+trunc per tap, no FMA)", "achieved": round(t, 2),
                "peak": DP_ISSUE_PEAK_TOPS, "frac": round(t / DP_ISSUE_PEAK_TOPS, 4)}
    tr = measured_traffic(w.key, w.kernels)
    if tr:
        rf["traffic"], rf["traffic_source"] = round(tr["bytes"]), "profiles/" + tr["source"]
        if "valu_busy" in tr or "mfma_busy" in tr:
            comp = comp or {}
            comp.update({k: tr[k] for k in ("valu_busy", "mfma_busy") if k in tr})
            comp["busy_source"] = "profiles/" + tr["source"]
    if comp:
        rf["compute"] = comp
    return rf


class Telemetry:
    """Shader clock and socket power of one device while a phase runs, read from amdgpu's sysfs files by a side THREAD of
    this process (plain file reads: no HIP call, no child process, nothing re-executed): hwmon freq1_input (sclk, Hz) or
    the starred line of pp_dpm_sclk, and hwmon power1_average / power1_input (microwatts). Whatever the box does not
    expose stays None; `source` says what was read."""

    def __init__(self, pci_bus_id, period_s=0.02):
        import glob
        self.period, self.sclk, self.power, self.files = period_s, [], [], {}
        devs = []
        for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                real = os.path.realpath(d)
            except OSError:
                continue
            if pci_bus_id and os.path.basename(real).lower() != pci_bus_id.lower():
                continue
            devs.append(d)
        if not devs and not pci_bus_id:
            devs = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))[:1]
        for d in devs[:1]:
            for h in sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*"))):
                for key, names in (("sclk_hz", ["freq1_input"]), ("power_uw", ["power1_average", "power1_input"]), ("power_cap_uw", ["power1_cap"])):
                    for nm in names:
                        f = os.path.join(h, nm)
                        if key not in self.files and os.access(f, os.R_OK):
                            self.files[key] = f
            f = os.path.join(d, "pp_dpm_sclk")
            if "sclk_hz" not in self.files and os.access(f, os.R_OK):
                self.files["sclk_dpm"] = f
        self._stop, self._thr = None, None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return f.read()
        except OSError:
            return None

    def _sample(self):
        if "sclk_hz" in self.files:
            t = self._read(self.files["sclk_hz"])
            if t and t.strip().isdigit():
                self.sclk.append(int(t) / 1e6)
        elif "sclk_dpm" in self.files:
            for line in (self._read(self.files["sclk_dpm"]) or "").splitlines():
                if line.rstrip().endswith("*"):
                    try:
                        self.sclk.append(float(line.split(":")[1].strip().lower().split("mhz")[0]))
                    except (IndexError, ValueError):
                        pass
        if "power_uw" in self.files:
            t = self._read(self.files["power_uw"])
            if t and t.strip().isdigit():
                self.power.append(int(t) / 1e6)

    def start(self):
        import threading
        if not self.files:
            return
        self.sclk, self.power = [], []
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                self._sample()
                self._stop.wait(self.period)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()

    def stop(self):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=2.0)
            self._thr = None
        avg = lambda v: round(sum(v) / len(v), 1) if v else None
        cap = self._read(self.files["power_cap_uw"]) if "power_cap_uw" in self.files else None
        return {"sclk_mhz": avg(self.sclk), "sclk_mhz_min": round(min(self.sclk), 1) if self.sclk else None,
                "power_w": avg(self.power), "power_cap_w": round(int(cap) / 1e6, 1) if cap and cap.strip().isdigit() else None,
                "samples": max(len(self.sclk), len(self.power)),
                "source": {k: v for k, v in self.files.items()} or None}


def error_line(a, text, world=None):
    """The one JSON line of a run that could not measure: same keys, value null, an `error` text (the caller exits non-zero)."""
    return json.dumps({"metric": "Msamples/s through baseband->FIR->demod chain", "value": None, "unit": "Msamples/s",
                       "n_gpus": world if world is not None else a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": None,
                       "higher_is_better": True, "scaling": "strong" if a.global_channels else "weak", "vs_baseline": None,
                       "data": "synthetic", "error": str(text)[:600]})


class BenchError(RuntimeError):
    pass


class Workload:
    """One --workload: the plan, resident inputs, double-buffered outputs, the launch, and the oracle check."""
    pass


def multi_verify(verify1, B, N, D, real):
    """The oracle check of a multi-buffer step (--buffers B): the LAST buffer of the last step against the oracle, which is
    primed with the buffer before it — `verify1` is the one-buffer check, handed those two buffers, the outputs from the last
    buffer's first one on, and the absolute index of the last buffer's first sample."""
    def verify(prev, last, out, orc, n0=0, pre=None, **kw):
        M = (B - 1) * N
        q0 = ((n0 + M) // D - n0 // D) if real else ((n0 + M - 1) // D - (n0 - 1) // D if n0 else (M - 1) // D)
        pre1 = last[(B - 3) * N:(B - 2) * N] if B >= 3 else prev[(B - 1) * N:]
        return verify1(last[(B - 2) * N:(B - 1) * N], last[(B - 1) * N:], out[q0:], orc, n0=n0 + M, pre=pre1, **kw)
    return verify


def mfma_compute(info, samples_per_s, D, ovl, cu8):
    """int8 matrix work of a K1 plan, priced from the plan itself (sdrhip_iqbb_i16_plan_info): a wave slice of 512 samples
    takes (S + NH) K steps x 2 sample planes (complex<uint8>: 1) MFMAs of 32x32x32 = 65536 int8 multiply-adds... x 2 ops; a
    slice emits 64 - ovl groups at decimation 8 and 512 // D groups otherwise."""
    if info["path"] not in (1, 3, 4) or info["NH"] == 0:
        return None
    per_slice = (info["S"] + info["NH"]) * (1 if cu8 else 2)
    fresh = (64 - ovl) * 8 if D == 8 else (512 // D) * D if D <= 512 else 512
    ops = per_slice * 2.0 * 32 * 32 * 32 * samples_per_s / fresh
    return {"unit": "TOPS int8 MFMA", "achieved": round(ops / 1e12, 1), "peak": INT8_MFMA_PEAK_TOPS, "frac": round(ops / 1e12 / INT8_MFMA_PEAK_TOPS, 4),
            "mfma_per_slice": per_slice}


def build_workload(a, wl, sa, torch, shard, ctx, dev, rank, nbuf_out):
    import numpy as np
    C, N = a.channels, a.samples
    w = Workload()
    w.name, w.N, w.verify = wl, N, None
    order, D = a.order, a.decim
    B = max(1, a.buffers)
    NB = N * B   # samples per channel per step (--buffers reference-sized buffers in one launch)
    cs16 = lambda n=N: [synth_cs16(torch, C, n, dev, 1234 + b, chan0=rank * C) for b in range(a.batches)]
    w.bound, w.compute = "hbm", None
    if B > 1 and wl not in ("iqbb_fm", "iqbb_usb", "iqbb_fm_cu8", "bb_real_fm"):
        raise BenchError("--buffers is for the iqbb_* and bb_real_fm workloads")
    if wl in ("iqbb_fm", "iqbb_usb", "iqbb_fm_cu8"):
        taps = torch.from_numpy(sa.design_iqbb_taps(a.fc, a.width, a.fs, order)).to(dev)
        lut = torch.from_numpy(sa.design_freqshift_lut_i16()).to(dev)
        shard.broadcast_design([taps, lut], src=0)
        taps_h, lut_h, inc = taps.cpu().numpy(), lut.cpu().numpy(), sa.design_freqshift_inc(a.fc, a.fs)
        epi = sa.EPI_USB if wl == "iqbb_usb" else sa.EPI_FM
        node = sa.IQBaseBandI16(ctx, taps_h, lut_h, inc, False, D, channels=C, max_in=NB, epilogue=epi)
        w.in_bytes, w.alg_bytes = 4.0, 4.0 + 2.0 / D
        n_out = node.out_count(NB) + 1
        w.outs = [torch.zeros((C, n_out), dtype=torch.int16, device=dev) for _ in range(nbuf_out)]
        w.ins = cs16(NB)
        cu8 = wl == "iqbb_fm_cu8"
        if cu8:   # RTL-SDR bytes: the same signal as offset-binary complex<uint8>, AutoCast fused into the load
            node.set_input_format(sa.abi.IN_CU8)
            w.ins = [((x.to(torch.int32) >> 6) + 127).clamp_(0, 255).to(torch.uint8) for x in w.ins]
            w.in_bytes, w.alg_bytes = 2.0, 2.0 + 2.0 / D
        w.run = lambda b, o: node.process_dev(w.ins[b].data_ptr(), N, N, w.outs[o].data_ptr(), n_out)
        w.dtype, w.kernels = "i16", node.kernel_names
        if B > 1:
            w.run = lambda b, o: node.process_dev_multi(w.ins[b].data_ptr(), B, N, NB, w.outs[o].data_ptr(), n_out)
            if epi == sa.EPI_FM:   # (one untimed call tells whether the hot kernel writes every buffer boundary itself or a fix-up launch follows)
                for _ in range(2):
                    w.run(0, 0)
                node.reset()
                if node.plan_info["multi_left"] != 0:
                    w.kernels = w.kernels + ["iqbb_fm_multi_fixup_kernel"]
        w.desc = "IQBaseBand<int16>(%d-tap Q14 FIR, %s, /%d) -> %s" % (order, "LUT shift %g kHz" % (a.fc / 1e3) if inc else "no shift", D,
                                                                      "USBDemod" if wl == "iqbb_usb" else "FMDemod")
        if cu8:
            w.desc = "complex<uint8> -> AutoCast + " + w.desc
        if B > 1:
            w.desc += ", %d buffers of %d samples per channel in ONE launch (buffer boundaries kept)" % (B, N)
        w.key = "%s/order%d/d%d" % (wl, order, D) + ("" if a.fc == 100e3 else "/fc%g" % a.fc) + ("" if a.fs == FS and a.width == 50e3 else "/fs%g/w%g" % (a.fs, a.width))
        w.key += "/B%d" % B if B > 1 else ""
        w.plan = node.plan_info
        # what limits the kernel (profiles/README.md): the 9- and 17-step plans at decimation 8 run at the socket's power limit with
        # the matrix pipe 60 % busy; every other K1 plan is bound by vector-instruction issue
        hot8 = D == 8 and w.plan["path"] == 1
        w.bound = "power/mfma" if hot8 and w.plan["S"] >= 9 and not cu8 else "valu-issue"
        w.mfma = lambda sps: mfma_compute(w.plan, sps, D, 1 if epi == sa.EPI_FM and D == 8 else 0, cu8)
        w.n_valid = node.out_count(N)   # (every call after the first emits N / D outputs)
        de_alpha = 0
        if a.deemph and epi == sa.EPI_FM:
            # the rest of the reference's FM receiver chain: the demodulated rows go through FMDeemph (a second launch; the
            # recurrence is sequential per channel). The call's output count is known on the host (out_count).
            de_alpha = sa.design_fmdeemph_alpha(FS / D)
            if B > 1:
                raise BenchError("--deemph with --buffers: not a bench line")
            de = sa.FMDeemphI16(ctx, de_alpha, channels=C, max_in=n_out)
            w.mids = w.outs
            w.outs = [torch.zeros((C, n_out), dtype=torch.int16, device=dev) for _ in range(nbuf_out)]

            def run_chain(b, o):
                k = node.process_dev(w.ins[b].data_ptr(), N, N, w.mids[o].data_ptr(), n_out)
                de.process_dev(w.mids[o].data_ptr(), k, n_out, w.outs[o].data_ptr(), n_out)
            w.run = run_chain
            w.kernels = w.kernels + de.kernel_names(node.out_count(N))
            w.desc += " -> FMDeemph(alpha %d)" % de_alpha
            w.key += "/deemph"
            w.verify_needs_chan = True

        w.verify_needs_n0 = w.verify_needs_pre = True

        def check(out, r, orc, n_expect, chan, last_i):
            if len(r) != n_expect:
                return False
            if not de_alpha:
                return bool(np.array_equal(out[:len(r)], r))
            # with FMDeemph: the demodulator's rows are the intermediate buffer; the filter's state before the last call is
            # the last value it wrote in the call before (its output IS its running average, src/demod.hh FMDeemph)
            if not np.array_equal(w.mids[last_i & 1][chan].cpu().numpy()[:len(r)], r):
                return False
            n_prev = (n0_of[0] - 1) // D - (n0_of[0] - N - 1) // D
            d = orc.FMDeemphI16.__new__(orc.FMDeemphI16)
            d.alpha, d.avg = de_alpha, w.outs[(last_i - 1) & 1][chan].cpu().numpy()[n_prev - 1:n_prev].copy()
            return bool(np.array_equal(out[:len(r)], d.process(r)))
        n0_of = [0]

        def verify(prev, last, out, orc, n0=0, pre=None, chan=0, last_i=0):
            n0_of[0] = n0
            # the state a call starts from (FIR history, open window, FM angle) depends on the previous buffer only;
            # LUT and decimator phases repeat every 32768 / D samples, so they are those of a stream's second call
            bb, fm = orc.IQBaseBandI16(taps_h, lut_h, inc, False, D), orc.FMDemodI16()
            cast = (lambda x: orc.autocast_cu8_cs16(x)) if cu8 else (lambda x: x)
            if N % 32768 or N % D:
                # any other decimation: the oracle is put at the last emission in front of the previous buffer (absolute
                # index s0 = g*D + 1 <= n0 - N: decimator and LUT phase by seek(), the FIR ring primed with the `order`
                # samples before s0 — both from the tail of the buffer before the previous one) and runs from there
                n_prev = n0 - N
                s0 = ((n_prev - 1) // D) * D + 1
                if pre is None or D < 2 or s0 < D + 1 or n_prev - s0 + order > N or order > N:
                    return None
                tail = cast(pre[N - (n_prev - s0) - order:])
                bb.process(tail[:order])
                bb.seek(s0)
                r0 = np.concatenate([bb.process(tail[order:]), bb.process(cast(prev))])
                if epi == sa.EPI_FM:
                    fm.process(r0)
                r = bb.process(cast(last))
                r = fm.process(r) if epi == sa.EPI_FM else orc.usb_i16(r)
                return check(out, r, orc, (n0 + N - 1) // D - (n0 - 1) // D, chan, last_i)
            if order > N:
                return None
            r0 = bb.process(cast(prev))
            if epi == sa.EPI_FM:
                fm.process(r0)
            r = bb.process(cast(last))
            r = fm.process(r) if epi == sa.EPI_FM else orc.usb_i16(r)
            return check(out, r, orc, N // D, chan, last_i)
        w.verify = multi_verify(verify, B, N, D, False) if B > 1 else verify
    elif wl == "iqbb_fm_cs8":   # the int8 chain IQBaseBand<int8_t> -> FMDemod<int8_t,int16_t> (src/sdr.hh:225-240), 2 bytes per sample in
        if B > 1:
            raise BenchError("--buffers: not for the int8 chain")
        taps_h = sa.design_iqbb_taps(a.fc if a.fc else 100e3, a.width, a.fs, order)
        lut_h, inc = sa.design_freqshift_lut_i8(), sa.design_freqshift_inc(a.fc, a.fs)
        node = sa.IQBaseBandI8(ctx, taps_h, lut_h, inc, False, D, channels=C, max_in=N, epilogue=sa.EPI_FM)
        w.in_bytes, w.alg_bytes = 2.0, 2.0 + 2.0 / D
        n_out = node.out_count(N) + 1
        w.outs = [torch.zeros((C, n_out), dtype=torch.int16, device=dev) for _ in range(nbuf_out)]
        w.ins = [(x.to(torch.int32) >> 7).clamp_(-128, 127).to(torch.int8) for x in cs16()]   # (the same signal at 8 bits)
        w.run = lambda b, o: node.process_dev(w.ins[b].data_ptr(), N, N, w.outs[o].data_ptr(), n_out)
        w.dtype, w.kernels = "i8", node.kernel_names
        w.desc = "IQBaseBand<int8>(%d-tap Q14 FIR, %s, /%d) -> FMDemod<int8,int16>" % (order, "LUT shift %g kHz" % (a.fc / 1e3) if inc else "no shift", D)
        w.key = "%s/order%d/d%d" % (wl, order, D) + ("" if a.fc == 100e3 else "/fc%g" % a.fc)
        w.plan = node.plan_info
        w.bound = "valu-issue"
        w.mfma = lambda sps: mfma_compute(w.plan, sps, D, 1 if D == 8 else 0, True)

        def verify(prev, last, out, orc, n0=0, pre=None):
            if N % 32768 or N % D or order > N:
                return None
            bb, fm = orc.IQBaseBandI8(taps_h, lut_h, inc, False, D), orc.FMDemodI8()
            fm.process(bb.process(prev))
            r = fm.process(bb.process(last))
            return bool(np.array_equal(out[:len(r)], r)) and len(r) == N // D
        w.verify = verify
    elif wl == "bb_real_fm":   # the real-input BaseBand<int16_t> (2 bytes per sample in)
        taps_h = sa.design_bb_taps(100e3, 50e3, FS, order)
        lut_h, inc = sa.design_freqshift_lut_i16(), sa.design_freqshift_inc(100e3, FS)
        node = sa.BaseBandI16(ctx, taps_h, lut_h, inc, False, D, channels=C, max_in=NB, epilogue=sa.EPI_FM)
        w.in_bytes, w.alg_bytes = 2.0, 2.0 + 2.0 / D
        n_out = node.out_count(NB) + 1
        w.outs = [torch.zeros((C, n_out), dtype=torch.int16, device=dev) for _ in range(nbuf_out)]
        w.ins = [x[..., 0].contiguous() for x in cs16(NB)]
        w.run = lambda b, o: node.process_dev(w.ins[b].data_ptr(), N, N, w.outs[o].data_ptr(), n_out)
        w.dtype, w.kernels = "i16", node.kernel_names
        if B > 1:
            w.run = lambda b, o: node.process_dev_multi(w.ins[b].data_ptr(), B, N, NB, w.outs[o].data_ptr(), n_out)
            for _ in range(2):   # (as the complex workloads: does a fix-up launch follow the hot kernel?)
                w.run(0, 0)
            node.reset()
            if node.plan_info["multi_left"] != 0:
                w.kernels = w.kernels + ["iqbb_fm_multi_fixup_kernel"]
        w.desc = "BaseBand<int16> real input (%d-tap Q16 FIR, LUT shift 100 kHz, /%d) -> FMDemod" % (order, D)
        w.key = "%s/order%d/d%d" % (wl, order, D) + ("/B%d" % B if B > 1 else "")
        w.plan = node.plan_info
        w.bound = "valu-issue"
        w.mfma = lambda sps: mfma_compute(w.plan, sps, D, 1 if D == 8 else 0, False)

        def verify(prev, last, out, orc, n0=0, pre=None):
            if order > N:
                return None
            bb, fm = orc.BaseBandI16(taps_h, lut_h, inc, False, D), orc.FMDemodI16()
            if N % 32768 or N % D:
                # any other decimation: the oracle is put at the last group boundary in front of the previous buffer (absolute
                # index s0 = g*D <= n0 - N: decimator and LUT phase by seek(), the ring primed with the `order` samples before it)
                n_prev = n0 - N
                s0 = (n_prev // D) * D
                if pre is None or s0 < D or n_prev - s0 + order > N:
                    return None
                tail = pre[N - (n_prev - s0) - order:]
                bb.process(tail[:order])
                bb.seek(s0)
                fm.process(np.concatenate([bb.process(tail[order:]), bb.process(prev)]))
                r = fm.process(bb.process(last))
                return bool(np.array_equal(out[:len(r)], r)) and len(r) == (n0 + N) // D - n0 // D
            fm.process(bb.process(prev))
            r = fm.process(bb.process(last))
            return bool(np.array_equal(out[:len(r)], r)) and len(r) == N // D
        w.verify, w.verify_needs_n0, w.verify_needs_pre = (multi_verify(verify, B, N, D, True) if B > 1 else verify), True, True
    elif wl in ("fir255_fm", "fir127_fm"):
        order = 255 if wl == "fir255_fm" else 127
        alpha = torch.from_numpy(sa.design_fir_lowpass(order, 100e3, FS)).to(dev)
        shard.broadcast_design([alpha], src=0)
        alpha_h = alpha.cpu().numpy()
        node = sa.FIR(ctx, sa.FIR_CS16_EXACT, alpha_h, channels=C, max_in=N, epilogue=sa.EPI_FM)
        w.in_bytes, w.alg_bytes = 4.0, 6.0
        w.outs = [torch.zeros((C, N), dtype=torch.int16, device=dev) for _ in range(nbuf_out)]
        w.ins = cs16()
        w.run = lambda b, o: node.process_dev(w.ins[b].data_ptr(), N, N, w.outs[o].data_ptr(), N)
        w.dtype, w.kernels = "f64", ["fir_cs16_exact_kernel"]
        w.bound = "fp64-issue"   # 6 dependent DP instructions per tap and sample (mul, add, trunc per component)
        w.dp_slots = 6.0 * order
        w.desc = "FIRLowPass<complex<int16>>(%d taps, exact per-tap truncation) -> FMDemod" % order
        w.key = wl

        def verify(prev, last, out, orc):
            fir, fm = orc.FIR(alpha_h), orc.FMDemodI16()
            fm.process(fir.process_cs16(prev))
            return bool(np.array_equal(out, fm.process(fir.process_cs16(last))))
        w.verify = verify
    elif wl == "fbb_f32":
        alpha_h = sa.design_fir_lowpass(127, 100e3, FS)
        node = sa.FloatBaseBand(ctx, 100e3, FS, alpha_h, 8, channels=C, max_in=N)
        w.in_bytes, w.alg_bytes = 8.0, 9.0
        n_out = N // 8 + 1
        w.outs = [torch.zeros((C, n_out, 2), dtype=torch.float32, device=dev) for _ in range(nbuf_out)]
        w.ins = [torch.randn((C, N, 2), dtype=torch.float32, device=dev) * 0.3 for b in range(a.batches)]
        w.run = lambda b, o: node.process_dev(w.ins[b].data_ptr(), N, N, w.outs[o].data_ptr(), n_out)
        w.dtype, w.kernels = "f32", node.kernel_names(N)
        w.desc = "float baseband: shift 100 kHz -> FIRLowPass<cf32>(127) -> /8"
        w.key = wl

        def verify(prev, last, out, orc, n0=0):
            if N % 8:
                return None
            fir, sub = orc.FIR(alpha_h), orc.SubSample(8)
            sub.process_cf32(fir.process_cf32(orc.freqshift_cf32(prev, n0 - N, 100e3, FS)))
            r = sub.process_cf32(fir.process_cf32(orc.freqshift_cf32(last, n0, 100e3, FS)))
            o = out[:len(r)].astype(np.float64)
            return bool(np.abs(o - r).max() <= RTOL * max(np.abs(r).max(), 1e-30))
        w.verify, w.verify_needs_n0 = verify, True
    elif wl == "fftbank":   # FilterNode<float>: 4 bands behind ONE forward transform per block (2048-point, overlap-add)
        bands = [(50e3, 150e3), (-350e3, -250e3), (200e3, 300e3), (-120e3, -20e3)]
        Ks = [sa.design_fftfilt_spectrum(sa.design_fftfilt_kernel(1024, lo_, hi_, FS)) for lo_, hi_ in bands]
        node = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2048, Ks, channels=C, max_in=N)
        w.in_bytes, w.alg_bytes = 8.0, 8.0 + 8.0 * len(bands)
        w.outs = [torch.zeros((len(bands), C, N, 2), dtype=torch.float32, device=dev) for _ in range(nbuf_out)]
        w.ins = [torch.randn((C, N, 2), dtype=torch.float32, device=dev) * 0.3 for b in range(a.batches)]
        w.run = lambda b, o: node.process_dev(w.ins[b].data_ptr(), N, N, w.outs[o].data_ptr(), N)
        w.dtype, w.kernels, w.bound = "f32", ["fftconv_fused_kernel"], "lds/issue"
        w.desc = "FFT filter bank: 2048-point overlap-add, 1024-sample blocks, %d bands behind one forward transform" % len(bands)
        w.key, w.out_rows_axis = wl, 1

        def verify(prev, last, out, orc):   # out: [bands, N, 2]
            if N % 1024:
                return None
            ok = True
            for bi, K in enumerate(Ks):
                f = orc.FFTFilter(K)
                for blk in range(max(0, N // 1024 - 2), N // 1024):   # (the overlap-add tail reaches one block back)
                    f.process(prev[blk * 1024:(blk + 1) * 1024])
                r = np.concatenate([f.process(last[blk * 1024:(blk + 1) * 1024]) for blk in range(N // 1024)])
                ok = ok and bool(np.abs(out[bi].astype(np.float64) - r).max() <= RTOL * max(np.abs(r).max(), 1e-30))
            return ok
        w.verify = verify
    elif wl == "fftconv_ola":   # BASELINE config 4 (i): the reference's own mode — FilterSink -> FilterSource, block 8192, 16384-point
        Nb = 8192
        hk = sa.design_fftfilt_kernel(Nb, 50e3, 150e3, FS)
        Kk = sa.design_fftfilt_spectrum(hk)
        N = (N + Nb - 1) // Nb * Nb   # (the reference only accepts whole blocks: FilterSink::config throws otherwise)
        w.N = N
        node = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * Nb, Kk, channels=C, max_in=N)
        w.in_bytes, w.alg_bytes = 8.0, 16.0
        w.outs = [torch.zeros((C, N, 2), dtype=torch.float32, device=dev) for _ in range(nbuf_out)]
        w.ins = [torch.randn((C, N, 2), dtype=torch.float32, device=dev) * 0.3 for b in range(a.batches)]
        w.run = lambda b, o: node.process_dev(w.ins[b].data_ptr(), N, N, w.outs[o].data_ptr(), N)
        w.dtype, w.kernels, w.bound = "f32", ["fftconv_fused_kernel"], "lds/issue"
        w.desc = "FFT filter, reference mode: overlap-add, blocks of 8192, 16384-point transforms, 8192-tap FilterSource kernel 50..150 kHz"
        w.key = wl

        def verify(prev, last, out, orc):
            f = orc.FFTFilter(orc.fftfilt_design_K(hk))
            f.process(prev[-Nb:])   # (the overlap-add tail reaches one block back)
            r = np.concatenate([f.process(last[blk * Nb:(blk + 1) * Nb]) for blk in range(N // Nb)])
            return bool(np.abs(out.astype(np.float64) - r).max() <= RTOL * max(np.abs(r).max(), 1e-30))
        w.verify = verify
    elif wl == "fftconv":
        alpha_h = sa.design_fir_lowpass(4097, 100e3, FS)
        tapsf = np.stack([alpha_h[::-1], np.zeros_like(alpha_h)], 1).astype(np.float32)   # h[k] = alpha[order-1-k]
        if a.fft_whole_blocks:
            N = (N + 12287) // 12288 * 12288   # whole hops (16384 - 4096): a ragged last block is a full transform for part of a hop
            w.N = N
        node = sa.FFTConv(ctx, sa.FFTCONV_OLS, 16384, tapsf, channels=C, max_in=N)
        w.in_bytes, w.alg_bytes = 8.0, 16.0
        w.outs = [torch.zeros((C, N, 2), dtype=torch.float32, device=dev) for _ in range(nbuf_out)]
        w.ins = [torch.randn((C, N, 2), dtype=torch.float32, device=dev) * 0.3 for b in range(a.batches)]
        w.run = lambda b, o: node.process_dev(w.ins[b].data_ptr(), N, N, w.outs[o].data_ptr(), N)
        w.dtype, w.kernels, w.bound = "f32", ["fftconv_fused_kernel"], "lds/issue"
        w.desc = "FFT convolution, overlap-save L=16384, 4097 taps (hop 12288), %d samples per channel per step" % N
        w.key = "%s/N%d" % (wl, N)

        def verify(prev, last, out, orc):   # y = h (*) x, h = the reversed low-pass taps (causal convolution over the stream)
            from scipy.signal import fftconvolve
            x = np.concatenate([prev[-4096:], last]).astype(np.float64)
            xc = x[:, 0] + 1j * x[:, 1]
            r = fftconvolve(xc, alpha_h[::-1].astype(np.float64))[4096:4096 + N]
            o = out[:, 0].astype(np.float64) + 1j * out[:, 1].astype(np.float64)
            return bool(np.abs(o - r).max() <= RTOL * max(np.abs(r).max(), 1e-30))
        w.verify = verify
    elif wl in ("fm_demod", "subsample8"):
        w.ins = cs16()
        if wl == "fm_demod":
            node = sa.Demod(ctx, sa.EPI_FM, sa.T_CS16, channels=C, max_in=N)
            w.in_bytes, w.alg_bytes = 4.0, 6.0
            w.outs = [torch.zeros((C, N), dtype=torch.int16, device=dev) for _ in range(nbuf_out)]
            w.run = lambda b, o: node.process_dev(w.ins[b].data_ptr(), N, N, w.outs[o].data_ptr(), N)
            w.kernels, w.desc = ["demod_cs16_kernel"], "FMDemod<int16> alone (complex<int16> -> int16)"

            def verify(prev, last, out, orc):   # (out of place: FMDemod never writes index 0, src/demod.hh:245-250)
                fm = orc.FMDemodI16()
                fm.process(prev)
                return bool(np.array_equal(out[1:], fm.process(last)[1:]))
        else:
            node = sa.SubSample(ctx, sa.T_CS16, 8, channels=C, max_in=N)
            w.in_bytes, w.alg_bytes = 4.0, 4.5
            w.outs = [torch.zeros((C, N // 8 + 1, 2), dtype=torch.int16, device=dev) for _ in range(nbuf_out)]
            w.run = lambda b, o: node.process_dev(w.ins[b].data_ptr(), N, N, w.outs[o].data_ptr(), N // 8 + 1)
            w.kernels, w.desc = ["subsample8_cs16_kernel"], "SubSample<complex<int16>>(8) alone"

            def verify(prev, last, out, orc):
                if N % 8:
                    return None
                r = orc.SubSample(8).process_cs16(last)
                return bool(np.array_equal(out[:len(r)], r))
        w.verify, w.dtype, w.key = verify, "i16", wl
    else:
        raise BenchError("unknown workload " + wl)
    w.node = node
    if B > 1:
        w.N = NB
    w.key += "/C%d/N%d" % (C, w.N)
    return w


def main():
    """Runs the bench; whatever goes wrong (fewer devices than ranks, RCCL that does not come up, a failed launch) still
    ends in ONE JSON line on rank 0 — with an `error` key and value null — and a non-zero exit code."""
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    if a.comm == "sdrhip":   # one process whatever --gpus says: the library owns the ranks
        try:
            run_sdrhip(a)
        except BenchError as e:
            print(error_line(a, e), flush=True)
            raise SystemExit(2)
        except SystemExit:
            raise
        except BaseException as e:
            import traceback
            traceback.print_exc()
            print(error_line(a, "%s: %s" % (type(e).__name__, e)), flush=True)
            raise SystemExit(3)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        seen = count_gpus_sysfs()   # (KFD topology in sysfs: neither torch nor HIP is loaded in this parent)
        if seen is not None and seen < a.gpus and a.force_device < 0:
            print(error_line(a, "--gpus %d but only %d HIP device(s) visible on this node" % (a.gpus, seen)), flush=True)
            raise SystemExit(2)
        raise SystemExit(spawn_ranks(a))
    try:
        run(a)
    except BenchError as e:
        if rank == 0:
            print(error_line(a, e, world=int(os.environ.get("WORLD_SIZE", "1"))), flush=True)
            if int(os.environ.get("WORLD_SIZE", "1")) > 1:
                time.sleep(2.0)   # (the launcher tears the other ranks down as soon as one exits: give them the time to say why THEY stop)
        else:
            sys.stderr.write("bench.py rank %d: %s\n" % (rank, e))
            sys.stderr.flush()
        raise SystemExit(2)
    except SystemExit:
        raise
    except BaseException as e:   # (a HIP / RCCL error surfaces as RuntimeError or SdrHipError)
        import traceback
        traceback.print_exc()
        if rank == 0:
            print(error_line(a, "%s: %s" % (type(e).__name__, e), world=int(os.environ.get("WORLD_SIZE", "1"))), flush=True)
        raise SystemExit(3)


def verify_last(a, w, calls, rank, np, torch):
    """The LAST step's output of a few channels against the CPU oracle (outside every timed region). `calls` = steps run so
    far on this plan (the step counter is global: step i read batch i % batches and wrote output buffer i & 1)."""
    if a.no_verify or w.verify is None or calls < 2:
        return None
    C, N = w.outs[0].shape[1 if getattr(w, "out_rows_axis", 0) == 1 else 0], w.N
    last_i = calls - 1
    try:
        from oracle import pyoracle as orc   # checker only: never on the measured path
        rng = np.random.default_rng(12345 + rank)
        chans = sorted(rng.choice(C, size=min(a.verify_channels, C), replace=False).tolist())
        bl, bp = last_i % a.batches, (last_i - 1) % a.batches
        xo = w.outs[last_i & 1]
        oks = []
        for c in chans:
            prev, last = w.ins[bp][c].cpu().numpy(), w.ins[bl][c].cpu().numpy()
            out = (xo[:, c] if getattr(w, "out_rows_axis", 0) == 1 else xo[c]).cpu().numpy()
            kw = {"n0": (calls - 1) * N} if getattr(w, "verify_needs_n0", False) else {}
            if getattr(w, "verify_needs_pre", False) and calls >= 3:   # (the buffer before the previous one)
                kw["pre"] = w.ins[(last_i - 2) % a.batches][c].cpu().numpy()
            if getattr(w, "verify_needs_chan", False):
                kw.update(chan=c, last_i=last_i)
            oks.append(w.verify(prev, last, out, orc, **kw))
        if all(o is None for o in oks):
            return {"ok": None, "why": "this sample count / decimation is outside what the last-step check covers"}
        return {"ok": bool(all(oks)), "channels": len(chans), "mode": "last timed step vs CPU oracle",
                "tolerance": "bit-exact" if w.dtype in ("i16", "i8", "f64") else "max|y-ref|/max|ref| <= 1e-5"}
    except Exception as e:
        return {"ok": None, "why": "oracle unavailable: %s" % str(e)[:120]}


# The other BASELINE.json configs, measured in the same default one-GPU run behind the headline ("configs": [...]).
# `args` override the command line's; steps / warmup are the run's own. SURVEY §8d gives each config's shape. The entries
# are COMPACT (the driver keeps the last 8 KB of stdout): what an entry is, its workload text and the samples of its CPU
# baseline go to stderr as one "bench.py details" line.
CONFIG_SPECS = [
    {"id": "config1", "baseline_config": 1, "workload": "fir127_fm", "cpu_chain": "fir127_fm_queue", "args": {"channels": 1024},
     "what": "BASELINE config 1: FIRLowPass<cs16>(127 taps, exact per-tap truncation) -> FMDemod; GPU: 1024 channels per launch; CPU: the "
             "reference's IQSigGen idle-driven on its Queue (src/queue.cc:83-125, examples/sdr_fm.cc plumbing), one thread"},
    {"id": "config2_c1", "baseline_config": 2, "workload": "fbb_f32", "args": {"channels": 1},
     "what": "single channel complex<float> baseband (shift -> 127-tap FIR -> /8), 65536 samples per buffer: launch-latency-bound"},
    {"id": "config2_c1024", "baseline_config": 2, "workload": "fbb_f32", "args": {"channels": 1024},
     "what": "the same float baseband on 1024 channels per launch"},
    {"id": "config3", "baseline_config": 3, "workload": "fir255_fm", "args": {"channels": 1024},
     "what": "1024 int16 IQ channels, FIRLowPass<cs16>(255 taps, exact per-tap truncation in fp64) -> FMDemod"},
    {"id": "config4_i_ola8192", "baseline_config": 4, "workload": "fftconv_ola", "args": {"channels": 1024},
     "what": "fftplan FFT filter in the reference's mode (overlap-add, 16384-point, 8192-tap kernel), 1024 channels"},
    {"id": "config4_ii_ols4097", "baseline_config": 4, "workload": "fftconv", "args": {"channels": 1024, "fft_whole_blocks": True},
     "what": "overlap-save 16384-point FFT convolution with the 4097 FIRLowPass taps (vs the time-domain FIRFilter), 1024 channels"},
    {"id": "config5_g1", "baseline_config": 5, "workload": "iqbb_usb", "args": {"channels": 8192, "batches": 2}, "h2d": True,
     "what": "8192 channels, IQBaseBand<int16>(127, /8) -> USBDemod, the whole job on ONE GPU (G = 1 point of the scaling curve); "
             "with_h2d_*: the same step from PINNED HOST buffers (H2D over PCIe, kernel, D2H) — never `value`"},
    {"id": "sdr_fm_plan", "baseline_config": None, "workload": "iqbb_fm_cu8", "cpu_chain": "sdr_fm_cu8",
     "args": {"channels": 1024, "order": 21, "decim": 125, "fs": 1e6, "width": 12.5e3},
     "what": "the reference's own FM receiver plan (examples/sdr_fm.cc:38-43): complex<uint8> -> AutoCast -> IQBaseBand<int16>(21 taps, /125) -> FMDemod"},
    {"id": "bb_real_d20", "baseline_config": None, "workload": "bb_real_fm", "cpu_chain": None, "args": {"channels": 1024, "decim": 20},
     "what": "BaseBand<int16> (real input, src/baseband.hh:305-529) 127 taps, /20 -> FMDemod: the any-D form on the matrix cores"},
    {"id": "multi_buffer", "baseline_config": None, "workload": "iqbb_fm", "args": {"channels": 1024, "buffers": 4, "batches": 2},
     "what": "the headline chain, FOUR 65536-sample buffers per channel in one launch (sdrhip_iqbb_i16_process_dev_multi: buffer "
             "boundaries kept, FMDemod restarts per buffer): the launch ramp, border slices and state hand-over amortised"},
]


def measure_h2d(aa, w, torch, stream, K):
    """SURVEY §8d config 5 "with and without PCIe H2D": the same step fed from PINNED host memory — H2D copy of the step's
    input, the kernel, D2H copy of its output, all on the kernel's stream, K steps back to back (no overlap between steps:
    what a host that hands over host buffers pays). Returns (ms per step, input bytes per step)."""
    x = w.ins[0]
    hin = torch.empty(x.shape, dtype=x.dtype).pin_memory()
    hin.copy_(x.cpu())
    hout = torch.empty(w.outs[0].shape, dtype=w.outs[0].dtype).pin_memory()
    with torch.cuda.stream(stream):
        for _ in range(2):
            x.copy_(hin, non_blocking=True); w.run(0, 0); hout.copy_(w.outs[0], non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            x.copy_(hin, non_blocking=True); w.run(0, 0); hout.copy_(w.outs[0], non_blocking=True)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3, x.numel() * x.element_size()


def measure_config(a, spec, sa, torch, shard, ctx, dev, np, stream=None):
    """One entry of "configs": its own plan and resident inputs; pre-condition, W warm-up steps, K timed steps (wall clock
    around a synchronised region + HIP events on the kernel's stream), then the oracle check of the last step. Returns
    (compact entry for the JSON line, details for stderr)."""
    aa = argparse.Namespace(**vars(a))
    for k, v in spec["args"].items():
        setattr(aa, k, v)
    if a.configs_max_channels:
        aa.channels = min(aa.channels, a.configs_max_channels)
    t_setup = time.perf_counter()
    w = build_workload(aa, spec["workload"], sa, torch, shard, ctx, dev, 0, 2)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    C, N, K, W = aa.channels, w.N, a.steps, a.warmup
    timer = sa.Timer(ctx)
    it = [0]

    def step():
        w.run(it[0] % aa.batches, it[0] & 1)
        it[0] += 1
    pre_ms, pre_n, t1 = 0.0, 0, time.perf_counter()
    chunk = max(K, 20)
    while time.perf_counter() - t1 < a.config_sustain_seconds:
        timer.start()
        for _ in range(chunk):
            step()
        timer.stop()
        pre_ms += timer.elapsed_ms()
        pre_n += chunk
    pre_s = time.perf_counter() - t1
    for _ in range(W):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    timer.start()
    for _ in range(K):
        step()
    timer.stop()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = timer.elapsed_ms()
    ver = verify_last(aa, w, it[0], 0, np, torch)
    per_launch_s = dev_ms / 1e3 / K
    rf = roofline_block(w, C, N, per_launch_s, pre_ms / pre_n if pre_n else None)
    for k in ("peak", "unit"):   # (the same for every entry: the headline's roofline carries them)
        rf.pop(k, None)
    if rf.get("traffic") is None:
        rf.pop("traffic", None)
    rf.pop("traffic_source", None)
    if "compute" in rf:
        for k in ("busy_source", "peak"):
            rf["compute"].pop(k, None)
    e = {"id": spec["id"], "cfg": spec["baseline_config"], "key": w.key, "ms_per_step": round(wall / K * 1e3, 4),
         "value": round(float(C) * N * K / wall / 1e6, 1), "dtype": w.dtype, "roofline": rf, "verified": ver["ok"] if ver else None}
    det = {"id": spec["id"], "what": spec["what"], "workload": w.desc, "channels": C, "samples_per_channel_per_step": N, "steps": K, "warmup": W,
           "preconditioned_s": round(pre_s, 2), "unit": "Msamples/s", "verify": ver, "setup_s": round(t_setup, 2),
           "algorithmic_bytes_per_sample": w.alg_bytes, "kernels_per_step": w.kernels}
    if spec["workload"] == "fbb_f32" and C == 1:   # config 2 as SURVEY §8d states it: what a buffer costs, and against real time
        rf["per_buffer_us"] = round(per_launch_s * 1e6, 2)
        rf["real_time_factor"] = round((N / FS) / per_launch_s, 1)
        if pre_n:   # (K = 20 launches of a few microseconds each mostly time the launch ramp: the back-to-back figure beside it)
            rf["per_buffer_us_sustained"] = round(pre_ms / pre_n * 1e3, 2)
    if spec.get("h2d") and stream is not None:
        try:
            ms, inb = measure_h2d(aa, w, torch, stream, max(3, min(K, 10)))
            e["with_h2d_ms_per_step"] = round(ms, 3)
            e["with_h2d_value"] = round(float(C) * N / ms / 1e3, 1)
            e["with_h2d_pcie_gbs"] = round(inb / ms / 1e6, 1)
        except Exception as ex:
            e["with_h2d_error"] = str(ex)[:80]
    del timer
    w.node.close()
    del w
    torch.cuda.empty_cache()
    return e, det


def run(a):
    import numpy as np
    import torch
    import torch.distributed as dist
    import libsdr_amd as sa
    from libsdr_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise BenchError("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if a.force_device >= 0:
        local = a.force_device
    if world != a.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; reporting the ranks that run\n" % (a.gpus, world))
    if local >= torch.cuda.device_count():
        raise BenchError("rank %d wants device %d but only %d visible (dry runs: --backend gloo --force-device 0)"
                         % (rank, local, torch.cuda.device_count()))
    if a.global_channels:   # strong scaling: a fixed job, contiguous blocks of G / N channels per GPU
        if a.global_channels % world:
            raise BenchError("--global-channels %d is not a multiple of the %d ranks" % (a.global_channels, world))
        a.channels = a.global_channels // world
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # (--force-dist: a one-rank group)
            import socket
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1])); s_.close()
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        tmo = datetime.timedelta(seconds=300)   # (a collective that never completes becomes an error line, not a hung run)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(a.backend, timeout=tmo)
    # one GPU: the north-star chain (FM, no collective); several: BASELINE config 5 (USB, output gathered on the root)
    wl = a.workload or ("iqbb_usb" if world > 1 else "iqbb_fm")
    gather = (a.gather or (world > 1 and not a.workload)) and not a.no_gather and use_dist
    C, W, K = a.channels, a.warmup, a.steps
    t_run0 = time.perf_counter()

    stream = torch.cuda.Stream(device=dev)
    side = torch.cuda.Stream(device=dev)   # the gather's stream
    with torch.cuda.stream(stream):
        ctx = sa.Context(local, stream=stream.cuda_stream)
        # ---- config(): design on rank 0, broadcast over RCCL (KBs; outside the timed region) ----
        w = build_workload(a, wl, sa, torch, shard, ctx, dev, rank, 2)
        N = w.N
        gathered, gl, g_ok = None, None, gather and a.backend == "nccl"
        send = None
        if gather and w.outs[0].dim() == 2:
            # (RCCL has no int16 type: the rows travel as bytes — uint8 views of the same memory)
            send = [o.view(torch.uint8) for o in w.outs]
            if rank == 0:   # the root's landing zone: every rank's rows, in global channel order — no concatenation later
                gathered = torch.zeros((world * C, w.outs[0].shape[1]), dtype=w.outs[0].dtype, device=dev)
                g8 = gathered.view(torch.uint8)
                gl = [g8[r * C:(r + 1) * C] for r in range(world)]
        elif gather:
            raise BenchError("--gather needs a workload with one output row per channel")
        pending = [None, None]   # the gather that still reads outs[o]
        ev = [torch.cuda.Event(), torch.cuda.Event()]
        it = [0]                 # steps run so far on this plan: step i reads batch i % batches and writes output buffer i & 1

        def issue_gather(o):
            if g_ok:   # RCCL: on the side stream, behind this step's kernel; the next step starts meanwhile
                ev[o].record(stream)
                with torch.cuda.stream(side):
                    side.wait_event(ev[o])
                    pending[o] = dist.gather(send[o], gl if rank == 0 else None, dst=0, async_op=True)
            else:      # gloo dry runs / tests: host-staged, blocking
                shard.gather_output(w.outs[o], C * world, dst=0, out=gathered)

        def step(with_gather, compute=True):
            i = it[0]
            o = i & 1
            if pending[o] is not None:   # the compute stream waits (on the device) until the gather of step i-2 has read outs[o]
                pending[o].wait()
                pending[o] = None
            if compute:
                w.run(i % a.batches, o)
                it[0] = i + 1
            if with_gather:
                issue_gather(o)

        def drain():
            for o in (0, 1):
                if pending[o] is not None:
                    pending[o].wait()
                    pending[o] = None

        def barrier():
            drain()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        timer = sa.Timer(ctx)

        def timed(n, with_gather, compute=True):
            barrier()
            t0 = time.perf_counter()
            timer.start()
            for _ in range(n):
                step(with_gather, compute)
            drain()
            timer.stop()
            barrier()
            return time.perf_counter() - t0, timer.elapsed_ms()

        # ---- pre-conditioning = the sustained figure: >= --sustain-seconds of back-to-back launches on the same stream
        # BEFORE the warm-up and the timed steps, so that the timed region runs at the clock the chip HOLDS under this load
        # (DVFS), not on the ramp out of idle (every rank runs it; rank 0 reports its own) ----
        sustained, telemetry, tele, pre_s = None, None, None, 0.0
        for _ in range(2):   # (first launches: code objects load, LDS attributes are set)
            step(False)
        if a.sustain_seconds > 0:
            chunk = max(K, 20)
            if rank == 0:
                try:   # the device's sysfs directory by its PCI address (domain:bus:device.function)
                    pr = torch.cuda.get_device_properties(local)
                    tele = Telemetry("%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
                    if not tele.files:
                        tele = Telemetry("")   # (containers that remap PCI addresses: the first card)
                except Exception:
                    tele = None
            if tele is not None:
                tele.start()
            tot_ms, launches, t1 = 0.0, 0, time.perf_counter()
            while time.perf_counter() - t1 < a.sustain_seconds:
                timer.start()
                for _ in range(chunk):
                    step(False)
                timer.stop()
                tot_ms += timer.elapsed_ms()   # waits for the chunk; the next one follows within microseconds
                launches += chunk
            pre_s = time.perf_counter() - t1
            last_ms = timer.elapsed_ms() / chunk
            sustained = {"ms_per_launch": tot_ms / launches, "launches": launches, "last_chunk_ms_per_launch": last_ms}
        # ---- W warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides ----
        for _ in range(W):
            step(gather)
        wall, dev_ms = timed(K, gather)
        if tele is not None:
            telemetry = tele.stop()   # (stopped only now: nothing but launches between the pre-conditioning and the timed region)
        # ---- verification (outside the timed region): the LAST timed step's output of a few channels vs the CPU oracle ----
        verified = verify_last(a, w, it[0], rank, np, torch) if K + W >= 2 else None
        if a.dump_output and rank == 0:
            torch.cuda.synchronize()
            np.save(a.dump_output, (gathered if gathered is not None else w.outs[(it[0] - 1) & 1]).cpu().numpy())

        # the same K steps without the gather (the per-GPU kernel alone) and the gather alone (no kernel), reported
        # beside the headline of a gathered run
        no_gather, gather_only = None, None
        if gather:
            wl_, dm_ = timed(K, False)
            no_gather = {"wall": wl_, "dev_ms": dm_}
            wl_, dm_ = timed(K, True, compute=False)
            gather_only = {"wall": wl_}

        # ---- the other BASELINE configs, each to the same recipe (one GPU, default run only) ----
        configs, details = None, None
        if world == 1 and not use_dist and not a.workload and not a.no_configs and not a.global_channels and a.buffers == 1:
            w.node.close()
            w.ins = w.outs = None
            torch.cuda.empty_cache()
            configs, details = [], []
            for spec in CONFIG_SPECS:
                try:
                    e, det = measure_config(a, spec, sa, torch, shard, ctx, dev, np, stream)
                    configs.append(e)
                    details.append(det)
                except Exception as e:   # (one entry that fails says so; the headline and the other entries stand)
                    configs.append({"id": spec["id"], "error": "%s: %s" % (type(e).__name__, str(e)[:200])})
                    details.append({"id": spec["id"]})
                    torch.cuda.empty_cache()

    host_coll = use_dist and a.backend != "nccl"
    cdev = "cpu" if host_coll else dev
    mine = dev_ms / K
    red = torch.tensor([wall, no_gather["wall"] if no_gather else 0.0, gather_only["wall"] if gather_only else 0.0, mine, -mine],
                       dtype=torch.float64, device=cdev)
    ranks_seen = torch.ones(1, dtype=torch.float64, device=cdev)
    if use_dist:
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks_seen, op=dist.ReduceOp.SUM)   # (counted by the collective itself, not read from the launcher's env)
    wall, wall_ng, wall_go = float(red[0].item()), float(red[1].item()), float(red[2].item())
    launch_max, launch_min = float(red[3].item()), -float(red[4].item())

    if rank == 0:
        alg_bytes, in_bytes = w.alg_bytes, w.in_bytes
        total_samples = float(C) * N * K * world
        value = total_samples / wall / 1e6
        per_launch_s = dev_ms / 1e3 / K
        achieved = C * N * alg_bytes / per_launch_s / 1e9
        res = {
            "metric": "Msamples/s through baseband->FIR->demod chain",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(wall / K * 1e3, 4), "higher_is_better": True, "scaling": "strong" if a.global_channels else "weak",
            "vs_baseline": None, "dtype": w.dtype, "data": "synthetic",
            "preconditioned_s": round(pre_s, 2),
            "config": {"workload": w.desc, "workload_key": w.key, "channels_per_gpu": C, "samples_per_channel_per_step": N,
                       "global_channels": C * world,
                       "input": "int16 (real)" if wl == "bb_real_fm" else {2.0: "complex<uint8>", 4.0: "complex<int16>"}.get(in_bytes, "complex<float>"),
                       "parallelism": "channel-sharded x%d, %s" % (world, "output gathered on rank 0 every step (RCCL, side stream, double-buffered)" if gather else "no data-path collective")},
            "roofline": roofline_block(w, C, N, per_launch_s),
        }
        res["roofline"].update({"kernels_per_step": w.kernels, "algorithmic_bytes_per_launch": C * N * alg_bytes,
                                "algorithmic_bytes_per_sample": alg_bytes,
                                "hbm_read_frac": round(C * N * in_bytes / per_launch_s / 1e9 / HBM_PEAK_GBS, 5),
                                "per_gpu_msamples_s": round(C * N / per_launch_s / 1e6, 2),
                                "ranks_seen": int(round(float(ranks_seen.item()))),
                                "avg_launch_ms_min_rank": round(launch_min, 4), "avg_launch_ms_max_rank": round(launch_max, 4)})
        if verified is not None:
            res["verified"] = verified["ok"]
            res["verify"] = verified
        if no_gather:
            rf = res["roofline"]
            rf["with_gather_msamples_s"] = round(value, 2)
            rf["without_gather_msamples_s"] = round(total_samples / wall_ng / 1e6, 2)
            rf["without_gather_ms_per_step"] = round(wall_ng / K * 1e3, 4)
            rf["without_gather_avg_launch_ms"] = round(no_gather["dev_ms"] / K, 4)
            # what the gather moves: every rank's rows land on the root each step; the remote part crosses xGMI
            row_bytes = w.outs[0].shape[1] * w.outs[0].element_size()
            gb, gb_remote = world * C * row_bytes, (world - 1) * C * row_bytes
            rf["gather_bytes_per_step"], rf["gather_remote_bytes_per_step"] = gb, gb_remote
            rf["gather_only_ms_per_step"] = round(wall_go / K * 1e3, 4)
            rf["gather_gbs"] = round(gb / (wall_go / K) / 1e9, 2)                 # the gather alone, back to back (root ingress incl. its own copy)
            rf["gather_remote_gbs"] = round(gb_remote / (wall_go / K) / 1e9, 2)   # ... the part that crosses links
            rf["gather_gbs_in_step"] = round(gb / (wall / K) / 1e9, 2)            # what the gathered steps delivered
            rf["claim"] = "without_gather_* = how the kernel scales; value / with_gather_* = what the root receives (root-ingress-bound at 8 GPUs, DESIGN.md §5)"
        if sustained:
            rf = res["roofline"]
            rf["sustained_ms_per_launch"] = round(sustained["ms_per_launch"], 4)
            rf["sustained_frac"] = round(C * N * alg_bytes / (sustained["ms_per_launch"] / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
            rf["sustained_launches"] = sustained["launches"]
            rf["sustained_last_chunk_ms_per_launch"] = round(sustained["last_chunk_ms_per_launch"], 4)
            rf["sustained_per_gpu_msamples_s"] = round(C * N / (sustained["ms_per_launch"] / 1e3) / 1e6, 2)
            if wall * 1e3 < 20.0:   # a timed region this short: the same metric over the pre-conditioning launches beside it
                res["value_sustained"] = round(C * N * world / (sustained["ms_per_launch"] / 1e3) / 1e6, 2)
                res["value_sustained_note"] = "timed region %.1f ms < 20 ms: the same metric over the %.0f s of pre-conditioning launches" % (wall * 1e3, a.sustain_seconds)
        try:   # what this box's HBM delivers to a pure read of the same buffers, beside the nominal peak
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
        if wl == "fbb_f32" and C == 1:   # BASELINE config 2: one channel — what a buffer costs, and against real time
            res["roofline"]["per_buffer_us"] = round(per_launch_s * 1e6, 2)
            res["roofline"]["real_time_factor"] = round((N / FS) / per_launch_s, 1)
        if telemetry:   # the clock and power the chip HELD over the pre-conditioning and the timed steps (rank 0's device)
            rf = res["roofline"]
            rf["sclk_mhz"], rf["sclk_mhz_min"], rf["power_w"] = telemetry["sclk_mhz"], telemetry["sclk_mhz_min"], telemetry["power_w"]
            rf["power_cap_w"] = telemetry["power_cap_w"]   # (the socket's limit: a kernel at it runs at the clock the limit allows)
            rf["telemetry"] = {"samples": telemetry["samples"]}   # (amdgpu hwmon files, read over the pre-conditioning + warm-up + timed steps)
            sys.stderr.write("bench.py telemetry source: %s\n" % json.dumps(telemetry["source"]))
        if configs is not None:
            res["configs"] = configs
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the CPU baselines run once the GPU work is over and the process group is gone (the other ranks have left: the
        # host cores are idle), on rank 0, for every N
        if not a.no_cpu_baseline:
            cb = cpu_baseline(wl, a.cpu_seconds)
            if cb:
                res["cpu_baseline"] = cb
            for spec, e, det in zip(CONFIG_SPECS, configs or [], details or []):
                if "error" in e or ("cpu_chain" in spec and spec["cpu_chain"] is None):
                    continue
                try:
                    cb = cpu_baseline(spec["workload"], a.config_cpu_seconds, all_cores=False, chain=spec.get("cpu_chain"))
                except Exception as ex:
                    cb = {"error": str(ex)[:120]}
                if cb:
                    det["cpu_baseline"] = cb
                    e["cpu"] = {k: cb[k] for k in ("value", "cores", "kind", "error") if k in cb}
                    if "value" in cb and cb["value"]:
                        e["cpu"]["x"] = round(e["value"] / cb["value"])   # (GPU whole-launch rate / one host core: a ratio, not a quality claim)
        res["run_s"] = round(time.perf_counter() - t_run0, 1)
        line = json.dumps(res)
        if configs is not None and len(line) > 7600:   # the driver keeps the last 8 KB of stdout: the optional keys go first
            for drop in ("sustained_ms_per_launch", "mfma_per_slice", "kernels_per_step"):
                for e in configs:
                    e.get("roofline", {}).pop(drop, None)
                    e.get("roofline", {}).get("compute", {}).pop(drop, None)
                line = json.dumps(res)
                if len(line) <= 7600:
                    break
        if details:
            sys.stderr.write("bench.py details: " + json.dumps({"configs": details}) + "\n")
        print(line, flush=True)


class _CommDesign:
    """build_workload's `shard` seam in --comm sdrhip mode: the design tensors a rank builds its plan from are the bytes
    that travelled to that rank through sdrhip_comm_broadcast."""

    def __init__(self, received):
        self.received = received

    def broadcast_design(self, tensors, src=0):
        for t, b in zip(tensors, self.received):
            t.copy_(b.view(t.dtype).reshape(t.shape))
        return tensors


def run_sdrhip(a):
    """BASELINE config 5 through the C ABI's multi-GPU path: ONE process, --gpus N rank
    contexts from sdrhip_comm_create (one per device; with --force-device D all on device D, where the library uses its
    same-device transport), the design broadcast with sdrhip_comm_broadcast, every step = one *_process_dev per rank on
    the rank's own stream + one sdrhip_comm_gather of the demodulated rows to rank 0. torch only allocates and fills the
    device buffers."""
    import numpy as np
    import torch
    import libsdr_amd as sa

    if not torch.cuda.is_available():
        raise BenchError("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    G = a.gpus
    ndev = torch.cuda.device_count()
    devices = [a.force_device] * G if a.force_device >= 0 else list(range(G))
    if max(devices) >= ndev:
        raise BenchError("--comm sdrhip --gpus %d but only %d HIP device(s) visible (one box: --force-device 0)" % (G, ndev))
    wl = a.workload or "iqbb_usb"
    if wl not in ("iqbb_usb", "iqbb_fm", "iqbb_fm_cu8"):
        raise BenchError("--comm sdrhip runs the iqbb_* workloads (config 5's chain)")
    if a.global_channels:
        if a.global_channels % G:
            raise BenchError("--global-channels %d is not a multiple of the %d ranks" % (a.global_channels, G))
        a.channels = a.global_channels // G
    C, W, K = a.channels, a.warmup, a.steps
    t_run0 = time.perf_counter()
    comm = sa.Comm(devices)
    tdev = [torch.device("cuda", d) for d in devices]
    # ---- config(): designed once (rank 0), broadcast to every rank's device through the library ----
    taps0 = torch.from_numpy(sa.design_iqbb_taps(a.fc, a.width, a.fs, a.order)).to(tdev[0])
    lut0 = torch.from_numpy(sa.design_freqshift_lut_i16()).to(tdev[0])
    recv = []
    for t0_ in (taps0, lut0):
        bufs = [t0_ if r == 0 else torch.zeros_like(t0_, device=tdev[r]) for r in range(G)]
        for d in set(devices):
            torch.cuda.synchronize(d)
        comm.broadcast([b.data_ptr() for b in bufs], t0_.numel() * t0_.element_size(), 0)
        recv.append(bufs)
    comm.synchronize()
    ws, timers = [], []
    for r in range(G):
        torch.cuda.set_device(devices[r])
        ws.append(build_workload(a, wl, sa, torch, _CommDesign([recv[0][r], recv[1][r]]), comm.ctx[r], tdev[r], r, 2))
        timers.append(sa.Timer(comm.ctx[r]))
    N, w0 = ws[0].N, ws[0]
    row_elems = w0.outs[0].shape[1]
    row_bytes = row_elems * w0.outs[0].element_size()
    torch.cuda.set_device(devices[0])
    gathered = torch.zeros((G * C, row_elems), dtype=w0.outs[0].dtype, device=tdev[0])
    for d in set(devices):
        torch.cuda.synchronize(d)
    it = [0]

    def step(with_gather, compute=True):
        i = it[0]
        o = i & 1
        comm.gather_wait(o)   # (enqueued on the ranks' streams: the gather of step i - 2 has read outs[o]; a slot never begun: nothing)
        if compute:
            for r in range(G):
                ws[r].run(i % a.batches, o)
            it[0] = i + 1
        if with_gather:   # on the comm's own streams, behind this step's kernels: the next step's kernels start meanwhile
            comm.gather_begin(o, [ws[r].outs[o].data_ptr() for r in range(G)], [C * row_bytes] * G, gathered.data_ptr(), 0)

    def timed(n, with_gather, compute=True):
        comm.synchronize()
        t0 = time.perf_counter()
        for t in timers:
            t.start()
        for _ in range(n):
            step(with_gather, compute)
        comm.gather_wait(0)
        comm.gather_wait(1)
        for t in timers:
            t.stop()
        comm.synchronize()
        return time.perf_counter() - t0, [t.elapsed_ms() for t in timers]

    for _ in range(2):
        step(False)
    pre_s, sustained = 0.0, None
    if a.sustain_seconds > 0:
        chunk, tot, n, t1 = max(K, 20), 0.0, 0, time.perf_counter()
        while time.perf_counter() - t1 < a.sustain_seconds:
            _, ms = timed(chunk, False)
            tot += ms[0]
            n += chunk
        pre_s, sustained = time.perf_counter() - t1, tot / n
    for _ in range(W):
        step(True)
    wall, ms = timed(K, True)
    verified = None
    if K + W >= 2 and not a.no_verify:
        oks = []
        for r in range(G):
            torch.cuda.set_device(devices[r])
            v = verify_last(a, ws[r], it[0], r, np, torch)
            oks.append(v["ok"] if v else None)
        verified = {"ok": None if all(o is None for o in oks) else bool(all(o for o in oks if o is not None) and any(oks)),
                    "ranks": G, "channels_per_rank": min(a.verify_channels, C), "mode": "last timed step of every rank vs CPU oracle", "tolerance": "bit-exact"}
        o = (it[0] - 1) & 1   # ... and the root's landing zone holds exactly the ranks' rows, in global channel order
        torch.cuda.set_device(devices[0])
        same = all(bool(torch.equal(gathered[r * C:(r + 1) * C].cpu(), ws[r].outs[o].cpu())) for r in range(G))
        verified["gathered_equals_rank_rows"] = same
        if not same:
            verified["ok"] = False
    if a.dump_output:
        np.save(a.dump_output, gathered.cpu().numpy())
    wall_ng, ms_ng = timed(K, False)
    wall_go, _ = timed(K, True, compute=False)
    alg, inb = w0.alg_bytes, w0.in_bytes
    per_launch_s = ms[0] / 1e3 / K
    total = float(C) * N * K * G
    gb, gb_remote = G * C * row_bytes, (G - 1) * C * row_bytes
    res = {"metric": "Msamples/s through baseband->FIR->demod chain", "value": round(total / wall / 1e6, 2), "unit": "Msamples/s",
           "n_gpus": len(set(devices)), "ranks": G, "steps": K, "warmup": W, "ms_per_step": round(wall / K * 1e3, 4), "higher_is_better": True,
           "scaling": "strong" if a.global_channels else "weak", "vs_baseline": None, "dtype": w0.dtype, "data": "synthetic",
           "preconditioned_s": round(pre_s, 2), "comm": "sdrhip",
           "config": {"workload": w0.desc, "workload_key": w0.key, "channels_per_gpu": C, "samples_per_channel_per_step": N, "global_channels": C * G,
                      "input": {2.0: "complex<uint8>", 4.0: "complex<int16>"}.get(inb),
                      "parallelism": "ONE process, %d rank contexts on devices %s through sdrhip_comm_* (transport: %s); design broadcast, "
                                     "output gathered on rank 0 every step" % (G, devices, comm.transport)},
           "roofline": {"bound": w0.bound, "achieved": round(C * N * alg / per_launch_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(C * N * alg / per_launch_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": None, "kernel": w0.kernels[0],
                        "kernels_per_step": w0.kernels, "algorithmic_bytes_per_sample": alg,
                        "avg_launch_ms": round(ms[0] / K, 4), "avg_launch_ms_min_rank": round(min(ms) / K, 4), "avg_launch_ms_max_rank": round(max(ms) / K, 4),
                        "avg_launch_note": "HIP events on each rank's stream around the K steps; the gather runs on the comm's own streams (sdrhip_comm_gather_begin / _wait), double-buffered",
                        "ranks_seen": G,
                        "with_gather_msamples_s": round(total / wall / 1e6, 2), "without_gather_msamples_s": round(total / wall_ng / 1e6, 2),
                        "without_gather_ms_per_step": round(wall_ng / K * 1e3, 4), "without_gather_avg_launch_ms": round(ms_ng[0] / K, 4),
                        "gather_bytes_per_step": gb, "gather_remote_bytes_per_step": gb_remote,
                        "gather_only_ms_per_step": round(wall_go / K * 1e3, 4), "gather_gbs": round(gb / (wall_go / K) / 1e9, 2),
                        "gather_remote_gbs": round(gb_remote / (wall_go / K) / 1e9, 2), "gather_gbs_in_step": round(gb / (wall / K) / 1e9, 2)}}
    if sustained:
        res["roofline"]["sustained_ms_per_launch"] = round(sustained, 4)
    if verified is not None:
        res["verified"], res["verify"] = verified["ok"], verified
    for w in ws:
        w.node.close()
    del timers
    comm.close()
    if not a.no_cpu_baseline:
        cb = cpu_baseline(wl, a.cpu_seconds)
        if cb:
            res["cpu_baseline"] = cb
    res["run_s"] = round(time.perf_counter() - t_run0, 1)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
