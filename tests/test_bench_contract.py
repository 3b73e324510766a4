"""bench.py's one-line JSON contract (driver side): keys, types and the roofline / cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "i16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    # `bound` names what limits the kernel (the headline's 127-tap plan: the socket's power limit at the matrix pipe); achieved / peak /
    # frac stay the HBM figures the contract asks for, and `compute` prices the limiting pipe against its own peak
    assert rf["bound"] == "power/mfma" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["achieved"] > 0
    assert rf["compute"]["unit"] == "TOPS int8 MFMA" and 0 < rf["compute"]["frac"] < 1 and rf["compute"]["mfma_per_slice"] == 28
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb


def _error_line(r):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout + r.stderr)[-3000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "error"):
        assert k in d, k
    assert d["value"] is None and d["error"]
    return d


def test_bench_refuses_without_gpu():
    """No CPU fallback: on a machine without a HIP device bench.py prints ONE JSON line with an `error` key and a null
    value, and exits non-zero."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert "no CPU fallback" in _error_line(r)["error"]


def test_bench_more_gpus_than_devices_is_one_error_line():
    """`bench.py --gpus 8` on a node with fewer devices: no ranks are started, the line says so (the driver's scaling run
    must never end in a bare traceback)."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("eight devices are present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    d = _error_line(r)
    assert d["n_gpus"] == 8 and "visible" in d["error"]


def test_bench_gpus_flag_spawns_ranks_without_gpu():
    """`bench.py --gpus 2` outside torchrun must start 2 child ranks (torch.distributed.run; --force-device skips the
    parent's device count); here, with no GPU, every rank stops with the no-fallback message, rank 0 prints the error
    line and the parent hands back a non-zero exit code."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--force-device", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    txt = r.stderr + r.stdout
    assert r.returncode != 0 and "no CPU fallback" in txt
    assert "bench.py rank 1" in txt, txt[-3000:]
    assert "no CPU fallback" in _error_line(r)["error"]


@pytest.mark.gpu
def test_bench_strong_scaling_line():
    """--global-channels G: the job is fixed (BASELINE config 5's 8192 channels), a rank takes G / N of them, the line
    says `"scaling": "strong"` and counts G channels."""
    d = _bench(["--global-channels", "64", "--samples", "16384", "--steps", "3", "--warmup", "1", "--sustain-seconds", "0", "--no-cpu-baseline"])
    assert d["scaling"] == "strong" and d["config"]["global_channels"] == 64 and d["config"]["channels_per_gpu"] == 64
    assert d["verified"] is True


@pytest.mark.gpu
def test_bench_line_carries_clock_and_power_when_the_box_exposes_them():
    """roofline.sclk_mhz / power_w: averaged over the sustained phase from amdgpu's sysfs files by a side thread (keys
    present whenever telemetry was readable; values plausible)."""
    d = _bench(["--steps", "20", "--warmup", "5", "--sustain-seconds", "1", "--no-cpu-baseline", "--no-verify"])
    rf = d["roofline"]
    if "telemetry" not in rf:
        pytest.skip("no readable amdgpu hwmon files on this box")
    assert rf["telemetry"]["samples"] >= 5
    if rf["sclk_mhz"] is not None:
        assert 300 <= rf["sclk_mhz"] <= 3000
    if rf["power_w"] is not None:
        assert 50 <= rf["power_w"] <= 2000


def _bench(args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, (r.stderr + r.stdout)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["iqbb_fm", "iqbb_usb"])
def test_two_ranks_through_hip_nodes_match_single_process(tmp_path, workload):
    """N>1 path on hardware (BASELINE config 5's shape): `bench.py --gpus 2` spawns two ranks that both
    drive the HIP nodes on device 0 (gloo rendezvous: RCCL refuses two ranks on one device), design broadcast from
    rank 0, channel shards of 8, output gathered on rank 0 — and the gathered rows must equal the single-process
    run over the same 16 global channels bit for bit."""
    import numpy as np
    common = ["--workload", workload, "--samples", "8192", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
              "--sustain-seconds", "0"]
    two = str(tmp_path / "two.npy")
    one = str(tmp_path / "one.npy")
    d2 = _bench(["--gpus", "2", "--backend", "gloo", "--force-device", "0", "--channels", "8", "--gather",
                 "--dump-output", two] + common)
    d1 = _bench(["--gpus", "1", "--channels", "16", "--dump-output", one] + common)
    assert d2["n_gpus"] == 2 and d2["roofline"]["ranks_seen"] == 2 and d2["config"]["global_channels"] == 16
    assert d1["n_gpus"] == 1 and d1["config"]["global_channels"] == 16
    a, b = np.load(two), np.load(one)
    assert a.shape == b.shape and a.dtype == np.int16 and np.array_equal(a, b)
    assert np.count_nonzero(a) > a.size // 2   # real demodulated output, not zeros


@pytest.mark.gpu
def test_eight_ranks_preflight_through_the_real_launcher(tmp_path):
    """Multi-GPU pre-flight at the REAL rank count on one device (the driver's scaling run is first contact with 8
    devices): `bench.py --gpus 8 --global-channels 8192` through the same parent -> torch.distributed.run -> 8 ranks path the
    driver's command takes (gloo rendezvous, every rank on device 0), BASELINE config 5's chain with the gather — all 8 ranks
    counted by the all-reduce, per-rank launch times present, and the gathered 8192 rows equal the single-rank run bit for bit."""
    import numpy as np
    common = ["--samples", "16384", "--steps", "3", "--warmup", "1", "--batches", "2", "--no-cpu-baseline", "--sustain-seconds", "0", "--verify-channels", "2"]
    eight, one = str(tmp_path / "eight.npy"), str(tmp_path / "one.npy")
    d8 = _bench(["--gpus", "8", "--backend", "gloo", "--force-device", "0", "--global-channels", "8192", "--dump-output", eight] + common, timeout=1500)
    assert d8["n_gpus"] == 8 and d8["scaling"] == "strong" and d8["config"]["global_channels"] == 8192 and d8["config"]["channels_per_gpu"] == 1024
    rf = d8["roofline"]
    assert rf["ranks_seen"] == 8 and 0 < rf["avg_launch_ms_min_rank"] <= rf["avg_launch_ms_max_rank"]
    assert rf["without_gather_msamples_s"] > 0 and d8["verified"] is True
    d1 = _bench(["--gpus", "1", "--workload", "iqbb_usb", "--channels", "8192", "--dump-output", one] + common)
    a, b = np.load(eight), np.load(one)
    assert a.shape == b.shape and a.shape[0] == 8192 and a.shape[1] >= 2048 and np.array_equal(a, b) and np.count_nonzero(a) > a.size // 2
    assert rf["gather_bytes_per_step"] == a.size * 2


@pytest.mark.gpu
def test_eight_rank_contexts_through_the_c_abi(tmp_path):
    """The same job through `--comm sdrhip`: ONE process, 8 rank contexts from sdrhip_comm_create (all on device 0: the
    library's same-device transport), design broadcast, the overlapped sdrhip_comm_gather_begin / _wait every step."""
    import numpy as np
    common = ["--samples", "16384", "--steps", "3", "--warmup", "1", "--batches", "2", "--no-cpu-baseline", "--sustain-seconds", "0", "--verify-channels", "2"]
    eight, one = str(tmp_path / "eight.npy"), str(tmp_path / "one.npy")
    d8 = _bench(["--comm", "sdrhip", "--gpus", "8", "--force-device", "0", "--global-channels", "8192", "--dump-output", eight] + common, timeout=1500)
    assert d8["ranks"] == 8 and d8["roofline"]["ranks_seen"] == 8 and d8["verified"] is True and d8["verify"]["gathered_equals_rank_rows"] is True
    assert 0 < d8["roofline"]["avg_launch_ms_min_rank"] <= d8["roofline"]["avg_launch_ms_max_rank"]
    d1 = _bench(["--gpus", "1", "--workload", "iqbb_usb", "--channels", "8192", "--dump-output", one] + common)
    a, b = np.load(eight), np.load(one)
    assert a.shape == b.shape and a.shape[0] == 8192 and np.array_equal(a, b)


@pytest.mark.gpu
def test_eight_gpus_asked_of_a_smaller_box_is_one_error_line():
    """`bench.py --gpus 8` (no --force-device) where fewer devices exist — the driver's scaling command on the wrong box: one
    JSON line with `error` and a null value, a non-zero exit, and no rank ever started (the parent counts devices from sysfs
    without touching HIP: nothing that has initialised the GPU is ever re-executed)."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("eight devices are present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    d = _error_line(r)
    assert d["n_gpus"] == 8 and "visible" in d["error"] and "rank" not in r.stderr.lower()


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["iqbb_fm", "iqbb_usb", "iqbb_fm_cu8", "iqbb_fm_cs8", "bb_real_fm", "fir255_fm", "fir127_fm", "fbb_f32", "fftconv", "fftconv_ola", "fftbank", "fm_demod", "subsample8"])
def test_every_workload_emits_its_line_and_verifies(workload):
    """Every `--workload` (the BASELINE configurations and the stand-alone kernels) runs on a small batch, prints one JSON
    line whose roofline names the kernel it timed, and the LAST timed step's output of the sampled channels equals the
    CPU oracle's (bit-exact for int16, <= 1e-5 for float / FFT): `verified` ties the number to the work."""
    d = _bench(["--workload", workload, "--channels", "16", "--samples", "32768", "--steps", "3", "--warmup", "2",
                "--no-cpu-baseline", "--sustain-seconds", "0", "--verify-channels", "4"])
    assert d["value"] > 0 and d["config"]["workload"] and d["roofline"]["kernel"] and 0 < d["roofline"]["frac"] < 1
    assert d["verified"] is True, d.get("verify")
    assert d["config"]["workload_key"].startswith(workload)
    assert "value_sustained" not in d   # (no sustained pass was asked for)


@pytest.mark.gpu
def test_short_timed_region_reports_the_sustained_value():
    d = _bench(["--channels", "64", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--sustain-seconds", "0.3", "--verify-channels", "2"])
    assert d["verified"] is True and d["value_sustained"] > 0 and d["roofline"]["sustained_frac"] > 0


@pytest.mark.gpu
def test_config2_single_channel_latency_line():
    """BASELINE config 2 as SURVEY §8d states it: one complex<float> channel, per-buffer latency beside Msamples/s."""
    d = _bench(["--workload", "fbb_f32", "--channels", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--sustain-seconds", "0"])
    assert d["verified"] is True and d["roofline"]["per_buffer_us"] > 0 and d["roofline"]["real_time_factor"] > 1


def test_traffic_is_keyed_by_workload(tmp_path, monkeypatch):
    """A PMC profile only counts for the workload it was cut on (a kernel name is shared by many workloads)."""
    sys.path.insert(0, ROOT)
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r99_pmc.json").write_text(json.dumps({"_meta": {"workload_key": "iqbb_fm/order127/d8/C1024/N65536"},
                                                   "iqbb_hot_kernel": {"derived": {"hbm_traffic_bytes_per_launch": 3.0e8}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.measured_traffic("iqbb_fm/order127/d8/C1024/N65536", ["iqbb_hot_kernel"])["bytes"] == 3.0e8
    assert bench.measured_traffic("iqbb_fm_cu8/order127/d8/C1024/N65536", ["iqbb_hot_kernel"]) is None
    assert bench.measured_traffic("iqbb_fm/order127/d8/C1024/N65536", ["iqbb_hot_kernel", "other_kernel"]) is None


@pytest.mark.gpu
def test_rccl_gather_path_runs_on_one_rank(tmp_path):
    """BASELINE config 5's collective as the multi-GPU line issues it — RCCL, side stream, `async_op=True`, byte views of the
    int16 rows landing in one preallocated tensor — on a ONE-rank RCCL group (a one-GPU box cannot hold two RCCL ranks): the
    gathered rows must equal the plain single-process run, and the line must carry both rates."""
    import numpy as np
    common = ["--workload", "iqbb_usb", "--channels", "16", "--samples", "32768", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
              "--sustain-seconds", "0", "--verify-channels", "4"]
    one, two = str(tmp_path / "plain.npy"), str(tmp_path / "rccl.npy")
    d1 = _bench(common + ["--dump-output", one])
    d2 = _bench(common + ["--force-dist", "--gather", "--dump-output", two])
    assert d2["verified"] is True and d2["roofline"]["ranks_seen"] == 1
    assert d2["roofline"]["with_gather_msamples_s"] > 0 and d2["roofline"]["without_gather_msamples_s"] > 0
    assert "gathered" in d2["config"]["parallelism"]
    a, b = np.load(one), np.load(two)
    assert a.shape == b.shape and np.array_equal(a, b) and np.count_nonzero(a) > a.size // 2


@pytest.mark.gpu
def test_default_line_carries_every_baseline_config():
    """The default one-GPU line (what the driver runs) appends BASELINE configs 2, 3, 4 (i)/(ii), 5 at G = 1 and the
    reference's own sdr_fm plan as "configs": each entry timed to the same recipe, checked against the oracle on its last
    step, with its roofline and the reference CPU chain beside it. (Channel counts capped here; the driver's run is full size.)"""
    d = _bench(["--channels", "16", "--samples", "32768", "--steps", "3", "--warmup", "2", "--sustain-seconds", "0.2",
                "--config-sustain-seconds", "0.05", "--cpu-seconds", "0.5", "--config-cpu-seconds", "0.3", "--configs-max-channels", "16",
                "--verify-channels", "4"])
    assert d["verified"] is True and d["preconditioned_s"] >= 0.2 and d["roofline"]["sustained_ms_per_launch"] > 0
    ids = [e["id"] for e in d["configs"]]
    assert ids == ["config1", "config2_c1", "config2_c1024", "config3", "config4_i_ola8192", "config4_ii_ols4097", "config5_g1", "sdr_fm_plan",
                   "bb_real_d20", "multi_buffer"]
    bounds = {"config1": "fp64-issue", "config2_c1": "hbm", "config2_c1024": "hbm", "config3": "fp64-issue", "config4_i_ola8192": "lds/issue",
              "config4_ii_ols4097": "lds/issue", "config5_g1": "power/mfma", "sdr_fm_plan": "valu-issue", "bb_real_d20": "valu-issue",
              "multi_buffer": "power/mfma"}
    for e in d["configs"]:
        assert "error" not in e, e
        assert e["verified"] is True, e
        assert e["value"] > 0 and e["ms_per_step"] > 0 and e["key"]
        rf = e["roofline"]
        assert rf["kernel"] and 0 < rf["frac"] < 1 and rf["bound"] == bounds[e["id"]]
        if e["id"] != "bb_real_d20":   # (no reference chain is compiled for that plan)
            assert e["cpu"]["value"] > 0 and e["cpu"]["kind"] in ("reference", "port")
    by = {e["id"]: e for e in d["configs"]}
    assert by["config2_c1"]["roofline"]["per_buffer_us"] > 0
    assert by["config5_g1"]["with_h2d_ms_per_step"] > by["config5_g1"]["ms_per_step"] and by["config5_g1"]["with_h2d_pcie_gbs"] > 0
    assert by["multi_buffer"]["key"].endswith("/B4/C16/N131072") and by["multi_buffer"]["roofline"]["kernel"] == "iqbb_hot_kernel"
    assert "kernels_per_step" not in by["multi_buffer"]["roofline"]   # (ONE launch: the hot kernel writes the buffer boundaries itself)
    assert by["config3"]["roofline"]["compute"]["unit"].startswith("T fp64")
    assert len(json.dumps(d)) < 7700   # (the driver keeps the last 8 KB of stdout: the whole line must fit)
    lim = d["cpu_baseline"]["all_cores"]["limits"]
    assert lim["cpus_in_affinity_mask"] >= 1 and "cgroup_quota_cores" in lim


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["iqbb_usb", "iqbb_fm"])
def test_comm_sdrhip_ranks_on_one_device_match_single_rank(tmp_path, workload):
    """BASELINE config 5 through the C ABI's multi-GPU path (`--comm sdrhip`): ONE process, 2 and 4 rank contexts from
    sdrhip_comm_create — all on device 0 here, where the library uses its same-device transport — design broadcast with
    sdrhip_comm_broadcast, sdrhip_comm_gather every step. The gathered rows equal the single-rank run over the same
    global channels bit for bit."""
    import numpy as np
    common = ["--workload", workload, "--samples", "8192", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--sustain-seconds", "0"]
    one = str(tmp_path / "one.npy")
    d1 = _bench(["--gpus", "1", "--channels", "16", "--dump-output", one] + common)
    b = np.load(one)
    for G in (2, 4):
        out = str(tmp_path / ("comm%d.npy" % G))
        d = _bench(["--comm", "sdrhip", "--gpus", str(G), "--force-device", "0", "--channels", str(16 // G), "--dump-output", out] + common)
        assert d["comm"] == "sdrhip" and d["ranks"] == G and d["config"]["global_channels"] == 16
        assert "same-device" in d["config"]["parallelism"]
        assert d["verified"] is True and d["verify"]["gathered_equals_rank_rows"] is True, d["verify"]
        rf = d["roofline"]
        assert rf["gather_bytes_per_step"] == b.size * 2 and rf["gather_gbs"] > 0 and rf["without_gather_msamples_s"] > 0
        a = np.load(out)
        assert a.shape == b.shape and a.dtype == np.int16 and np.array_equal(a, b)
    assert np.count_nonzero(b) > b.size // 2


@pytest.mark.gpu
def test_dist_line_reports_the_gather_and_the_ranks_it_counted():
    """The N > 1 line's extra keys, visible on a one-rank RCCL group: bytes and GB/s of the gather (alone and inside the
    steps), per-rank launch time min / max, and `ranks_seen` counted by an all-reduce."""
    d = _bench(["--workload", "iqbb_usb", "--channels", "16", "--samples", "32768", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                "--sustain-seconds", "0", "--verify-channels", "4", "--force-dist", "--gather"])
    rf = d["roofline"]
    assert rf["ranks_seen"] == 1 and rf["gather_bytes_per_step"] > 0 and rf["gather_bytes_per_step"] % 32 == 0
    assert rf["gather_gbs"] > 0 and rf["gather_gbs_in_step"] > 0 and rf["gather_only_ms_per_step"] > 0
    assert 0 < rf["avg_launch_ms_min_rank"] <= rf["avg_launch_ms_max_rank"]


def test_host_side_helpers_touch_no_gpu(tmp_path, monkeypatch):
    """cpu_limits(): affinity and cgroup quota as numbers; count_gpus_sysfs(): the KFD topology (nodes with SIMDs) or, where
    that is not readable, a short-lived child — the bench's parent process itself never loads torch or HIP before it starts
    the ranks (`spawn_ranks`)."""
    sys.path.insert(0, ROOT)
    import bench
    lim = bench.cpu_limits()
    assert lim["cpus_in_affinity_mask"] >= 1 and lim["cpus_online"] >= lim["cpus_in_affinity_mask"]
    assert lim["cgroup_quota_cores"] is None or lim["cgroup_quota_cores"] > 0
    n = bench.count_gpus_sysfs()
    assert n is None or n >= 0
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def spawn_ranks(a):"):src.index("def synth_cs16(")]
    assert "import torch" not in body and "libsdr_amd" not in body


@pytest.mark.gpu
@pytest.mark.parametrize("workload,extra", [("iqbb_fm", []), ("iqbb_usb", []), ("iqbb_fm_cu8", ["--order", "21", "--decim", "125", "--fs", "1e6", "--width", "12.5e3"]),
                                            ("bb_real_fm", ["--decim", "20"]), ("bb_real_fm", [])])
def test_multi_buffer_workloads_verify(workload, extra):
    """`--buffers 4`: four reference-sized buffers per channel in ONE launch (sdrhip_iqbb_i16_process_dev_multi); the LAST buffer of
    the last step against the oracle primed with the buffer before it — bit-exact, boundaries included."""
    d = _bench(["--workload", workload, "--buffers", "4", "--batches", "2", "--channels", "16", "--samples", "32768", "--steps", "3", "--warmup", "2",
                "--no-cpu-baseline", "--sustain-seconds", "0", "--verify-channels", "4"] + extra)
    assert d["verified"] is True, d.get("verify")
    assert "/B4/" in d["config"]["workload_key"] and d["config"]["samples_per_channel_per_step"] == 4 * 32768
    # FM at decimation 8: the hot kernel writes the buffers' first two outputs itself (one launch); any other decimation: a second, tiny launch
    ks = d["roofline"].get("kernels_per_step", [d["roofline"]["kernel"]])
    if workload == "iqbb_usb" or not extra:
        assert ks == ["iqbb_hot_kernel"], ks
    else:
        assert ks[-1] == "iqbb_fm_multi_fixup_kernel", ks
