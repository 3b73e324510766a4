#!/usr/bin/env python3
"""Condenses a tools/prof.sh output directory (gpurun_out/prof_<tag>) into small, committable files under
profiles/: <tag>_kernel_stats.csv (our kernels only), <tag>_pmc.json (per-launch means per counter and
kernel) and a derived per-launch HBM traffic figure following MI355X_MICROARCH.md §HBM
(FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a streaming read -> doubled)."""
import collections
import re
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"   # (on the GPU box: a directory under gpurun_out/, copied into profiles/ afterwards)
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs(dst, exist_ok=True)
ours = ("iqbb", "bb_real", "fir_", "fftconv", "demod_", "deemph_", "subsample", "freqshift", "hist_roll", "fft_c2c")

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
rows = []
if stats:
    with open(stats[0]) as f:
        rd = list(csv.reader(f))
    rows = [rd[0]] + [r for r in rd[1:] if any(k in r[0] for k in ours)]
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows)

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if any(s in k for s in ours):
            m = re.search(r"(\w*(?:kernel|roll)\w*)", k)
            short = m.group(1) if m else k[:40]
            pmc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in pmc.items():
    out[k] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f = out[k]["FETCH_SIZE"]["mean"] * 1024.0
        w = out[k]["WRITE_SIZE"]["mean"] * 1024.0
        out[k]["derived"] = {"hbm_read_bytes_per_launch": 2.0 * f, "hbm_write_bytes_per_launch": w,
                             "hbm_traffic_bytes_per_launch": 2.0 * f + w,
                             "note": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), WRITE_SIZE as read; KiB units"}
# which workload the passes ran: bench.py's own JSON line in the kernel-trace pass (bench.py's measured_traffic() only
# accepts a profile for the workload it was cut on)
meta = {}
try:
    for line in open(os.path.join(src, "trace.log")):
        if line.startswith("{"):
            d = json.loads(line)
            meta = {"workload_key": d["config"].get("workload_key"), "workload": d["config"]["workload"],
                    "kernels_per_step": d["roofline"].get("kernels_per_step"), "verified": d.get("verified")}
except Exception as e:
    meta = {"error": str(e)[:100]}
out["_meta"] = meta
json.dump(out, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
for r in rows[:6]:
    print(r[:4])
print(json.dumps({k: v.get("derived") for k, v in out.items() if k != "_meta"}, indent=1), meta)
