#!/usr/bin/env python3
"""Markdown table of one profiling round for profiles/README.md, from the files tools/prof_all.sh leaves
(`profiles/<tag>_<name>_kernel_stats.csv`, `<tag>_<name>_pmc.json`, `<tag>_bench.json`, `<tag>_bench_other_workloads.jsonl`).
usage: tools/profiles_table.py r14 [name ...]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1]
names = sys.argv[2:] or ("fm127 fm16 fm21 fm64 fm255 usb127 cu8 real fir255 fbb fftconv fftbank fmdemod sub8 sdrfm sdrrec sdrfmchain "
                         "wfmchain pocsag ssb").split()
lines = {}
for fn in (tag + "_bench.json", tag + "_bench_other_workloads.jsonl"):
    path = os.path.join(P, fn)
    if not os.path.exists(path):
        continue
    for l in open(path):
        if l.startswith("{"):
            d = json.loads(l)
            lines[d["config"]["workload_key"]] = d
print("| | kernels, average µs (kernel trace) | bench: ms per step sustained, frac of 8 TB/s | PMC traffic / algorithmic MB | shader clock, socket power (sustained phase) |")
print("|---|---|---|---|---|")
for n in names:
    pj = os.path.join(P, "%s_%s_pmc.json" % (tag, n))
    if not os.path.exists(pj):
        continue
    pm = json.load(open(pj))
    key = pm["_meta"].get("workload_key")
    with open(os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, n))) as f:
        rd = list(csv.reader(f))
    ks = []
    for r in rd[1:]:
        nm = r[0].replace("void ", "").replace("(anonymous namespace)::", "")
        nm = re.sub(r"\((?:sdrhip::|short|DeemphSpecArgs|SubArgs).*$", "", nm)
        nm = re.sub(r"\([^<>]*\)$", "", nm)
        ks.append("`%s` %.1f" % (nm.strip(), float(r[3]) / 1000))
    b = lines.get(key)
    if b is None:
        continue
    r = b["roofline"]
    tr = sum(v["derived"]["hbm_traffic_bytes_per_launch"] for k, v in pm.items() if k != "_meta" and "derived" in v) / 1e6
    print("| %s | %s | %.4f, %.1f %% | %.0f / %.0f | %s MHz, %s W |" % (n, " + ".join(ks), r["sustained_ms_per_launch"], 100 * r["sustained_frac"], tr,
                                                                    r["algorithmic_bytes_per_launch"] / 1e6, r.get("sclk_mhz"), r.get("power_w")))
