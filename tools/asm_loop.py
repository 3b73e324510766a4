#!/usr/bin/env python3
"""Static instruction mix of a hot kernel's slice loop (iqbb_hot.hpp) from the compiler's assembly.
    tools/asm_loop.py <file.s> <mangled-name-substring> [--dump]
The loop body holds TWO slices (window buffer parity 0 / 1): it is the span from the label the last backward branch
that encloses the first s_setprio jumps to, to that branch. Counts are per slice (body / 2)."""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and name in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
L = src[start:end]
op = lambda l: (re.match(r"\s+([a-z_0-9]+)", l) or [None, ""])[1]
labels = {l.split(":")[0]: i for i, l in enumerate(L) if re.match(r"\.LBB[0-9_]+:", l)}
prio = [i for i, l in enumerate(L) if op(l) == "s_setprio"]
best = None
for i, l in enumerate(L):
    o = op(l)
    if o.startswith("s_cbranch") or o == "s_branch":
        t = l.split()[-1]
        if t in labels and labels[t] < i and prio and labels[t] <= prio[0] <= i:
            if best is None or (i - labels[t]) > (best[1] - best[0]):
                best = (labels[t], i)
if best is None:
    sys.exit("no loop around the first s_setprio")
a, b = best
body = L[a:b + 1]
nsl = max(1, sum(1 for l in body if "s_setprio" in l) // 4) if False else 2
k = {}
cls = {"valu": 0, "mfma": 0, "salu": 0, "lds": 0, "vmem": 0, "other": 0}
for l in body:
    o = op(l)
    if not o:
        continue
    if o.startswith("v_mfma"): cls["mfma"] += 1
    elif o.startswith("v_"):
        cls["valu"] += 1
        o2 = o.replace("_e32", "").replace("_e64", "")
        if "dpp" in l and "dpp" not in o2: o2 += "(dpp)"
        k[o2] = k.get(o2, 0) + 1
    elif o.startswith("s_"): cls["salu"] += 1
    elif o.startswith("ds_"): cls["lds"] += 1; k[o] = k.get(o, 0) + 1
    elif o.startswith(("global_", "buffer_", "flat_")): cls["vmem"] += 1
    else: cls["other"] += 1
print("loop lines %d..%d of the kernel; per SLICE (body / %d):" % (a, b, nsl))
print("  " + ", ".join("%s %.1f" % (n, c / nsl) for n, c in cls.items()))
print("  " + ", ".join("%s %.1f" % (n, c / nsl) for n, c in sorted(k.items(), key=lambda x: -x[1])))
if "--dump" in sys.argv:
    print("\n".join(body))
