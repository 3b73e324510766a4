#!/usr/bin/env python3
"""Phase map of a fused-FFT kernel's assembly: where its spills, barriers, global loads / stores and branches sit.
    tools/k7_asm_map.py <file.s> <mangled-name-substring>"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and name in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end]
ev = []
for i, l in enumerate(body):
    t = l.strip()
    if t.startswith("scratch_"): ev.append((i, "spill-store" if "store" in t else "spill-load"))
    elif t.startswith("s_barrier"): ev.append((i, "BARRIER"))
    elif t.startswith("global_load"): ev.append((i, "gload"))
    elif t.startswith("global_store"): ev.append((i, "gstore"))
    elif t.startswith("ds_read") or t.startswith("ds_load"): ev.append((i, "ds_read"))
    elif t.startswith("ds_write") or t.startswith("ds_store"): ev.append((i, "ds_write"))
    elif re.match(r"^\.LBB\S+:", t): ev.append((i, t.split()[0]))
    elif t.startswith("s_cbranch") or t.startswith("s_branch"): ev.append((i, t.split()[0] + " " + t.split()[-1]))
out = []; last = None; cnt = 0; st = 0
for i, e in ev:
    if e == last: cnt += 1
    else:
        if last: out.append(f"{st:6d}: {last} x{cnt}")
        last = e; cnt = 1; st = i
out.append(f"{st:6d}: {last} x{cnt}")
print(len(body), "lines"); print("\n".join(out))
