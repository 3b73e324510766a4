#!/usr/bin/env python3
"""Static instruction buckets of the K1 hot kernel's tile loop, from the compiler's assembly:
    hipcc ... -S --cuda-device-only -o iqbb_i16.s libsdr_amd/csrc/iqbb_i16.hip ; tools/asm_buckets.py iqbb_i16.s
Phases are cut at markers in the instruction stream (s_setprio, the first v_perm_b32, the first / last MFMA, the
reciprocal of fm_phi's division). One pass of the loop body = one wave slice (512 samples)."""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
name = sys.argv[2] if len(sys.argv) > 2 else "iqbb_i16_hot_kernelILi9ELi2ELi5ELb1ELi1E"
start = next(i for i, l in enumerate(src) if l.startswith("_ZN") and name in l and l.rstrip().endswith(":") or (name in l and l.startswith("_ZN") and ":" in l))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
L = src[start:end]
op = lambda l: (re.match(r"\s+([a-z_0-9]+)", l) or [None, ""])[1]
idx = lambda pred, a=0: next(i for i in range(a, len(L)) if pred(op(L[i])))
prio = idx(lambda o: o == "s_setprio")
loop_top = max(i for i in range(prio) if L[i].startswith(".LBB") and i < prio - 0) if False else None
perm = idx(lambda o: o.startswith("v_perm_b32"), prio)
mf0 = idx(lambda o: o.startswith("v_mfma"), perm)
mf1 = max(i for i in range(len(L)) if op(L[i]).startswith("v_mfma"))
rcp = idx(lambda o: o.startswith("v_rcp"), mf1)
store = idx(lambda o: o.startswith("global_store"), rcp)
# loop top: the label the back edge after the store jumps to; bookkeeping = from there to the first v_perm
back = next(i for i in range(store, len(L)) if op(L[i]) in ("s_branch", "s_cbranch_execnz", "s_cbranch_vccnz", "s_cbranch_scc1", "s_cbranch_scc0"))
tgt = L[back].split()[-1]
top = next(i for i, l in enumerate(L) if l.startswith(tgt + ":"))
sumstart = mf1 + 1
fin = max(i for i in range(sumstart, rcp) if "sdwa" in L[i]) + 1 if any("sdwa" in L[i] for i in range(sumstart, rcp)) else rcp - 20
phases = [("tile bookkeeping (next slice, hot test, priority)", top, perm), ("raw window -> byte planes", perm, mf0 - 12),
          ("K loop: operand reads, MFMAs, DMA issue", mf0 - 12, mf1 + 1), ("recombine, >>14, rotate, window sum", sumstart, fin),
          ("trunc /8, fm_phi, neighbour angle, store", fin, store + 1)]
print("| phase | VALU (excl. MFMA) | MFMA | SALU | LDS | VMEM | largest VALU buckets |\n|---|---|---|---|---|---|---|")
tot = [0] * 5
for nm, a, b in phases:
    v = m = s = d = g = 0
    k = {}
    for l in L[a:b]:
        o = op(l)
        if not o:
            continue
        if o.startswith("v_mfma"): m += 1
        elif o.startswith("v_"):
            v += 1; o2 = o.replace("_e32", "").replace("_e64", ""); k[o2] = k.get(o2, 0) + 1
        elif o.startswith("s_"): s += 1
        elif o.startswith("ds_"): d += 1
        elif o.startswith(("global_", "buffer_")): g += 1
    top6 = sorted(((c, n) for n, c in k.items()), reverse=True)[:6]
    print("| %s | %d | %d | %d | %d | %d | %s |" % (nm, v, m, s, d, g, ", ".join("%s %d" % (n, c) for c, n in top6)))
    tot = [tot[0] + v, tot[1] + m, tot[2] + s, tot[3] + d, tot[4] + g]
print("| **one pass of the loop (static)** | **%d** | %d | %d | %d | %d | |" % tuple(tot))
