#!/usr/bin/env python3
"""Instruction mix per basic block that contains MFMAs, for one kernel of a device
assembly listing (hipcc -S --cuda-device-only).  usage: asm_mix.py file.s name-substring"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if sys.argv[2] in l and re.match(r"^_Z\w+:", l)][0]
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], None
for l in lines[start:end]:
    if re.match(r"^\.LBB\d+_\d+:", l):
        cur = {"label": l.split(":")[0], "mfma": 0, "valu": 0, "ds": 0, "salu": 0, "vmem": 0, "total": 0}
        blocks.append(cur)
        continue
    t = l.strip().split()[0] if l.strip() else ""
    if cur is None or not t or t[0] in ";.":
        continue
    cur["total"] += 1
    kind = ("mfma" if t.startswith("v_mfma") else "valu" if t.startswith("v_") else "ds" if t.startswith("ds_")
            else "salu" if t.startswith("s_") else "vmem" if t.startswith(("global_", "buffer_", "flat_")) else None)
    if kind:
        cur[kind] += 1
for b in blocks:
    if b["mfma"]:
        print(b)
for k in ("NumVgprs", "ScratchSize", "Occupancy"):
    print(next(l for l in lines[end:] if k in l).strip())
