#!/usr/bin/env python3
"""Basic-block instruction counts of one kernel in a hipcc -S listing: asm_blocks.py <file.s> <kernel-substring> [min_instr]"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and re.match(r"^_Z\S+:", l))
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or lines[i].strip() == "s_endpgm")
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", s):
        blocks.append((name, cur)); cur, name = [], s.split(":")[0]
    elif re.match(r"^(v_|s_|ds_|global_|buffer_|flat_)", s):
        cur.append(s)
blocks.append((name, cur))
tot = 0
for name, b in blocks:
    tot += len(b)
    if len(b) < mn: continue
    c = lambda p: sum(1 for i in b if re.match(p, i))
    last = b[-1].split()[0:2] if b else ""
    print("%-10s n=%4d valu=%4d mfma=%3d salu=%3d ds=%3d vmem=%3d  ends: %s" % (name, len(b), c(r"v_(?!mfma)"), c(r"v_mfma"), c(r"s_"), c(r"ds_"), c(r"(global|buffer|flat)_"), " ".join(last)))
print("total", tot)
