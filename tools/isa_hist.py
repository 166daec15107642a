#!/usr/bin/env python3
"""opcode histogram per large basic block of one kernel: tools/isa_hist.py file.hip mangled-substring [min_block_size]"""
import re, subprocess, sys
from collections import Counter
src, pat = sys.argv[1], sys.argv[2]
minb = int(sys.argv[3]) if len(sys.argv) > 3 else 60
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-S",
                       "--cuda-device-only", src, "-o", "/tmp/isa.s"], stderr=subprocess.DEVNULL)
lines = open("/tmp/isa.s").read().split("\n")
start = end = None
for i, l in enumerate(lines):
    if start is None and re.match(r"^_Z\w+:", l) and pat in l:
        start = i
    if start is not None and "s_endpgm" in l and i > start:
        end = i
        break
blocks, cur = [], ["entry", []]
for l in lines[start:end]:
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append(cur)
        cur = [l.split(":")[0], []]
    elif l.startswith("\t") and not l.strip().startswith((".", ";")):
        cur[1].append(l.strip().split()[0])
blocks.append(cur)
for name, ins in blocks:
    if len(ins) >= minb:
        c = Counter(re.sub(r"_e32$|_e64$", "", i) for i in ins)
        print(name, len(ins), " ".join("%s:%d" % kv for kv in c.most_common(40)))
