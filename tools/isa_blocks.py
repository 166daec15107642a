#!/usr/bin/env python3
"""per-basic-block instruction statistics of one kernel: tools/isa_blocks.py file.hip mangled-substring"""
import re, subprocess, sys
from collections import Counter
src, pat = sys.argv[1], sys.argv[2]
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
tot = 0
for name, ins in blocks:
    tot += len(ins)
    if len(ins) > 40:
        c = Counter(ins)
        print("%-10s n=%4d valu=%4d pk=%3d ds=%3d vmem=%2d scratch=%2d salu=%3d" % (
            name, len(ins), sum(v for k, v in c.items() if k.startswith("v_")), sum(v for k, v in c.items() if k.startswith("v_pk")),
            sum(v for k, v in c.items() if k.startswith("ds_")), sum(v for k, v in c.items() if k.startswith(("global_", "buffer_"))),
            sum(v for k, v in c.items() if k.startswith("scratch")), sum(v for k, v in c.items() if k.startswith("s_"))))
print("total static instructions", tot)
