#!/usr/bin/env python3
"""Scan the gfx950 ISA of every kernel source for the store-data hazard found in round 5 (k_modular_vh.hip, vh_store): a vector-memory
store of more than 8 bytes whose data registers are written by a VALU instruction (or are the destination of an LDS / memory load)
within the next few instructions, counting only instructions that really occupy an issue cycle (an s_waitcnt does not).
    python tools/scan_store_hazard.py [window=2]"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WINDOW = int(sys.argv[1]) if len(sys.argv) > 1 else 2  # issue cycles the data registers must stay untouched (measured: 2 are enough)
sys.path.insert(0, ROOT)
from jxlatte_amd import build as B

def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()

def disassemble(src):
    stem = src[:-4]
    out = "/tmp/scan_%s_%d.s" % (stem, os.getpid())
    subprocess.run([B.HIPCC] + [f for f in B.FLAGS if f != "-fPIC"] + B.EXTRA.get(stem, []) + ["-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    return out

from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(max_workers=6) as ex:
    outs = list(ex.map(disassemble, B.SOURCES))
total = 0
for src, out in zip(B.SOURCES, outs):
    lines = [l.strip() for l in open(out) if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    hits = 0
    for i, l in enumerate(lines):
        m = re.match(r"(buffer|global|flat|scratch)_store_dwordx[34]\s+(.*)", l)
        if not m:
            continue
        ops = [o.strip() for o in m.group(2).split(",")]
        data = regs(ops[0]) if m.group(1) == "buffer" else regs(ops[1]) if len(ops) > 1 else set()
        real = 0
        for j in range(i + 1, min(i + 12, len(lines))):
            ins = lines[j]
            op = ins.split()[0]
            if op.startswith(("s_waitcnt", ";;")):
                continue
            if op.startswith("s_nop"):
                n = re.search(r"s_nop\s+(\d+)", ins)
                real += (int(n.group(1)) + 1) if n else 1
                if real >= WINDOW:
                    break
                continue
            if op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                dst = regs(ins.split()[1].rstrip(","))
                if dst & data:
                    hits += 1
                    print("%s: %s   <- %d issue cycle(s) later: %s" % (src, l, real, ins))
                    break
            real += 1
            if real >= WINDOW or op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                break
    total += hits
    os.remove(out)
    print("%-28s %d suspicious store(s)" % (src, hits))
print("total", total)
sys.exit(1 if total else 0)
