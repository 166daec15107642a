#!/usr/bin/env python3
"""Issue-cycle estimate per basic block of one kernel, with the gfx950 prices measured by tools/ubench/op_rate.hip:
   tools/isa_cost.py file.s mangled-substring [min_block_size]
full-rate VALU (add/sub/mul/fma/mov/and/xor/ashr with VGPR or literal operands) 2.5, SGPR operand / DPP / cmp / cndmask /
min / max / cvt / div_* / lshl 4.4, v_rcp 8.2"""
import re, sys
from collections import Counter
src, pat = sys.argv[1], sys.argv[2]
minb = int(sys.argv[3]) if len(sys.argv) > 3 else 60
lines = open(src).read().split("\n")
start = end = None
for i, l in enumerate(lines):
    if start is None and re.match(r"^_Z\w+:", l) and pat in l:
        start = i
    if start is not None and "s_endpgm" in l and i > start:
        end = i
        break
FULL = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32", "v_and_b32", "v_xor_b32", "v_or_b32",
        "v_add_u32", "v_sub_u32", "v_ashrrev_i32", "v_accvgpr")
def cost(l):
    op = l.split()[0]
    base = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", op)
    if not op.startswith("v_"):
        return 0.0, op.split("_")[0] + "_"
    if base in ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32"):
        return 8.2, "trans"
    if "dpp" in op or "wave_sh" in l or "row_sh" in l:
        return 4.4, "dpp"
    if base in FULL:
        ops = l.split(None, 1)[1] if len(l.split(None, 1)) > 1 else ""
        srcs = ops.split(",")[1:]
        if any(re.match(r"\s*[-|]*(s\d+|s\[|vcc|exec|ttmp)", s) for s in srcs):
            return 4.4, "sgpr-operand"
        return 2.5, "full"
    return 4.4, "half:" + base
blocks, cur = [], ["entry", []]
for l in lines[start:end]:
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append(cur)
        cur = [l.split(":")[0], []]
    elif l.startswith("\t") and not l.strip().startswith((".", ";")):
        cur[1].append(l.strip())
blocks.append(cur)
for name, ins in blocks:
    if len(ins) >= minb:
        tot = 0.0
        cls = Counter()
        ncls = Counter()
        for l in ins:
            c, k = cost(l)
            tot += c
            cls[k] += c
            ncls[k] += 1
        print("%s: %d insts, %.0f VALU issue cycles" % (name, len(ins), tot))
        for k, v in cls.most_common(30):
            print("   %-28s %5d insts %7.0f cycles" % (k, ncls[k], v))
