"""Instruction-class histogram of a kernel's gfx950 assembly (static counts per region).
usage: isa_histogram.py <file.s> <first line> <last line> [label]   (lines of the .s file; regions are picked by hand
from the barrier / s_setprio landmarks, see profiles/r03_smr_isa_histogram.txt)"""
import collections
import re
import sys

path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
label = sys.argv[4] if len(sys.argv) > 4 else "%d-%d" % (a, b)
lines = open(path).read().splitlines()[a - 1:b]


def klass(op):
    if re.match(r"v_(fma|fmac|mul|add|max|min|ldexp|rndne|floor|ceil|trunc|fract|frexp|div_fmas|div_fixup|div_scale|rcp|rsq|sqrt)_f64", op):
        return "fp64 arithmetic"
    if re.match(r"v_cvt_", op):
        return "conversions"
    if re.match(r"v_cmp|v_cmpx", op):
        return "vector compares"
    if re.match(r"v_cndmask", op):
        return "selects (v_cndmask)"
    if re.match(r"v_mov_b32_dpp|v_.*_dpp|v_permlane|v_readlane|v_readfirstlane|v_writelane|v_mbcnt|ds_bpermute|ds_swizzle", op) or op.endswith("_dpp"):
        return "cross-lane (dpp, permlane, readlane)"
    if re.match(r"v_mov|v_accvgpr", op):
        return "register moves"
    if re.match(r"v_", op):
        return "integer / bit VALU"
    if re.match(r"ds_", op):
        return "LDS"
    if re.match(r"global_|buffer_|flat_|scratch_", op):
        return "global memory"
    if re.match(r"s_load|s_buffer_load|s_memtime", op):
        return "scalar loads"
    if re.match(r"s_waitcnt|s_nop|s_barrier|s_setprio|s_sleep", op):
        return "waits / barriers"
    if re.match(r"s_cbranch|s_branch|s_endpgm|s_setpc|s_swappc|s_getpc", op):
        return "branches"
    if re.match(r"s_", op):
        return "scalar ALU"
    return "other"


cnt = collections.Counter()
ops = collections.Counter()
for ln in lines:
    ln = ln.split(";")[0].strip()
    if not ln or ln.endswith(":") or ln.startswith("."):
        continue
    op = ln.split()[0]
    cnt[klass(op)] += 1
    ops[op] += 1
tot = sum(cnt.values())
valu = sum(v for k, v in cnt.items() if k in ("fp64 arithmetic", "conversions", "vector compares", "selects (v_cndmask)",
                                               "cross-lane (dpp, permlane, readlane)", "register moves", "integer / bit VALU"))
print("== %s: %d instructions, %d VALU (%.0f %% fp64 arithmetic)" % (label, tot, valu, 100.0 * cnt["fp64 arithmetic"] / max(valu, 1)))
for k, v in cnt.most_common():
    print("   %-40s %6d  %5.1f %%" % (k, v, 100.0 * v / tot))
print("   top opcodes:", ", ".join("%s %d" % kv for kv in ops.most_common(14)))
