#!/usr/bin/env python3
"""What is in the PMC's un-named VALU class?  The SQ_INSTS_VALU_* counters name FMA / ADD / MUL / TRANS (F32, F64), INT32, INT64 and CVT; at C3 43 % of the trace
kernel's vector instructions are in none of them ("OTHER": bench.py prices them at ONE cost).  This script splits that class STATICALLY: it counts, in the ISA of one
kernel instantiation (make -C raytrace_clj_amd/csrc asm), the vector opcodes no PMC class claims, by family, and prices every family with the cost measured for
its members (profiles/round3_ubench_valu_cost.txt, 4 waves per SIMD).  Static = every instruction counts once, whatever its loop executes (the hot loops -- the
node visit, the exact tests -- are a few hundred of the kernel's ~7 000 instructions); with `--hot` only the instructions between the phase markers of the
descent / leaf / big-primitive phases count (needs the -DRTMI_MARKERS asm: scripts/isa_phases.py).  The result is a cross-check of the single OTHER price.

usage: python scripts/isa_other_mix.py [asm file] [mangled kernel prefix]"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "rtmi-hip-amdgcn-amd-amdhsa-gfx950.s")
pat = sys.argv[2] if len(sys.argv) > 2 else "_ZN12_GLOBAL__N_112trace_kernelIdLb0ELi4ELb0ELb0ELb1ELb0ELb1E"  # <double, false, BVH, false, false, SLICE, false, LST>: the C3 kernel
lines, on = [], False
for l in open(path):
    if l.startswith(pat):
        on = True
    if on:
        lines.append(l.strip())
        if "s_endpgm" in l:
            break
# opcode families of the un-named class and the measured cost of a member (cycles per wave64 instruction per SIMD at 4 waves per SIMD)
FAMILIES = [
    ("v_mov / v_accvgpr", ("v_mov_b32", "v_mov_b64", "v_accvgpr"), 2.66),
    ("v_cndmask", ("v_cndmask",), 4.43),
    ("v_cmp / v_cmpx", ("v_cmp",), 4.50),
    ("v_min / v_max (f32, f64, int)", ("v_min_", "v_max_", "v_min3", "v_max3", "v_med3"), 4.30),
    ("v_readlane / v_readfirstlane / v_writelane", ("v_readlane", "v_readfirstlane", "v_writelane"), 4.45),
    ("v_div_scale / v_div_fixup / v_ldexp / v_frexp / v_rndne / v_floor / v_fract / v_trunc", ("v_div_scale", "v_div_fixup", "v_ldexp", "v_frexp", "v_rndne", "v_floor", "v_fract", "v_trunc", "v_ceil"), 4.95),
    ("v_perm / v_bfi / v_mbcnt / v_nop / other", ("v_perm", "v_bfi", "v_mbcnt", "v_nop", "v_swap"), 4.40),
]
NAMED = ("v_fma", "v_fmac", "v_mad_", "v_add", "v_sub", "v_mul", "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos", "v_cvt", "v_xor", "v_and", "v_or", "v_not",
         "v_lsh", "v_ash", "v_bfe", "v_alignbit", "v_pk_", "v_dot", "v_div_fmas", "v_xad", "v_bcnt", "v_ffb", "v_bfm", "v_sad")
cnt, named, unknown = collections.Counter(), 0, collections.Counter()
for t in lines:
    if not t or t[0] in ";./" or t.endswith(":"):
        continue
    op = t.split()[0]
    if not op.startswith("v_"):
        continue
    for name, prefixes, price in FAMILIES:
        if op.startswith(prefixes):
            cnt[name] += 1
            break
    else:
        if op.startswith(NAMED):
            named += 1
        else:
            unknown[op] += 1
total_other = sum(cnt.values()) + sum(unknown.values())
print("kernel %s: %d vector instructions, %d in a PMC class, %d un-named (%.0f %%)" % (pat[-40:], named + total_other, named, total_other, 100.0 * total_other / max(1, named + total_other)))
price = {name: p for name, _, p in FAMILIES}
cyc = 0.0
for name, c in cnt.most_common():
    print("  %-90s %5d  %5.1f %%  x %.2f cycles" % (name, c, 100.0 * c / total_other, price[name]))
    cyc += c * price[name]
for op, c in unknown.most_common():
    print("  %-90s %5d  %5.1f %%  x 4.40 cycles (not classified)" % (op, c, 100.0 * c / total_other))
    cyc += c * 4.4
print("static price of the un-named class: %.2f cycles per instruction (bench.py's single price: 3.9; without the moves: %.2f)" % (
    cyc / max(1, total_other), (cyc - cnt["v_mov / v_accvgpr"] * 2.66) / max(1, total_other - cnt["v_mov / v_accvgpr"])))
