#!/usr/bin/env python3
"""What is in the PMC's un-named VALU class?  The SQ_INSTS_VALU_* counters name FMA / ADD / MUL / TRANS (F32, F64), INT32, INT64 and CVT; at C3 43 % of the trace
kernel's vector instructions are in none of them ("OTHER": bench.py prices them at ONE cost).  This script splits that class STATICALLY: it counts, in the ISA of one
kernel instantiation (make -C raytrace_clj_amd/csrc asm), the vector opcodes no PMC class claims, by family, and prices every family with the cost measured for
its members (profiles/round3_ubench_valu_cost.txt, 4 waves per SIMD).  Static = every instruction counts once, whatever its loop executes (the hot loops -- the
node visit, the exact tests -- are a few hundred of the kernel's ~7 000 instructions); with `--hot` only the instructions between the phase markers of the
descent / leaf / big-primitive phases count (needs the -DRTMI_MARKERS asm: scripts/isa_phases.py).  The result is a cross-check of the single OTHER price.

usage: python scripts/isa_other_mix.py [asm file] [mangled kernel prefix]
       python scripts/isa_other_mix.py --phases <markers asm> <phases file> [config tag]   (weighted: see the end of this file)"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = len(sys.argv) > 1 and sys.argv[1] == "--phases"
path = sys.argv[2] if PHASES else (sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "rtmi-hip-amdgcn-amd-amdhsa-gfx950.s"))
pat = (sys.argv[2] if len(sys.argv) > 2 and not PHASES else "_ZN12_GLOBAL__N_112trace_kernelIdLb0ELi4ELb0ELb0ELb1ELi0ELb1E")  # <double, false, BVH, false, false, SLICE, 0, LST>: the C3 kernel
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


# ---- weighted by the phase stamps ------------------------------------------------------------------------------------------------------------------
# `--phases <asm of a -DRTMI_MARKERS build> <profiles/roundN_phases.txt> [tag]`: the instructions between two phase markers (program order, scripts/isa_phases.py)
# are weighted with the number of times the diagnostic build's stamp of that phase executed in the launch (the `stamps` column of the phase table): an estimate
# of the EXECUTED mix of the un-named class -- program order is not control flow, so a phase's instructions that sit in a branch a trip skips are over-counted.
if len(sys.argv) > 3 and sys.argv[1] == "--phases":
    asm, table, tag = sys.argv[2], sys.argv[3], (sys.argv[4] if len(sys.argv) > 4 else "C3 bvh")
    names = {"loop/tail": "PH_LOOP", "refill: generate 64 camera rays": "PH_REFILL_GEN", "refill: claim + deal": "PH_REFILL_DEAL", "bvh: ray setup / resume": "PH_BVH_SETUP",
             "bvh: big primitives (exact)": "PH_BIG", "bvh: descent (node visits)": "PH_DESCENT", "bvh: leaf exact tests": "PH_LEAF", "bvh: loop control / park": "PH_BVH_POST",
             "shade: hit record": "PH_HITREC", "shade: |d| normalise": "PH_DNORM", "shade: rand-in-unit-sphere": "PH_SAMPLER", "shade: material record + directions": "PH_DIRS",
             "shade: texture": "PH_TEXTURE", "shade: store / rest": "PH_STORE", "shade: sphere uv": "PH_UV",
             "bvh: grid entry / next piece of the walk": "PH_GRID", "media: chords, draws, log": "PH_MEDIA"}
    weight, on = {}, False
    for l in open(table):
        if l.startswith("== "):
            on = tag in l
        m = re.match(r"\[phases\] (.+?)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+(\d+)\s*$", l)
        if on and m and m.group(1).strip() in names:
            weight[names[m.group(1).strip()]] = int(m.group(6))
    lines, on = [], False
    for l in open(asm):
        if l.startswith(pat):
            on = True
        if on:
            lines.append(l.strip())
            if "s_endpgm" in l:
                break
    per, cur = collections.defaultdict(collections.Counter), collections.Counter()
    for t in lines:
        m = re.search(r"; PHASE_(END|BEGIN) (\w+)", t)
        if m:
            if m.group(1) == "END":
                per[m.group(2)].update(cur)
            cur = collections.Counter()
            continue
        if not t or t[0] in ";./" or t.endswith(":"):
            continue
        op = t.split()[0]
        if not op.startswith("v_"):
            continue
        for name, prefixes, price_ in FAMILIES:
            if op.startswith(prefixes):
                cur[name] += 1
                break
        else:
            cur["(named)" if op.startswith(NAMED) else "v_perm / v_bfi / v_mbcnt / v_nop / other"] += 1
    tot = collections.Counter()
    for ph, c in per.items():
        w = weight.get(ph, 0)
        for k, v in c.items():
            tot[k] += v * w
    other = sum(v for k, v in tot.items() if k != "(named)")
    print("\nweighted by the stamps of %s (%s): un-named share of the executed vector instructions %.0f %% (PMC: 43 %% at C3)" % (tag, os.path.basename(table), 100.0 * other / max(1, other + tot["(named)"])))
    cyc = 0.0
    for k, v in tot.most_common():
        if k == "(named)":
            continue
        print("  %-90s %5.1f %%  x %.2f cycles" % (k, 100.0 * v / other, price.get(k, 4.4)))
        cyc += v * price.get(k, 4.4)
    print("weighted price of the un-named class: %.2f cycles per instruction (bench.py's single price: 3.9)" % (cyc / max(1, other)))
