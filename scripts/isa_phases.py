#!/usr/bin/env python3
"""Static instruction mix of the default trace kernel per phase: build with -DRTMI_MARKERS (an assembler comment at every phase boundary, no code),
then count, between consecutive markers in program order, the instructions by issue class and price them with profiles/valu_prices.json-style
costs.  Program order is not control flow (loops and skipped branches), so this is a map of WHERE the instructions are, to be weighed with
the per-phase stamp counts of the diagnostic (stamps) build.
usage: python scripts/isa_phases.py [kernel-asm.s] [mangled-kernel-prefix]"""
import re, subprocess, sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/mk/rtmi-hip-amdgcn-amd-amdhsa-gfx950.s"
pat = sys.argv[2] if len(sys.argv) > 2 else "_ZN12_GLOBAL__N_112trace_kernelIdLb0ELi4ELb0ELb0ELb1E"
lines, on = [], False
for l in open(path):
    if l.startswith(pat):
        on = True
    if on:
        lines.append(l.rstrip())
        if "s_endpgm" in l:
            break
HALF = ("v_fma_mix", "v_pk_", "v_max3", "v_min3", "v_med3", "v_alignbit", "v_bfe", "v_and_or", "v_mad_", "v_mul_lo", "v_mul_hi", "v_cmp", "v_cvt", "v_lshrrev_b64", "v_lshlrev_b64",
        "v_add3", "v_lshl_add", "v_lshl_or", "v_readlane", "v_readfirstlane", "v_writelane", "v_bitop3", "v_perm", "v_xad", "v_add_lshl", "v_div_", "v_ldexp", "v_frexp", "v_rndne", "v_fract",
        "v_trunc_f64", "v_floor_f64", "v_cndmask_b32_e64", "v_bfi", "v_sad", "v_mbcnt", "v_or3", "v_xor3", "v_sub_co", "v_add_co", "v_addc", "v_subb")
def price(op):
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): return 16.4
    if op.startswith(("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp", "v_log", "v_sin", "v_cos", "v_rcp_iflag")): return 8.25
    if "_f64" in op: return 4.7
    if op.startswith(HALF): return 4.5
    return 2.8
phase, stats = "(prologue)", collections.OrderedDict()
for l in lines:
    m = re.search(r"; PHASE_(END|BEGIN) (\w+)", l)
    t = l.strip()
    if m:
        if m.group(1) == "END":
            stats.setdefault(m.group(2), collections.Counter())
            # instructions since the previous marker belong to the phase that ends here
            for k, v in cur.items():
                stats[m.group(2)][k] += v
            cur = collections.Counter()
        else:
            stats.setdefault("(before " + m.group(2) + ")", collections.Counter()).update(cur)
            cur = collections.Counter()
        continue
    if "cur" not in globals():
        cur = collections.Counter()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cur["valu"] += 1; cur["valu_cycles"] += price(op); cur["f64"] += "_f64" in op
        cur["mov/cnd"] += op.startswith(("v_mov", "v_cndmask", "v_accvgpr"))
    elif op.startswith("s_"):
        cur["salu"] += 1; cur["branch"] += op.startswith(("s_cbranch", "s_branch"))
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        cur["vmem"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
stats.setdefault("(epilogue)", collections.Counter()).update(cur)
print("%-34s %6s %9s %6s %8s %6s %6s %5s %5s" % ("phase (instructions that END at its markers)", "VALU", "cycles", "f64", "mov/cnd", "SALU", "branch", "VMEM", "LDS"))
tot = collections.Counter()
for k, c in stats.items():
    tot.update(c)
    print("%-34s %6d %9.0f %6d %8d %6d %6d %5d %5d" % (k, c["valu"], c["valu_cycles"], c["f64"], c["mov/cnd"], c["salu"], c["branch"], c["vmem"], c["lds"]))
print("%-34s %6d %9.0f %6d %8d %6d %6d %5d %5d" % ("total", tot["valu"], tot["valu_cycles"], tot["f64"], tot["mov/cnd"], tot["salu"], tot["branch"], tot["vmem"], tot["lds"]))
