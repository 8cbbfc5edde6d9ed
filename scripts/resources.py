#!/usr/bin/env python3
"""VGPR / SGPR / scratch / occupancy of every kernel instantiation of librtmi.so (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scripts/resources.py [extra hipcc flags...]  ->  table on stdout (profiles/roundN_resources.txt keeps a copy)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "raytrace_clj_amd", "csrc", "rtmi.hip")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
       "-I" + os.path.join(ROOT, "include"), "-Wno-unused-function", "-mllvm", "-disable-machine-licm", "-Rpass-analysis=kernel-resource-usage",
       "-o", "/dev/null", src] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (?:Function Name: (\S+)|\s*(\w[\w \[\]/]*): (\S+))", line)
    if not m:
        continue
    if m.group(1):
        cur = {"name": m.group(1)}
        rows.append(cur)
    elif cur is not None:
        cur[m.group(2).strip()] = m.group(3)
def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n
print("%-70s %5s %5s %7s %4s %5s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS"))
for r in rows:
    name = demangle(r["name"])
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    print("%-70s %5s %5s %7s %4s %5s" % (name[:70], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
