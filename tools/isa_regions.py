#!/usr/bin/env python3
"""Static instruction count of one kernel by REGION: in a -DRIM_ISA_MARKS build (hipcc -S) the region timers of the
diagnostic build (RIM_PROF_T / RIM_PROF_ADD, detmath.h) become "; REGION_BEGIN name" / "; REGION_END name" comments in
the listing; this walks the kernel's listing in layout order and charges every instruction to the innermost open region.
Layout order is not execution order and blocks are not weighted by how often they run: a map of where the instructions
ARE, to be read next to the region timers (tools/region_profile_group.py), which say where the TIME goes.
usage: isa_regions.py FILE.s KERNEL_SUBSTRING"""
import collections, re, sys

path, kern = sys.argv[1], sys.argv[2]
inside = False
stack = []
tally = collections.defaultdict(collections.Counter)
FP64 = re.compile(r"^\s+v_(fma|fmac|add|mul|max|min|rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|ldexp|frexp_mant|trunc|floor|rndne|fract|cmp\w*)_f64")
for l in open(path):
    if l.startswith("_Z") and ":" in l and kern in l.split(":")[0]:
        inside = True
        continue
    if inside and l.startswith(".Lfunc_end"):
        break
    if not inside:
        continue
    m = re.search(r"; REGION_(BEGIN|END) (\w+)", l)
    if m:
        if m.group(1) == "BEGIN":
            stack.append(m.group(2))
        elif m.group(2) in stack:
            while stack and stack.pop() != m.group(2):
                pass
        continue
    key = stack[-1] if stack else "(outside)"
    t = tally[key]
    if re.match(r"^\s+v_", l):
        t["valu"] += 1
        if FP64.match(l):
            t["fp64"] += 1
        if re.match(r"^\s+v_read(first)?lane", l):
            t["readlane"] += 1
        if re.match(r"^\s+v_cndmask", l):
            t["cndmask"] += 1
    elif re.match(r"^\s+ds_", l):
        t["lds"] += 1
    elif re.match(r"^\s+s_(?!waitcnt|nop|cbranch|branch|barrier|sleep|setprio|endpgm)", l):
        t["salu"] += 1
    elif re.match(r"^\s+(scratch_|global_|flat_|buffer_)", l):
        t["vmem"] += 1
tot = collections.Counter()
for t in tally.values():
    tot.update(t)
print("kernel %s: %d VALU (%d fp64), %d SALU, %d LDS, %d VMEM" % (kern, tot["valu"], tot["fp64"], tot["salu"], tot["lds"], tot["vmem"]))
for k, t in sorted(tally.items(), key=lambda kv: -kv[1]["valu"]):
    print("%-14s VALU %5d  fp64 %5d  readlane %4d  cndmask %4d  SALU %5d  LDS %4d  VMEM %4d" %
          (k, t["valu"], t["fp64"], t["readlane"], t["cndmask"], t["salu"], t["lds"], t["vmem"]))
