#!/usr/bin/env python3
"""Static vector-instruction count of one kernel by SOURCE LINE (hipcc -S -gline-tables-only output): which lines of the
sources the instructions of a kernel come from, summed over source-line ranges given on the command line.
usage: isa_lines.py FILE.s KERNEL_SUBSTRING [file.h:first-last=label ...]
Without ranges: the 60 heaviest (file, line) pairs.  Static counts: not weighted by how often a block runs."""
import collections, re, sys

path, kern = sys.argv[1], sys.argv[2]
ranges = []
for a in sys.argv[3:]:
    loc, label = a.split("=")
    f, r = loc.split(":")
    lo, hi = r.split("-")
    ranges.append((f, int(lo), int(hi), label))
files = {}
valu = collections.Counter()
fp64 = collections.Counter()
lds = collections.Counter()
inside = False
cur = None
FP64 = re.compile(r"^\s+v_(fma|fmac|add|mul|max|min|rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|ldexp|frexp_mant|trunc|floor|rndne|fract|cmp\w*)_f64")
for l in open(path):
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if l.startswith("_Z") and ":" in l and kern in l.split(":")[0]:
        inside = True
        continue
    if inside and l.startswith(".Lfunc_end"):
        break
    if not inside:
        continue
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if re.match(r"^\s+v_", l):
        valu[cur] += 1
        if FP64.match(l):
            fp64[cur] += 1
    elif re.match(r"^\s+ds_", l):
        lds[cur] += 1
print("kernel %s: %d vector instructions (%d fp64), %d LDS" % (kern, sum(valu.values()), sum(fp64.values()), sum(lds.values())))
if ranges:
    rest = sum(valu.values())
    for f, lo, hi, label in ranges:
        v = sum(c for (ff, ln), c in valu.items() if ff == f and lo <= ln <= hi)
        d = sum(c for (ff, ln), c in fp64.items() if ff == f and lo <= ln <= hi)
        s = sum(c for (ff, ln), c in lds.items() if ff == f and lo <= ln <= hi)
        rest -= v
        print("%-44s %6d VALU (%5d fp64) %5d LDS   %s:%d-%d" % (label, v, d, s, f, lo, hi))
    print("%-44s %6d VALU" % ("(everything else)", rest))
else:
    byfile = collections.Counter()
    for (f, ln), c in valu.items():
        byfile[f] += c
    for f, c in byfile.most_common():
        print("%-28s %6d" % (f, c))
    for (f, ln), c in valu.most_common(60):
        print("%-28s %5d  %5d VALU %5d fp64 %4d LDS" % (f, ln, c, fp64[(f, ln)], lds[(f, ln)]))
