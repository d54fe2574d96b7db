#!/usr/bin/env python3
"""Static instruction MIX of one kernel by region (a -DRIM_ISA_MARKS listing, see tools/isa_regions.py): per region the
vector instructions by class -- fp64 arithmetic, moves (literal materialisation, copies), selects, compares, lane traffic,
integer / conversion -- so that the non-arithmetic part of a pass can be itemised.
Static counts in layout order; weight them with the execution frequencies of tools/hit_profile.py.
usage: isa_region_mix.py FILE.s KERNEL_SUBSTRING"""
import collections, re, sys
path, kern = sys.argv[1], sys.argv[2]
CLASSES = [("fp64", r"v_(fma|fmac|add|mul|max|min|rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|ldexp|frexp_mant|trunc|floor|rndne|fract)_f64"),
           ("cmp64", r"v_cmpx?_\w+_f64"), ("mov", r"v_mov_b32|v_mov_b64|v_accvgpr|v_pk_mov"), ("cndmask", r"v_cndmask"),
           ("cmp32", r"v_cmpx?_"), ("lane", r"v_readlane|v_readfirstlane|v_writelane|v_permlane|v_mov_b32_dpp|v_\w+_dpp"),
           ("int/cvt", r"v_(add|sub|lshl|lshr|ashr|and|or|xor|bfe|bfi|mul_lo|mul_hi|mad|add3|lshl_add|lshl_or|and_or|or3|cvt|frexp_exp|ldexp|not|min|max|subrev|alignbit|perm)")]
CRE = [(n, re.compile(r"^\s+(" + p + ")")) for n, p in CLASSES]
inside = False
stack = []
tally = collections.defaultdict(collections.Counter)
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
    if not re.match(r"^\s+v_", l):
        if re.match(r"^\s+scratch_", l):
            tally[stack[-1] if stack else "(outside)"]["scratch"] += 1
        elif re.match(r"^\s+ds_", l):
            tally[stack[-1] if stack else "(outside)"]["lds"] += 1
        continue
    t = tally[stack[-1] if stack else "(outside)"]
    t["valu"] += 1
    for n, r in CRE:
        if r.match(l):
            t[n] += 1
            break
    else:
        t["other"] += 1
cols = ["valu", "fp64", "cmp64", "mov", "cndmask", "cmp32", "lane", "int/cvt", "other", "scratch", "lds"]
print("%-12s " % "region" + " ".join("%8s" % c for c in cols))
tot = collections.Counter()
for k, t in sorted(tally.items(), key=lambda kv: -kv[1]["valu"]):
    print("%-12s " % k + " ".join("%8d" % t[c] for c in cols))
    tot.update(t)
print("%-12s " % "total" + " ".join("%8d" % tot[c] for c in cols))
