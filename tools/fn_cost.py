#!/usr/bin/env python3
"""Static vector-instruction cost of the leaf functions (tools/fn_cost.hip): compiles to gfx950 ISA and counts, per
kernel, VALU instructions, the fp64 arithmetic among them, and SALU; the `empty` kernel's count is the harness."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "fn_cost.s")
subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-mllvm", "-disable-machine-licm",
                "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "rimphony_amd", "csrc"),
                os.path.join(ROOT, "tools", "fn_cost.hip"), "-o", out] + sys.argv[1:], check=True)
cur, tally = None, {}
F64 = re.compile(r"v_(fma|fmac|add|mul|max|min|rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|ldexp|frexp_mant|trunc|floor|rndne|fract|cmp\w*|cmpx\w*)_f64")
for l in open(out):
    m = re.match(r"^(k_\w+):", l)
    if m:
        cur = m.group(1); tally[cur] = [0, 0, 0, 0]; continue
    if l.startswith(".Lfunc_end"):
        cur = None
    if cur is None or not l.startswith("\t") or l.lstrip().startswith((";", ".")):
        continue
    op = l.split()[0]
    if op.startswith("v_"):
        tally[cur][0] += 1
        if F64.match(op): tally[cur][1] += 1
    elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_endpgm")):
        tally[cur][2] += 1
    elif op.startswith(("scratch_",)):
        tally[cur][3] += 1
base = tally["k_empty"]
print("%-22s %6s %6s %6s %7s   (harness subtracted: %d VALU)" % ("function", "VALU", "fp64", "SALU", "scratch", base[0]))
for k, v in tally.items():
    print("%-22s %6d %6d %6d %7d" % (k[2:], v[0] - base[0], v[1] - base[1], v[2] - base[2], v[3]))
