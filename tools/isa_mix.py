#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc -S output, grouped by the loop each basic block belongs to.
usage: isa_mix.py FILE.s KERNEL_SUBSTRING [LOOP_HEADER ...]
Prints, per loop header (LLVM's "in Loop: Header=BBx_y Depth=d" block comments), the count of instructions by class.
Static counts: blocks are not weighted by how often they run."""
import collections
import re
import sys

CLASSES = [
    ("fma64", r"v_fma_f64|v_fmac_f64"), ("add64", r"v_add_f64"), ("mul64", r"v_mul_f64"),
    ("trans64", r"v_rcp_f64|v_rsq_f64|v_sqrt_f64"), ("div_fix", r"v_div_scale_f64|v_div_fmas_f64|v_div_fixup_f64"),
    ("cmp", r"v_cmp|v_cmpx"), ("cndmask", r"v_cndmask"), ("mov", r"v_mov_b32|v_mov_b64|v_accvgpr"),
    ("readlane", r"v_readlane|v_readfirstlane"), ("writelane", r"v_writelane"),
    ("cvt/ldexp/frexp", r"v_cvt|v_ldexp|v_frexp|v_rndne|v_floor|v_trunc|v_fract"),
    ("minmax64", r"v_max_f64|v_min_f64"), ("other_valu", r"v_"),
    ("salu", r"s_(?!waitcnt|nop|cbranch|branch|barrier|sleep|setprio|endpgm|load|store|buffer|dcache|memtime|memrealtime)"),
    ("sbranch", r"s_cbranch|s_branch"), ("swait/nop", r"s_waitcnt|s_nop"), ("smem", r"s_load|s_store|s_buffer"),
    ("lds", r"ds_"), ("vmem", r"global_|flat_|buffer_|scratch_"),
]
CRE = [(n, re.compile(r"^\s+(" + p + ")")) for n, p in CLASSES]


def main():
    path, kern = sys.argv[1], sys.argv[2]
    want = set(sys.argv[3:])
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kern in l and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    cur = "top"
    depth = {}
    tally = collections.defaultdict(collections.Counter)
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
        if m and not mm:
            mh = re.search(r"Loop Header: Depth=(\d+)", l)
            cur = "top"
        if mm:
            cur = mm.group(1)
            depth[cur] = int(mm.group(2))
            continue
        mh = re.search(r"=>\s+This (?:Inner )?Loop Header: Depth=(\d+)", l)
        if mh:
            # the label line precedes; find it
            pass
        if l.startswith(";") or l.strip().startswith(";") or not l.startswith("\t"):
            continue
        for n, r in CRE:
            if r.match(l):
                tally[cur][n] += 1
                break
    keys = sorted(tally, key=lambda k: -sum(tally[k].values()))
    for k in keys:
        if want and k not in want:
            continue
        c = tally[k]
        valu = sum(v for n, v in c.items() if n in {x[0] for x in CLASSES[:13]})
        print("%-12s depth %s  total %5d  VALU %5d :: %s" % (k, depth.get(k, "-"), sum(c.values()), valu,
              "  ".join("%s %d" % (n, c[n]) for n, _ in CLASSES if c[n])))


main()
