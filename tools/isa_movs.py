#!/usr/bin/env python3
"""Classify the v_mov / v_cndmask / v_readlane instructions of the blocks of one kernel at loop depth >= D.
usage: isa_movs.py FILE.s KERNEL_SUBSTRING DEPTH [pattern]"""
import collections, re, sys
path, kern, mind = sys.argv[1], sys.argv[2], int(sys.argv[3])
pat = re.compile(sys.argv[4] if len(sys.argv) > 4 else r"v_mov_b32|v_mov_b64")
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kern in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
depth = 0
cats = collections.Counter()
ex = collections.defaultdict(list)
prev = ""
hdr_depth = {}
for l in lines[start:end]:
    mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
    mh = re.search(r"Loop Header: Depth=(\d+)", l)
    if re.match(r"^\.LBB\d+_\d+:", l):
        if mm: depth = int(mm.group(2))
        elif "Parent Loop" in l or "=>" in l: depth = depth   # header line; depth comes on the following comment lines
        else: depth = 0 if not mh else int(mh.group(1))
    if mh: depth = int(mh.group(1))
    if mm and not l.startswith(".LBB"): depth = int(mm.group(2))
    if not l.startswith("\t") or l.strip().startswith(";"):
        continue
    if depth >= mind and pat.search(l):
        m = re.match(r"\s+(\S+)\s+(\S+),\s*(.*)", l)
        op, src = m.group(1), m.group(3).strip()
        first = src.split(",")[0].strip()
        if re.match(r"s\[|s\d|vcc|exec|ttmp|src_", first): c = "sgpr"
        elif first.startswith("v"): c = "vgpr"
        else: c = "const"
        cats[(op, c)] += 1
        if len(ex[(op, c)]) < 6: ex[(op, c)].append((prev.strip(), l.strip()))
    prev = l
for k, v in cats.most_common():
    print(v, k)
    for e in ex[k]: print("      ", e)
