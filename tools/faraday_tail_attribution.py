#!/usr/bin/env python3
"""Which lock-step reformulation owns the Faraday tail?  

The HIP kernels are bit-identical to the oracle's DETERMINISTIC flavour; against the LITERAL flavour (glibc libm,
unfused, GSL's summation order: tests/golden/literal_*.npz) rho_Q / rho_V show a tail beyond 1e-6 that the literal
flavour's own contracted control build does not.  This tool runs the ATTRIBUTION build of the oracle
(oracle/liboracle_attr.so, `make -C oracle attr`: the deterministic flavour with every reformulation of DESIGN.md
section 3 item 5 switchable back to its literal form at run time, oracle/rimo_math.h RIMO_ATTR_*) on the rows of the
four eight-coefficient literal tables, Faraday slots only, and tabulates the distance from the literal vectors

  * with NO form reverted  (= the kernels; must reproduce the HIP-vs-literal numbers of the bench line),
  * with ALL forms reverted (must BE the literal vectors, bit for bit: the switch list is complete),
  * with ONE form reverted  (what removing that form alone would buy),
  * with ALL BUT ONE reverted (what that form alone costs on an otherwise literal evaluation),
  * and with any further masks given on the command line (--mask 0x.. or names joined by +).

CPU only.  python tools/faraday_tail_attribution.py [--threads N] [--rows N] [--tables cfg,..] [--mask NAME+NAME ...] [--out FILE]
Results are cached per (table, mask, rows) in gpurun_out/attr_cache.json so that a run can be resumed."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_bind
from rimphony_amd import workload

BITS = [("third_powers", 1 << 0), ("rgamma_near", 1 << 1), ("series_pairs", 1 << 2), ("nr_upowers", 1 << 3),
        ("pow15_limits", 1 << 4), ("own_exp_log_pow", 1 << 5), ("powexp_one_exp", 1 << 6), ("gk_tree_order", 1 << 7),
        ("rescale_xsqrtx", 1 << 8), ("own_sincos", 1 << 9), ("dinv_products", 1 << 12), ("g_one_root", 1 << 13), ("jy_joint", 1 << 14)]
ALL = (1 << 15) - 1
FARADAY_ALL = sum(b for _, b in BITS)
TABLES = [("cfg2_powerlaw_8", 1000000), ("cfg3_thermal_8", 0), ("cfg4_pitchypl_8", 0), ("cfg5_pitchykappa_8", 0)]


def parse_mask(text):
    if text.startswith("0x"):
        return int(text, 16)
    names = dict(BITS)
    return sum(names[n] for n in text.split("+"))


def mask_name(m):
    if m == 0:
        return "none reverted (= kernels)"
    if m == ALL or m == FARADAY_ALL:
        return "all reverted (= literal)"
    on = [n for n, b in BITS if m & b]
    off = [n for n, b in BITS if not m & b]
    if len(off) <= 2 and len(on) > 2:
        return "all reverted but " + " + ".join(off)
    return "reverted: " + " + ".join(on)


def stats(got, lit):
    res = {}
    for name, cols in (("rho_Q", [0]), ("rho_V", [1]), ("both", [0, 1])):
        g, l = got[:, cols], lit[:, cols]
        gn, ln = np.isnan(g), np.isnan(l)
        ok = ~gn & ~ln
        rel = np.abs(g[ok] - l[ok]) / np.abs(l[ok])
        res[name] = dict(median=float(np.median(rel)), p99=float(np.percentile(rel, 99)), max=float(rel.max()),
                         over=int((rel > 1e-6).sum()), n=int(ok.sum()), nan_here=int((gn & ~ln).sum()),
                         nan_there=int((~gn & ln).sum()), bit_equal=int(((g.view(np.uint64) == l.view(np.uint64)) | (gn & ln)).all()))
    return res


def main():
    argv = sys.argv[1:]
    threads, rows, extra, out_path = 7, 2048, [], os.path.join(ROOT, "profiles", "r4_faraday_tail_attribution.txt")
    only_extra = False
    only_tables = None
    while argv:
        a = argv.pop(0)
        if a == "--threads": threads = int(argv.pop(0))
        elif a == "--rows": rows = int(argv.pop(0))
        elif a == "--mask": extra.append(parse_mask(argv.pop(0)))
        elif a == "--out": out_path = argv.pop(0)
        elif a == "--only-extra": only_extra = True
        elif a == "--tables": only_tables = argv.pop(0).split(",")
        else: raise SystemExit("unknown argument " + a)
    L = oracle_bind.load("attr")
    L.rimo_set_attr_mask.restype = None
    L.rimo_set_attr_mask.argtypes = [ctypes.c_uint]
    assert b"attribution" in L.rimo_build_flavour()
    masks = [0, ALL] if not only_extra else []
    if not only_extra:
        masks += [b for _, b in BITS] + [ALL ^ b for _, b in BITS]
    masks += extra
    cache_path = os.path.join(ROOT, "gpurun_out", "attr_cache.json")
    os.makedirs(os.path.dirname(cache_path), exist_ok=True)
    cache = json.load(open(cache_path)) if os.path.exists(cache_path) else {}
    for cfg, start in TABLES:
        if only_tables and cfg not in only_tables:
            continue
        kind, mask8, s, th, params = workload.make_batch(cfg, rows, start=start)
        z = np.load(os.path.join(ROOT, "tests", "golden", "literal_%s.npz" % cfg))
        assert int(z["start"]) == start and int(z["n"]) >= rows
        lit = z["out"][:rows, 6:8]
        ctl = z["out_fma"][:rows, 6:8] if "out_fma" in z else None
        if ctl is not None:
            cache["%s|contracted-control|%d" % (cfg, rows)] = stats(ctl, lit)
        for m in masks:
            key = "%s|%#x|%d" % (cfg, m, rows)
            if key in cache:
                continue
            L.rimo_set_attr_mask(m)
            t0 = time.time()
            got = oracle_bind.batch(L, kind, s, th, params, 0xC0, nthreads=threads)[:, 6:8]
            cache[key] = stats(got, lit)
            cache[key]["seconds"] = round(time.time() - t0, 1)
            print("%s %-60s %5.0f s  p99 %.2e max %.2e >1e-6: %d" % (cfg, mask_name(m), time.time() - t0,
                  cache[key]["both"]["p99"], cache[key]["both"]["max"], cache[key]["both"]["over"]), flush=True)
            json.dump(cache, open(cache_path, "w"), indent=0)
    L.rimo_set_attr_mask(0)
    # report
    lines = ["Faraday tail attribution (tools/faraday_tail_attribution.py): oracle attribution build vs tests/golden/literal_*.npz,",
             "%d rows per table, slots rho_Q and rho_V; per cell: p99 / max / coefficients beyond 1e-6 / NaN here : there" % rows,
             "'none reverted' is the arithmetic of the HIP kernels; 'all reverted' must be bit-equal to the literal vectors.", ""]
    all_masks = []
    for k in cache:
        c, m, r = k.split("|")
        if int(r) == rows and m not in all_masks:
            all_masks.append(m)
    def order(m):
        if m == "contracted-control": return (-1, 0)
        v = int(m, 16)
        return (bin(v & FARADAY_ALL).count("1"), v)
    all_masks.sort(key=order)
    for slot in ("both", "rho_Q", "rho_V"):
        lines.append("== %s ==" % slot)
        lines.append("%-62s" % "form(s) evaluated literally" + "".join("%-40s" % c for c, _ in TABLES))
        for m in all_masks:
            name = "literal flavour, -ffp-contract=fast (control)" if m == "contracted-control" else mask_name(int(m, 16))
            row = "%-62s" % name
            for cfg, _ in TABLES:
                st = cache.get("%s|%s|%d" % (cfg, m, rows))
                if st is None:
                    row += "%-40s" % "-"
                    continue
                q = st[slot]
                cell = "%.1e / %.1e / %3d / %d:%d%s" % (q["p99"], q["max"], q["over"], q["nan_here"], q["nan_there"], " =" if q["bit_equal"] else "")
                row += "%-40s" % cell
            lines.append(row)
        lines.append("")
    text = "\n".join(lines)
    open(out_path, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
