#!/usr/bin/env python3
"""Golden vectors from the LITERAL flavour of the oracle (oracle/liboracle_libm.so: glibc libm, unfused
polynomials, GSL's sequential GK31 summation order -- the closest thing to the Rust/GSL binary's arithmetic that can
be built here), for bench.py's `parity` object and tests/test_gpu_parity.py: the HIP path is bit-identical to the
DETERMINISTIC flavour; against this one it shows the distribution BASELINE.json's "max rel-err vs Rust/GSL ref"
asks about (rounding-level differences amplified by the reference's noise-driven control flow).

Writes tests/golden/literal_<config>.npz: start, n, mask, out [n][8] (NaN where not selected or failed).
CPU only; run in the build container:  python tools/make_literal_fixtures.py [threads] [config ...]

--controls (round 3): instead of `out`, ADD two more tables to each file -- the noise floor of the reference's own
arithmetic.  out_rev: the literal flavour with the GK31 terms added in the opposite order (oracle `make controls`,
liboracle_libm_rev.so); out_fma: the literal flavour compiled with -ffp-contract=fast (liboracle_libm_fma.so).  Both are
equally legitimate evaluations of the reference's formulas; the distance literal <-> control is what "agreement with
the Rust/GSL binary" can mean at best, and bench.py prints it next to the HIP <-> literal distance (`parity.control`).

--faraday-only (round 4): recompute slots rho_Q / rho_V of `out`, `out_rev` and `out_fma` of the eight-coefficient files
and splice them in (the Symphony slots do not go through the functions that changed).  Round 4 replaced the literal
flavour's Gamma function -- exp(lgamma(w)) / prod, 15 to 27 ulp off at the four arguments the quasi-resonant elements
need -- by the C library's long double Gamma (oracle/rimo_heyvaerts.c rimo_gamma_real); the previous Faraday columns
are kept in the files as `out_r3_gamma_explgamma` for the record."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_bind
from rimphony_amd import workload

CASES = [("cfg2_powerlaw_jI_aI", 0, 8192), ("cfg2_powerlaw_8", 1000000, 2048), ("cfg3_thermal_8", 0, 2048),
         ("cfg4_pitchypl_8", 0, 2048), ("cfg5_pitchykappa_8", 0, 2048)]
argv = [a for a in sys.argv[1:] if a not in ("--controls", "--faraday-only")]
controls = "--controls" in sys.argv
faraday_only = "--faraday-only" in sys.argv
threads = int(argv[0]) if argv else 8
only = argv[1:] or None
if faraday_only:
    import ctypes, subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all", "controls"], check=True, stdout=subprocess.DEVNULL)
    for cfg, start, n in CASES:
        if (only and cfg not in only) or cfg == "cfg2_powerlaw_jI_aI":
            continue
        path = os.path.join(ROOT, "tests", "golden", "literal_%s.npz" % cfg)
        z = dict(np.load(path))
        kind, mask, s, th, params = workload.make_batch(cfg, n, start=start)
        if "out_r3_gamma_explgamma" not in z:
            z["out_r3_gamma_explgamma"] = z["out"][:, 6:8].copy()
        for key, lib in (("out", "liboracle_libm.so"), ("out_rev", "liboracle_libm_rev.so"), ("out_fma", "liboracle_libm_fma.so")):
            L = ctypes.CDLL(os.path.join(ROOT, "oracle", lib))
            L.rimo_batch.restype = ctypes.c_int
            L.rimo_batch.argtypes = oracle_bind.load("libm").rimo_batch.argtypes
            L.rimo_build_flavour.restype = ctypes.c_char_p
            t0 = time.time()
            far = oracle_bind.batch(L, kind, s, th, params, 0xC0, nthreads=threads)
            z[key] = z[key].copy()
            z[key][:, 6:8] = far[:, 6:8]
            print("%s %s (%s): Faraday slots of %d rows in %.0f s" % (cfg, key, L.rimo_build_flavour().decode(), n, time.time() - t0), flush=True)
        np.savez_compressed(path, **z)
    sys.exit(0)
if controls:
    import ctypes, subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "controls"], check=True, stdout=subprocess.DEVNULL)
    for cfg, start, n in CASES:
        if only and cfg not in only:
            continue
        path = os.path.join(ROOT, "tests", "golden", "literal_%s.npz" % cfg)
        z = dict(np.load(path))
        kind, mask, s, th, params = workload.make_batch(cfg, n, start=start)
        for key, lib in (("out_rev", "liboracle_libm_rev.so"), ("out_fma", "liboracle_libm_fma.so")):
            L = ctypes.CDLL(os.path.join(ROOT, "oracle", lib))
            L.rimo_batch.restype = ctypes.c_int
            L.rimo_batch.argtypes = oracle_bind.load("libm").rimo_batch.argtypes
            L.rimo_build_flavour.restype = ctypes.c_char_p
            t0 = time.time()
            z[key] = oracle_bind.batch(L, kind, s, th, params, mask, nthreads=threads)
            print("%s %s (%s): %d rows in %.0f s" % (cfg, key, L.rimo_build_flavour().decode(), n, time.time() - t0), flush=True)
        np.savez_compressed(path, **z)
    sys.exit(0)
L = oracle_bind.load("libm")
assert b"libm" in L.rimo_build_flavour() or "libm" in str(L.rimo_build_flavour())
for cfg, start, n in CASES:
    if only and cfg not in only:
        continue
    kind, mask, s, th, params = workload.make_batch(cfg, n, start=start)
    t0 = time.time()
    out = oracle_bind.batch(L, kind, s, th, params, mask, nthreads=threads)
    path = os.path.join(ROOT, "tests", "golden", "literal_%s.npz" % cfg)
    np.savez_compressed(path, start=start, n=n, mask=mask, out=out)
    print("%s: %d rows in %.0f s -> %s" % (cfg, n, time.time() - t0, path), flush=True)
