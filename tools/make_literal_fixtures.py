#!/usr/bin/env python3
"""Golden vectors from the LITERAL flavour of the oracle (oracle/liboracle_libm.so: glibc libm, unfused
polynomials, GSL's sequential GK31 summation order -- the closest thing to the Rust/GSL binary's arithmetic that can
be built here), for bench.py's `parity` object and tests/test_gpu_parity.py: the HIP path is bit-identical to the
DETERMINISTIC flavour; against this one it shows the distribution BASELINE.json's "max rel-err vs Rust/GSL ref"
asks about (rounding-level differences amplified by the reference's noise-driven control flow).

Writes tests/golden/literal_<config>.npz: start, n, mask, out [n][8] (NaN where not selected or failed).
CPU only; run in the build container:  python tools/make_literal_fixtures.py [threads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_bind
from rimphony_amd import workload

CASES = [("cfg2_powerlaw_jI_aI", 0, 8192), ("cfg2_powerlaw_8", 1000000, 2048), ("cfg3_thermal_8", 0, 2048),
         ("cfg4_pitchypl_8", 0, 2048), ("cfg5_pitchykappa_8", 0, 2048)]
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
only = sys.argv[2:] or None
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
