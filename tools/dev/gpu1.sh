mkdir -p gpurun_out
timeout -k 10 300 python - > gpurun_out/g1_smoke.log 2>&1 <<'PY'
import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, oracle_bind
from rimphony_amd import api, workload
ctx = api.Context(0)
L = oracle_bind.load("det")
def same(a, b): return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
for cfg, n, mask in (("cfg2_powerlaw_jI_aI", 16, 0x3), ("cfg2_powerlaw_8", 16, 0x3f), ("cfg3_thermal_8", 16, 0x3f), ("cfg2_powerlaw_8", 64, 0x3f)):
    kind, _, s, th, params = workload.make_batch(cfg, n)
    t0 = time.time()
    got = ctx.compute_batch(kind, s, th, params, mask)
    t1 = time.time()
    ref = oracle_bind.batch(L, kind, s, th, params, mask, nthreads=16)
    sel = [k for k in range(8) if mask >> k & 1]
    ok = same(got[:, sel], ref[:, sel])
    print(cfg, n, hex(mask), "gpu %.2fs" % (t1 - t0), "bit-equal:", ok.all(), "mismatches", (~ok).sum(), flush=True)
    if not ok.all():
        bad = np.argwhere(~ok)[:6]
        for r, c in bad: print("   row", r, "slot", sel[c], got[r, sel[c]], ref[r, sel[c]])
    w = ctx.last_work() if hasattr(ctx, "last_work") else None
    print("   work", w, flush=True)
ctx.close()
PY
echo "exit $?" >> gpurun_out/g1_smoke.log
tail -30 gpurun_out/g1_smoke.log
