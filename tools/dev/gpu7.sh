mkdir -p gpurun_out
(
timeout -k 10 300 python - <<'PY'
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, oracle_bind
from rimphony_amd import api, workload
ctx = api.Context(0)
L = oracle_bind.load("det")
def same(a, b): return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
for cfg, n, mask in (("cfg2_powerlaw_8", 64, 0xC0), ("cfg3_thermal_8", 64, 0xC0), ("cfg5_pitchykappa_8", 32, 0xC0), ("cfg4_pitchypl_8", 32, 0xFF), ("cfg2_powerlaw_8", 48, 0x80)):
    kind, _, s, th, params = workload.make_batch(cfg, n)
    got, st = ctx.compute_batch(kind, s, th, params, mask, want_status=True)
    ref = oracle_bind.batch(L, kind, s, th, params, mask, nthreads=16)
    sel = [k for k in range(8) if mask >> k & 1]
    ok = same(got[:, sel], ref[:, sel])
    print(cfg, n, hex(mask), "bit-equal:", ok.all(), "mismatches", (~ok).sum(), ctx.last_work(), ctx.last_tail(), flush=True)
    if not ok.all():
        for r, c in np.argwhere(~ok)[:6]: print("   row", r, "slot", sel[c], got[r, sel[c]], ref[r, sel[c]])
ctx.close()
PY
) > gpurun_out/g7.log 2>&1
echo "smoke exit $?" >> gpurun_out/g7.log
(timeout -k 10 600 python tools/ab_multi.py cfg2_powerlaw_8 16384 0xC0 2 1000000 -,RIMPHONY_SYM_SOLO=1 - variants/hw3.so &&
 timeout -k 10 300 python tools/ab_multi.py cfg3_thermal_8 16384 0xC0 1 0 -,RIMPHONY_SYM_SOLO=1 - &&
 timeout -k 10 600 python tools/ab_multi.py cfg5_pitchykappa_8 8192 0xC0 1 0 -,RIMPHONY_SYM_SOLO=1 - ) >> gpurun_out/g7.log 2>&1
echo "ab exit $?" >> gpurun_out/g7.log
cat gpurun_out/g7.log
