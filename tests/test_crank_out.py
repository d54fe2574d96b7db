"""The crank-out TSV format (examples/crank-out-pitchypl.rs:139-194, crank-out-pitchykappa.rs:165-216)."""
import math
import os

import numpy as np
import pytest

from rimphony_amd import crank_out


def test_rust_exponent_format():
    # Rust `{:.16e}`: no '+', no zero padding of the exponent
    assert crank_out.rust_e16(1.0) == "1.0000000000000000e0"
    assert crank_out.rust_e16(1234.5) == "1.2345000000000000e3"
    assert crank_out.rust_e16(-2.5e-7) == "-2.4999999999999999e-7"   # exact decimal expansion, correctly rounded
    assert crank_out.rust_e16(6.02e23) == "6.0200000000000000e23"
    assert crank_out.rust_e16(float("nan")) == "NaN"
    assert crank_out.rust_e16(float("inf")) == "inf"
    assert float(crank_out.rust_e16(0.1 + 0.2)) == 0.1 + 0.2     # 17 significant digits round-trip


def test_driver_writes_reference_format(tmp_path, oracle):
    """Run the pitchy-pl driver with the per-block compute replaced by the oracle (CPU)."""
    import oracle_bind
    out = tmp_path / "train.txt"

    def compute(kind, s, th, params):
        return oracle_bind.batch(oracle, kind, s, th, params, 0xFF, nthreads=8)

    n = crank_out.run("pitchypl", str(out), [1.0, 100.0, 0.2, 1.4, 2.0, 3.0, 0.0, 1.0], count=5, block=3, seed=1,
                      compute=compute)
    assert n == 5
    lines = out.read_text().strip().split("\n")
    assert lines[0] == crank_out.HEADER_PITCHYPL
    assert len(lines) == 6
    for ln in lines[1:]:
        f = ln.split("\t")
        assert len(f) == 13
        vals = [float(x) if x != "NaN" else math.nan for x in f]
        assert 1.0 <= vals[0] <= 100.0 and 0.2 <= vals[1] <= 1.4 and 2.0 <= vals[2] <= 3.0 and 0.0 <= vals[3] <= 1.0
        assert all(math.isfinite(v) for v in vals[5:11])          # six Symphony coefficients
        assert "e+" not in ln and "e-0" not in ln
    assert not os.path.exists(str(out) + ".pending")
    # second start appends and re-writes the header, as the reference does
    crank_out.run("pitchypl", str(out), [1.0, 100.0, 0.2, 1.4, 2.0, 3.0, 0.0, 1.0], count=1, block=3, seed=2, compute=compute)
    lines = out.read_text().strip().split("\n")
    assert len(lines) == 8 and lines[6] == crank_out.HEADER_PITCHYPL


def _redraw(kind_name, ranges, count, block, seed):
    """The parameter columns crank_out.run draws for these arguments (same generator, same order)."""
    rng = np.random.default_rng(seed)
    logs = [True, False, False, False] if kind_name == "pitchypl" else [True, False, False, True, False]
    samplers = [crank_out.Sampler(lg, ranges[2 * i], ranges[2 * i + 1], rng) for i, lg in enumerate(logs)]
    blocks, written = [], 0
    while written < count:
        n = min(block, count - written)
        blocks.append([smp.get(n) for smp in samplers])
        written += n
    return [np.concatenate([b[j] for b in blocks]) for j in range(len(logs))]


@pytest.mark.gpu
@pytest.mark.parametrize("kind_name", ["pitchypl", "pitchykappa"])
def test_driver_on_gpu_every_value_is_the_oracles(tmp_path, gpu_ctx, oracle, kind_name):
    """The TSV the GPU driver writes, value by value: the parameter columns are the seeded draws, the eight result
    columns are the ORACLE's bits for those parameters (printed with 17 significant digits, which round-trips a
    double), the time column is positive and sums to the kernels' time (examples/crank-out-pitchypl.rs:157-195,
    crank-out-pitchykappa.rs:184-217).  Also with two contexts sharing the block (--gpus path): identical rows."""
    import oracle_bind
    from rimphony_amd import api, workload
    ranges = ([1.0, 50.0, 0.3, 1.3, 2.0, 3.5, 0.0, 2.0] if kind_name == "pitchypl"
              else [1.0, 50.0, 0.3, 1.3, 2.0, 4.0, 3.0, 10.0, 0.0, 2.0])
    out = tmp_path / "k.txt"
    crank_out.run(kind_name, str(out), ranges, count=10, block=4, seed=3, compute=crank_out.gpu_compute([gpu_ctx]))
    lines = out.read_text().strip().split("\n")
    header = crank_out.HEADER_PITCHYPL if kind_name == "pitchypl" else crank_out.HEADER_PITCHYKAPPA
    assert lines[0] == header and len(lines) == 11
    npar = 4 if kind_name == "pitchypl" else 5
    rows = np.array([[float(x) if x != "NaN" else np.nan for x in ln.split("\t")] for ln in lines[1:]])
    assert rows.shape == (10, npar + 9)
    cols = _redraw(kind_name, ranges, 10, 4, 3)
    for j in range(npar):
        assert (rows[:, j] == cols[j]).all()
    n = 10
    if kind_name == "pitchypl":
        kind, params = workload.PITCHY_PL, [cols[2], cols[3], np.ones(n), 1e12 * np.ones(n), 1e10 * np.ones(n)]
    else:
        kind, params = workload.PITCHY_KAPPA, [cols[2], cols[3], cols[4], 1e10 * np.ones(n)]
    ref = oracle_bind.batch(oracle, kind, cols[0], cols[1], params, 0xFF, nthreads=16)
    got = rows[:, npar + 1:]
    same = (got.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), np.argwhere(~same)
    assert (rows[:, npar] > 0).all()
    # the same run sharded over two contexts (second one on the same device): identical parameter and result columns
    c2 = api.Context(0)
    try:
        out2 = tmp_path / "k2.txt"
        crank_out.run(kind_name, str(out2), ranges, count=10, block=4, seed=3, compute=crank_out.gpu_compute([gpu_ctx, c2]))
        lines2 = out2.read_text().strip().split("\n")
        for a, b in zip(lines[1:], lines2[1:]):
            fa, fb = a.split("\t"), b.split("\t")
            assert fa[:npar] == fb[:npar] and fa[npar + 1:] == fb[npar + 1:]
    finally:
        c2.close()


def test_row_times_apportion_the_kernel_times():
    work = np.array([[10, 0, 0, 0, 0, 0, 5, 0], [30, 0, 0, 0, 0, 0, 15, 0]], dtype=np.int64)
    ms = crank_out.row_times_ms(work, 8.0, 2.0)
    assert np.allclose(ms, [2.0 + 0.5, 6.0 + 1.5]) and abs(ms.sum() - 10.0) < 1e-12


def test_demo_powerlaw_and_all8_text():
    """Text layout of the two remaining reference drivers (examples/demo-powerlaw.rs:127-175,
    examples/all-pitchykappa-cgs.rs:101-132) with a stand-in compute (no GPU here)."""
    from rimphony_amd import drivers
    seen = {}

    def fake(kind, s, th, params):
        seen["n"] = len(s)
        seen["p"] = params[0].copy()
        return np.tile(np.arange(1., 9.) * 1e-20, (len(s), 1))

    lines = drivers.demo_powerlaw_lines(fake)
    assert lines[0].split("\t")[:4] == ["s(lin)", "theta(lin)", "p(lin)", "d(meta)"] and len(lines) == 65
    assert seen["n"] == 64 and seen["p"][0] == 3. and abs(seen["p"][-1] - 2.5) < 1e-15
    first = lines[1].split("\t")
    assert first[0] == "1.0000000000000000e2" and first[1] == "5.0000000000000000e-1" and first[7] == "9.9999999999999995e-21"
    last = lines[-1].split("\t")
    assert last[0] == "9.0000000000000000e1" and last[3] == "3.0000000000000000e10" and len(last) == 15

    class Calc:
        def compute_all_cgs(self, nu, b, n_e, theta):
            return np.arange(1., 9.) * 1.5

    out = drivers.all_pitchykappa_cgs_lines(lambda dist: Calc(), 1e9, 100., 1., 0.7, 4., 10., 1.)
    assert out[0] == "    j_I: 1.500000000000000000e0" and out[7].startswith("  rho_V: 1.2") and len(out) == 8


@pytest.mark.gpu
def test_demo_and_all8_on_gpu(gpu_ctx):
    """The demo's 64 rows and the 8 cgs lines equal the per-point calculator interface (bitwise: the results do
    not depend on batch composition) and carry no NaN."""
    from rimphony_amd import api, drivers
    lines = drivers.demo_powerlaw_lines(lambda kind, s, th, params: gpu_ctx.compute_batch(kind, s, th, params, api.SLOTS_ALL))
    assert len(lines) == 65
    rows = np.array([[float(c) for c in ln.split("\t")] for ln in lines[1:]])
    assert np.isfinite(rows).all()
    calc = api.PowerLawDistribution(rows[5, 2]).gamma_limits(1., 1e12, 1e10).full_calculation(gpu_ctx)
    one = calc.compute_all_dimensionless(rows[5, 0], rows[5, 1])
    assert [drivers.rust_e16(x) for x in one] == lines[6].split("\t")[7:]
    out = drivers.all_pitchykappa_cgs_lines(lambda dist: dist.full_calculation(gpu_ctx), 1e9, 100., 1., 0.7, 4., 10., 1.)
    vals = [float(ln.split(": ")[1]) for ln in out]
    assert len(vals) == 8 and all(math.isfinite(v) for v in vals) and vals[0] > 0 and vals[1] > 0


def test_rust_e_format():
    """Rust's `{:e}` (one-powerlaw-normalized.rs:43-44 prints with it)."""
    from rimphony_amd.drivers import rust_e
    assert [rust_e(x) for x in (0.0, 1.0, 2.64399749412774e-21, 1e9, 123456.789, 0.1, -1.5e-300)] == \
        ["0e0", "1e0", "2.64399749412774e-21", "1e9", "1.23456789e5", "1e-1", "-1.5e-300"]
    assert rust_e(float("nan")) == "NaN" and rust_e(float("inf")) == "inf"


@pytest.mark.gpu
def test_one_point_normalized_drivers_on_gpu(gpu_ctx, oracle):
    """examples/one-pitchypl-normalized.rs and one-powerlaw-normalized.rs: the points their author hard-coded
    (a small-angle pitchy rho_Q, a high-s power-law alpha_I), through the drivers and against the oracle's bits."""
    import oracle_bind
    from rimphony_amd import drivers
    c = drivers.ONE_PITCHYPL_NORMALIZED
    line = drivers.one_pitchypl_normalized_lines(lambda dist: dist.full_calculation(gpu_ctx))[0]
    ref = oracle_bind.batch(oracle, 2, [c["s"]], [c["theta"]], [[c["p"]], [c["k"]], [1.], [1e12], [1e10]], 0x40, nthreads=1)
    assert line == drivers.rust_e18(ref[0, 6]) and math.isfinite(ref[0, 6])
    c = drivers.ONE_POWERLAW_NORMALIZED
    lines = drivers.one_powerlaw_normalized_lines(lambda dist: dist.full_calculation(gpu_ctx))
    ref = oracle_bind.batch(oracle, 0, [c["s"]], [c["theta"]], [[c["p"]], [1.], [1e12], [1e10]], 0x02, nthreads=1)
    assert lines[0].startswith("Inner Symphony: ") and lines[1].startswith("Outer Symphony: 0e0   Us: ")
    us = float(lines[1].split("Us: ")[1])
    # compute_cgs: alpha * n_e / nu with nu / nu_c = S by the choice of B (B is rounded, so s is S to an ulp or two)
    assert abs(us / (ref[0, 1] / 1e9) - 1.) < 1e-9 and us > 0
