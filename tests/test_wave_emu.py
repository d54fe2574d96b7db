"""The wave-level code (wave_qag.h, symphony_wave.h, heyvaerts_wave.h) executed on the CPU by a
64-thread wavefront emulator (tests/support/wave_emu.h) and compared bit for bit with the oracle.
This is how the state machines were debugged without a GPU; it is slow (barrier-based collectives),
so it only runs with RIMPHONY_SLOW=1."""
import ctypes
import os
import subprocess

import pytest

import oracle_bind
from rimphony_amd import workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.environ.get("RIMPHONY_SLOW"), reason="set RIMPHONY_SLOW=1 (minutes per case)")


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(ROOT, "tests", "support", "wave_emu_driver.cpp")
    so = os.path.join(ROOT, "tests", "support", "wave_emu.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-math-errno", "-mfma", "-msse4.1",
                    "-pthread", "-I" + os.path.join(ROOT, "tests", "support"), "-shared", src, "-o", so], check=True)
    E = ctypes.CDLL(so)
    E.emu_symphony.restype = ctypes.c_int
    E.emu_symphony.argtypes = [ctypes.c_int] * 3 + [ctypes.c_double] * 2 + [ctypes.POINTER(ctypes.c_double), ctypes.c_double,
                                                                            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int),
                                                                            ctypes.POINTER(ctypes.c_ulonglong)]
    gargs = [ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
             ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ulonglong)]
    for fn in (E.emu_symphony_group, E.emu_heyvaerts_group):
        fn.restype = ctypes.c_int
        fn.argtypes = gargs
    return E


def _run(E, L, kind, par, coeff, stokes, s, th):
    d, st = oracle_bind.mkdist(L, kind, par)
    c = oracle_bind.Counters()
    ref = L.rimo_compute_dimensionless(d, coeff, stokes, s, th, ctypes.byref(c))
    res, stt, w = ctypes.c_double(), ctypes.c_int(), (ctypes.c_ulonglong * 3)()
    pa = (ctypes.c_double * 5)(*(list(par) + [0.] * 5)[:5])
    uniform = E.emu_symphony(kind, coeff, stokes, s, th, pa, d.norm, ctypes.byref(res), ctypes.byref(stt), w)
    assert uniform == 1                       # every lane ended with the same value
    assert res.value == ref or (res.value != res.value and ref != ref)
    assert w[0] == c.integrand_evals          # same number of integrand samples as the oracle


def test_symphony_coefficient_in_emulator(emu, oracle):
    kind, mask, s, th, params = workload.make_batch("cfg2_powerlaw_jI_aI", 16)
    _run(emu, oracle, kind, [p[8] for p in params], 0, 0, s[8], th[8])


def test_faraday_coefficient_in_emulator(emu, oracle):
    _run(emu, oracle, 1, [10.], 2, 1, 4e4, 0.4)


def test_faraday_jy_branch_in_emulator(emu, oracle):
    """Points whose quasi-resonant part reaches g >= 10 (sigma0 < 3): the J/Y branch of the elements."""
    _run(emu, oracle, 1, [0.3], 2, 1, 1.5, 0.6)
    _run(emu, oracle, 0, [2.5, 1., 1e12, 1e10], 2, 2, 2.0, 0.9)


def _run_group(fn, L, kind, par, slots_list, s, th):
    """A whole group (symphony_group.h / heyvaerts_group.h) on the emulated wave: every member's value carries the bits of
    the oracle's per-coefficient run, and the members' sample counts add up to the oracle's."""
    d, st = oracle_bind.mkdist(L, kind, par)
    slots = 0
    for i, sl in enumerate(slots_list):
        slots |= sl << (4 * i)
    vals, stats, w = (ctypes.c_double * 4)(), (ctypes.c_int * 4)(), (ctypes.c_ulonglong * 48)()
    pa = (ctypes.c_double * 5)(*(list(par) + [0.] * 5)[:5])
    assert fn(kind, slots, len(slots_list), s, th, pa, d.norm, vals, stats, w) == 1
    total = 0
    for i, sl in enumerate(slots_list):
        c = oracle_bind.Counters()
        coeff, stokes = (2, 1 if sl == 6 else 2) if sl >= 6 else (sl & 1, sl >> 1)
        ref = L.rimo_compute_dimensionless(d, coeff, stokes, s, th, ctypes.byref(c))
        total += c.integrand_evals
        assert vals[i] == ref or (vals[i] != vals[i] and ref != ref), (sl, vals[i], ref)
    assert w[0] == total
    assert w[3] >= w[1]                       # passes the members would have run alone >= passes executed


def test_symphony_groups_in_emulator(emu, oracle):
    kind, mask, s, th, params = workload.make_batch("cfg2_powerlaw_jI_aI", 16)
    par = [p[8] for p in params]
    _run_group(emu.emu_symphony_group, oracle, kind, par, [0, 1, 2, 3], s[8], th[8])      # j_I, alpha_I, j_Q, alpha_Q
    _run_group(emu.emu_symphony_group, oracle, kind, par, [4, 5], s[8], th[8])            # j_V, alpha_V (two lobes)
    _run_group(emu.emu_symphony_group, oracle, kind, par, [1, 2], s[8], th[8])            # a partial mask


def test_faraday_pair_in_emulator(emu, oracle):
    _run_group(emu.emu_heyvaerts_group, oracle, 1, [10.], [6, 7], 4e4, 0.4)
