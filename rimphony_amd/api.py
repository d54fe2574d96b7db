"""Host-side mirror of rimphony's public API on top of the HIP C ABI.

Reference surface being mirrored (same names, argument meaning and error
behaviour; see src/lib.rs of pkgw/rimphony):

  Stokes, Coefficient                               lib.rs:74-107
  SynchrotronCalculator.compute_dimensionless       lib.rs:155-156
                       .compute_cgs                 lib.rs:163-173
                       .compute_all_dimensionless   lib.rs:178-191
                       .compute_all_cgs             lib.rs:196-209
  PowerLawDistribution(p).gamma_limits(..).full_calculation()          power_law.rs:71-111
  ThermalJuettnerDistribution(T).full_calculation()                    thermal_juettner.rs:45-72
  PitchyPowerLawDistribution(p, k).gamma_limits(..).full_calculation() pitchy_pl.rs:73-115
  PitchyKappaDistribution(kappa, width, k).gamma_cutoff(..)...         pitchy_kappa.rs:70-125

plus the batched compute() the north star adds: `compute_batch`.  PyTorch is
used only as plumbing (device buffers, the current HIP stream).  Everything is
evaluated by librimphony_hip.so on the GPU; there is no CPU path here.
"""
import ctypes
import enum
import math

import numpy as np
import torch

from . import capi

# lib.rs:55-67
PI = math.pi
TWO_PI = 2.0 * math.pi
MASS_ELECTRON = 9.1093826e-28
SPEED_LIGHT = 2.99792458e10
ELECTRON_CHARGE = 4.80320680e-10


class Stokes(enum.IntEnum):
    I = 0
    Q = 1
    V = 2


class Coefficient(enum.IntEnum):
    Emission = 0
    Absorption = 1
    Faraday = 2


POWER_LAW, THERMAL_JUETTNER, PITCHY_PL, PITCHY_KAPPA = 0, 1, 2, 3
NPARAMS = {POWER_LAW: 4, THERMAL_JUETTNER: 1, PITCHY_PL: 5, PITCHY_KAPPA: 4}

# slot order of compute_all_dimensionless (lib.rs:176-177)
SLOTS = [
    (Coefficient.Emission, Stokes.I), (Coefficient.Absorption, Stokes.I),
    (Coefficient.Emission, Stokes.Q), (Coefficient.Absorption, Stokes.Q),
    (Coefficient.Emission, Stokes.V), (Coefficient.Absorption, Stokes.V),
    (Coefficient.Faraday, Stokes.Q), (Coefficient.Faraday, Stokes.V),
]
SLOTS_ALL = 0xFF
SLOTS_SYMPHONY = 0x3F
# `precision` of the batch entry points (include/rimphony_hip.h)
PRECISION_F64 = 0               # the reference's arithmetic; bit-identical to the oracle
PRECISION_F32_INTEGRAND = 1     # refused by the library (RIMPHONY_ENOTSUP): slower and lossier than F64, include/rimphony_hip.h


def slot_of(coeff, stokes):
    return SLOTS.index((Coefficient(coeff), Stokes(stokes)))


class Context:
    """Owns a rimphony_ctx bound to one GPU."""

    def __init__(self, device=0):
        self.lib = capi.load()
        if not torch.cuda.is_available():
            raise capi.RimphonyError("no HIP device visible: rimphony_amd has no CPU fallback")
        self.device = int(device)
        h = ctypes.c_void_p()
        capi.check(self.lib.rimphony_ctx_create(self.device, ctypes.byref(h)), "rimphony_ctx_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.rimphony_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------
    def _dev(self):
        return torch.device("cuda", self.device)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self._dev()).cuda_stream)

    def _as_dev(self, x):
        t = torch.as_tensor(x, dtype=torch.float64)
        return t.to(self._dev()).contiguous()

    # -- batched compute() -------------------------------------------------------
    def _check_input(self, name, t, n):
        """The C ABI takes raw device pointers and cannot see what they point to: refuse anything but a contiguous
        float64 tensor of n elements on this context's GPU (a strided view, a float32 tensor, a tensor on another
        device or a length-1 'broadcast' parameter would be read as garbage or out of bounds)."""
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s: expected a torch.Tensor, got %s" % (name, type(t).__name__))
        if not t.is_cuda or t.device != self._dev():
            raise ValueError("%s: tensor is on %s, this context computes on %s" % (name, t.device, self._dev()))
        if t.dtype != torch.float64:
            raise TypeError("%s: dtype %s, expected float64" % (name, t.dtype))
        if t.numel() != n:
            raise ValueError("%s: %d elements, expected %d (one per parameter point; scalars are not broadcast)"
                             % (name, t.numel(), n))
        if not t.is_contiguous():
            raise ValueError("%s: not contiguous (pass t.contiguous())" % name)

    def compute_batch_device(self, kind, s, theta, params, coeff_mask=SLOTS_ALL, want_status=False, want_work=False,
                             precision=PRECISION_F64):
        """s, theta: CUDA float64 tensors [n]; params: list of CUDA float64 tensors [n].
        Returns (out [n, 8] CUDA tensor, status [n, 8] int32 CUDA tensor or None) -- and, with want_work, a third
        element: [n, 8] int64 integrand samples spent per coefficient.  Asynchronous on the current stream."""
        if kind not in NPARAMS:
            raise ValueError("unknown distribution kind %r" % (kind,))
        if len(params) != NPARAMS[kind]:
            raise ValueError("distribution kind %d takes %d parameter arrays, got %d" % (kind, NPARAMS[kind], len(params)))
        if not isinstance(s, torch.Tensor):
            raise TypeError("s: expected a torch.Tensor, got %s" % type(s).__name__)
        n = s.numel()
        self._check_input("s", s, n)
        self._check_input("theta", theta, n)
        for k, p in enumerate(params):
            self._check_input("params[%d]" % k, p, n)
        out = torch.empty((n, 8), dtype=torch.float64, device=self._dev())
        status = torch.empty((n, 8), dtype=torch.int32, device=self._dev()) if want_status else None
        work = torch.empty((n, 8), dtype=torch.int64, device=self._dev()) if want_work else None
        pp = (ctypes.c_void_p * len(params))(*[ctypes.c_void_p(p.data_ptr()) for p in params])
        capi.check(self.lib.rimphony_batch_compute_device_ex(
            self.handle, kind, n, ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(theta.data_ptr()), pp,
            coeff_mask, int(precision), ctypes.c_void_p(out.data_ptr()),
            ctypes.c_void_p(status.data_ptr()) if want_status else None,
            ctypes.c_void_p(work.data_ptr()) if want_work else None, self._stream()),
            "rimphony_batch_compute_device_ex")
        if want_work:
            return out, status, work
        return out, status

    def compute_batch(self, kind, s, theta, params, coeff_mask=SLOTS_ALL, want_status=False, want_work=False,
                      precision=PRECISION_F64):
        """Host arrays in, numpy arrays out (synchronous)."""
        ds, dth = self._as_dev(s), self._as_dev(theta)
        dp = [self._as_dev(p) for p in params]
        res = self.compute_batch_device(kind, ds, dth, dp, coeff_mask, want_status, want_work, precision)
        torch.cuda.synchronize(self._dev())
        ret = [res[0].cpu().numpy()]
        if want_status:
            ret.append(res[1].cpu().numpy())
        if want_work:
            ret.append(res[2].cpu().numpy())
        return ret[0] if len(ret) == 1 else tuple(ret)

    def shared_mode(self):
        """False when this context owns its GPU (full persistent grids, cooperative tail); True when another context
        or process had the device first (rimphony_ctx_create in include/rimphony_hip.h)."""
        return bool(self.lib.rimphony_ctx_shared_mode(self.handle))

    def status_histogram(self, status):
        """status: the [n, 8] int32 CUDA tensor of a batch call -> numpy [8 slots, 8] counts: columns 0..6 = rows
        with RIMPHONY_ST_* bit b set, column 7 = rows with status 0."""
        if not (isinstance(status, torch.Tensor) and status.is_cuda and status.dtype == torch.int32 and status.is_contiguous()):
            raise TypeError("status: expected a contiguous int32 CUDA tensor")
        hist = (ctypes.c_uint64 * 64)()
        capi.check(self.lib.rimphony_status_histogram_device(self.handle, status.numel() // 8, ctypes.c_void_p(status.data_ptr()),
                                                              hist, self._stream()), "rimphony_status_histogram_device")
        return np.array(list(hist), dtype=np.int64).reshape(8, 8)

    DETMATH_OPS = {"exp": 0, "log": 1, "log10": 2, "pow": 3, "sqrt": 4, "log10_region": 5, "lgamma": 6, "sin": 7, "cos": 8,
                   "div_by": 9, "cbrt": 10, "rgamma": 11, "third_powers": 12,
                   "rqrt4": 13, "powexp": 14}

    def detmath_batch(self, op, x, y=None):
        """Unit seam: a leaf function of detmath.h evaluated on the device over host arrays."""
        dx = self._as_dev(x)
        dy = self._as_dev(y) if y is not None else None
        out = torch.empty_like(dx)
        capi.check(self.lib.rimphony_detmath_batch_device(
            self.handle, self.DETMATH_OPS[op], dx.numel(), ctypes.c_void_p(dx.data_ptr()),
            ctypes.c_void_p(dy.data_ptr()) if dy is not None else None, ctypes.c_void_p(out.data_ptr()), self._stream()),
            "rimphony_detmath_batch_device")
        torch.cuda.synchronize(self._dev())
        return out.cpu().numpy()

    def highfreq_batch(self, kind, s, theta, params):
        """High-frequency closed forms (power law / thermal only): host arrays in, [n, 2] = {rho_Q, rho_V} out."""
        ds, dth = self._as_dev(s), self._as_dev(theta)
        dp = [self._as_dev(p) for p in params]
        n = ds.numel()
        out = torch.empty((n, 2), dtype=torch.float64, device=self._dev())
        pp = (ctypes.c_void_p * len(dp))(*[ctypes.c_void_p(p.data_ptr()) for p in dp])
        capi.check(self.lib.rimphony_highfreq_batch_device(
            self.handle, kind, n, ctypes.c_void_p(ds.data_ptr()), ctypes.c_void_p(dth.data_ptr()), pp,
            ctypes.c_void_p(out.data_ptr()), self._stream()), "rimphony_highfreq_batch_device")
        torch.cuda.synchronize(self._dev())
        return out.cpu().numpy()

    def last_work(self):
        w = capi.Work()
        capi.check(self.lib.rimphony_last_work(self.handle, ctypes.byref(w)), "rimphony_last_work")
        return {"samples": int(w.samples), "passes": int(w.passes), "inner_qags": int(w.inner_qags),
                "faraday_samples": int(w.faraday_samples), "faraday_passes": int(w.faraday_passes),
                "faraday_inner_qags": int(w.faraday_inner_qags)}

    def last_tail(self):
        """Heaviest task of the last call per kernel (its chain of batches bounds the launch's tail) and the group
        kernel's sharing counters: include/rimphony_hip.h, rimphony_last_tail."""
        o = (ctypes.c_uint64 * 8)()
        capi.check(self.lib.rimphony_last_tail(self.handle, o), "rimphony_last_tail")
        return {"symphony_heaviest_batches": int(o[0]), "symphony_heaviest_row": int(o[1]),
                "faraday_heaviest_batches": int(o[2]), "faraday_heaviest_row": int(o[3]),
                "member_passes": int(o[4]), "stash_filed": int(o[5]),
                "faraday_member_passes": int(o[6]), "faraday_stash_filed": int(o[7])}

    def last_symphony_ms(self):
        ms = ctypes.c_float()
        capi.check(self.lib.rimphony_last_symphony_ms(self.handle, ctypes.byref(ms)), "rimphony_last_symphony_ms")
        return float(ms.value)

    def last_faraday_ms(self):
        ms = ctypes.c_float()
        capi.check(self.lib.rimphony_last_faraday_ms(self.handle, ctypes.byref(ms)), "rimphony_last_faraday_ms")
        return float(ms.value)

    def debug_counters(self):
        arr = (ctypes.c_uint64 * 32)()
        capi.check(self.lib.rimphony_debug_counters(self.handle, arr), "rimphony_debug_counters")
        return [int(v) for v in arr]

    def heartbeat(self, task=0):
        """Diagnostics: returns a ctypes pointer to 16 host-mapped uint64 words (see rimphony_hip.h)."""
        p = ctypes.POINTER(ctypes.c_uint64)()
        capi.check(self.lib.rimphony_debug_heartbeat(self.handle, int(task), ctypes.byref(p)),
                   "rimphony_debug_heartbeat")
        return p

    def norm_batch(self, kind, params):
        dp = [self._as_dev(p) for p in params]
        n = dp[0].numel()
        out = torch.empty(n, dtype=torch.float64, device=self._dev())
        pp = (ctypes.c_void_p * len(dp))(*[ctypes.c_void_p(p.data_ptr()) for p in dp])
        capi.check(self.lib.rimphony_batch_norm_device(self.handle, kind, n, pp, ctypes.c_void_p(out.data_ptr()),
                                                       self._stream()), "rimphony_batch_norm_device")
        torch.cuda.synchronize(self._dev())
        return out.cpu().numpy()

    # -- unit seams ----------------------------------------------------------------
    def bessel_batch(self, n, x):
        dn, dx = self._as_dev(n), self._as_dev(x)
        j = torch.empty_like(dn)
        dj = torch.empty_like(dn)
        capi.check(self.lib.rimphony_bessel_batch_device(
            self.handle, dn.numel(), ctypes.c_void_p(dn.data_ptr()), ctypes.c_void_p(dx.data_ptr()),
            ctypes.c_void_p(j.data_ptr()), ctypes.c_void_p(dj.data_ptr()), self._stream()),
            "rimphony_bessel_batch_device")
        torch.cuda.synchronize(self._dev())
        return j.cpu().numpy(), dj.cpu().numpy()

    def gamma_integrand_batch(self, kind, params, coeff, stokes, s, theta, n, gamma):
        dn, dg = self._as_dev(n), self._as_dev(gamma)
        out = torch.empty_like(dn)
        par = (ctypes.c_double * len(params))(*params)
        capi.check(self.lib.rimphony_gamma_integrand_batch_device(
            self.handle, kind, par, int(coeff), int(stokes), s, theta, dn.numel(),
            ctypes.c_void_p(dn.data_ptr()), ctypes.c_void_p(dg.data_ptr()), ctypes.c_void_p(out.data_ptr()),
            self._stream()), "rimphony_gamma_integrand_batch_device")
        torch.cuda.synchronize(self._dev())
        return out.cpu().numpy()

    def gamma_integral_batch(self, kind, params, coeff, stokes, negative_lobe, s, theta, n):
        dn = self._as_dev(n)
        out = torch.empty_like(dn)
        par = (ctypes.c_double * len(params))(*params)
        capi.check(self.lib.rimphony_gamma_integral_batch_device(
            self.handle, kind, par, int(coeff), int(stokes), int(negative_lobe), s, theta, dn.numel(),
            ctypes.c_void_p(dn.data_ptr()), ctypes.c_void_p(out.data_ptr()), self._stream()),
            "rimphony_gamma_integral_batch_device")
        torch.cuda.synchronize(self._dev())
        return out.cpu().numpy()

    def n_integral_batch(self, kind, params, coeff, stokes, negative_lobe, s, theta, n_lo, n_hi):
        """diagnostic_symphony_n_integral over arrays of [n_lo, n_hi] ranges of one parameter point."""
        dlo, dhi = self._as_dev(n_lo), self._as_dev(n_hi)
        out = torch.empty_like(dlo)
        par = (ctypes.c_double * len(params))(*params)
        capi.check(self.lib.rimphony_n_integral_batch_device(
            self.handle, kind, par, int(coeff), int(stokes), int(negative_lobe), s, theta, dlo.numel(),
            ctypes.c_void_p(dlo.data_ptr()), ctypes.c_void_p(dhi.data_ptr()), ctypes.c_void_p(out.data_ptr()), self._stream()),
            "rimphony_n_integral_batch_device")
        torch.cuda.synchronize(self._dev())
        return out.cpu().numpy()

    def deriv_probe_batch(self, kind, params, coeff, stokes, negative_lobe, s, theta, n_start):
        """gsl::deriv_central as n_integration drives it (symphony.rs:238-240): d(gamma_integral)/dn at each n_start."""
        dn = self._as_dev(n_start)
        out = torch.empty_like(dn)
        par = (ctypes.c_double * len(params))(*params)
        capi.check(self.lib.rimphony_deriv_probe_batch_device(
            self.handle, kind, par, int(coeff), int(stokes), int(negative_lobe), s, theta, dn.numel(),
            ctypes.c_void_p(dn.data_ptr()), ctypes.c_void_p(out.data_ptr()), self._stream()),
            "rimphony_deriv_probe_batch_device")
        torch.cuda.synchronize(self._dev())
        return out.cpu().numpy()

    def gamma_contribution_batch(self, kind, params, coeff, stokes, s, theta, gamma):
        """diagnostic_symphony_gamma_contribution over an array of gammas of one parameter point."""
        dg = self._as_dev(gamma)
        out = torch.empty_like(dg)
        par = (ctypes.c_double * len(params))(*params)
        capi.check(self.lib.rimphony_gamma_contribution_batch_device(
            self.handle, kind, par, int(coeff), int(stokes), s, theta, dg.numel(),
            ctypes.c_void_p(dg.data_ptr()), ctypes.c_void_p(out.data_ptr()), self._stream()),
            "rimphony_gamma_contribution_batch_device")
        torch.cuda.synchronize(self._dev())
        return out.cpu().numpy()

    def calc_f_batch(self, kind, params, gamma, cos_xi, norm=None):
        """DistributionFunction::calc_f and calc_f_derivatives (lib.rs:111-146) of one distribution over arrays:
        returns (f, dfdg, dfdcx).  norm=None uses the distribution's own normalisation."""
        dg, dc = self._as_dev(gamma), self._as_dev(cos_xi)
        f, a, b = torch.empty_like(dg), torch.empty_like(dg), torch.empty_like(dg)
        par = (ctypes.c_double * len(params))(*params)
        capi.check(self.lib.rimphony_calc_f_batch_device(
            self.handle, kind, par, float("nan") if norm is None else float(norm), dg.numel(),
            ctypes.c_void_p(dg.data_ptr()), ctypes.c_void_p(dc.data_ptr()), ctypes.c_void_p(f.data_ptr()),
            ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), self._stream()),
            "rimphony_calc_f_batch_device")
        torch.cuda.synchronize(self._dev())
        return f.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy()

    def hey_element_batch(self, kind, params, stokes, s, theta, qr, fixed, v):
        """Heyvaerts inner integrand (h/f element, quasi-resonant if qr) of one parameter point over arrays."""
        df, dv = self._as_dev(fixed), self._as_dev(v)
        out = torch.empty_like(df)
        par = (ctypes.c_double * len(params))(*params)
        capi.check(self.lib.rimphony_hey_element_batch_device(
            self.handle, kind, par, int(stokes), s, theta, int(qr), df.numel(), ctypes.c_void_p(df.data_ptr()),
            ctypes.c_void_p(dv.data_ptr()), ctypes.c_void_p(out.data_ptr()), self._stream()), "rimphony_hey_element_batch_device")
        torch.cuda.synchronize(self._dev())
        return out.cpu().numpy()

    def hey_outer_batch(self, kind, params, stokes, s, theta, qr, u):
        """Heyvaerts outer integrand (one inner integral per abscissa) of one parameter point over an array."""
        du = self._as_dev(u)
        out = torch.empty_like(du)
        par = (ctypes.c_double * len(params))(*params)
        capi.check(self.lib.rimphony_hey_outer_batch_device(
            self.handle, kind, par, int(stokes), s, theta, int(qr), du.numel(), ctypes.c_void_p(du.data_ptr()),
            ctypes.c_void_p(out.data_ptr()), self._stream()), "rimphony_hey_outer_batch_device")
        torch.cuda.synchronize(self._dev())
        return out.cpu().numpy()

    def qag_selftest(self, family, p0, p1, a, b, epsabs, epsrel, limit):
        fam = torch.as_tensor(family, dtype=torch.int32).to(self._dev()).contiguous()
        d = [self._as_dev(v) for v in (p0, p1, a, b)]
        n = fam.numel()
        res = torch.empty(n, dtype=torch.float64, device=self._dev())
        err = torch.empty(n, dtype=torch.float64, device=self._dev())
        qst = torch.empty(n, dtype=torch.int32, device=self._dev())
        size = torch.empty(n, dtype=torch.int32, device=self._dev())
        capi.check(self.lib.rimphony_qag_selftest_device(
            self.handle, n, ctypes.c_void_p(fam.data_ptr()), *[ctypes.c_void_p(v.data_ptr()) for v in d],
            epsabs, epsrel, int(limit), ctypes.c_void_p(res.data_ptr()), ctypes.c_void_p(err.data_ptr()),
            ctypes.c_void_p(qst.data_ptr()), ctypes.c_void_p(size.data_ptr()), self._stream()),
            "rimphony_qag_selftest_device")
        torch.cuda.synchronize(self._dev())
        return res.cpu().numpy(), err.cpu().numpy(), qst.cpu().numpy(), size.cpu().numpy()


def compute_batch_multi(ctxs, kind, s, theta, params, coeff_mask=SLOTS_ALL, want_status=False, want_work=False):
    """The in-process multi-GPU batch (rimphony_batch_compute_multi): host arrays in, row i evaluated by
    ctxs[i mod len(ctxs)], numpy [n, 8] out (+ status, + work).  The table does not depend on len(ctxs)."""
    lib = capi.load()
    s = np.ascontiguousarray(s, dtype=np.float64)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    params = [np.ascontiguousarray(p, dtype=np.float64) for p in params]
    n = s.size
    if theta.size != n or any(p.size != n for p in params):
        raise ValueError("s, theta and every parameter array must have one element per point")
    if len(params) != NPARAMS[kind]:
        raise ValueError("distribution kind %d takes %d parameter arrays" % (kind, NPARAMS[kind]))
    dp = ctypes.POINTER(ctypes.c_double)
    out = np.empty((n, 8), dtype=np.float64)
    status = np.empty((n, 8), dtype=np.int32) if want_status else None
    work = np.empty((n, 8), dtype=np.uint64) if want_work else None
    handles = (ctypes.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
    pp = (dp * len(params))(*[p.ctypes.data_as(dp) for p in params])
    capi.check(lib.rimphony_batch_compute_multi(
        handles, len(ctxs), kind, n, s.ctypes.data_as(dp), theta.ctypes.data_as(dp), pp, coeff_mask, 0,
        out.ctypes.data_as(dp),
        status.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if want_status else None,
        work.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)) if want_work else None), "rimphony_batch_compute_multi")
    ret = [out] + ([status] if want_status else []) + ([work] if want_work else [])
    return ret[0] if len(ret) == 1 else tuple(ret)


def compute_batch_multi_device(ctxs, kind, shards, coeff_mask=SLOTS_ALL, want_status=False, synchronize=True):
    """rimphony_batch_compute_multi_device: device buffers per context.  shards[r] = (s, theta, [params...]) as torch
    tensors on ctxs[r]'s device; returns the per-context [n_r, 8] output tensors (+ status tensors)."""
    import torch
    lib = capi.load()
    n_ctx = len(ctxs)
    dp = ctypes.POINTER(ctypes.c_double)
    np_k = NPARAMS[kind]
    outs, stats, keep = [], [], []
    n_local = (ctypes.c_size_t * n_ctx)()
    d_s, d_th, d_out = (ctypes.c_void_p * n_ctx)(), (ctypes.c_void_p * n_ctx)(), (ctypes.c_void_p * n_ctx)()
    d_par = (ctypes.POINTER(ctypes.c_void_p) * n_ctx)()
    d_st = (ctypes.c_void_p * n_ctx)()
    streams = (ctypes.c_void_p * n_ctx)()
    for r, (c, (s, th, params)) in enumerate(zip(ctxs, shards)):
        if len(params) != np_k:
            raise ValueError("distribution kind %d takes %d parameter arrays" % (kind, np_k))
        n = s.numel()
        n_local[r] = n
        out = torch.empty((n, 8), dtype=torch.float64, device=s.device)
        st = torch.empty((n, 8), dtype=torch.int32, device=s.device) if want_status else None
        arr = (ctypes.c_void_p * np_k)(*[p.data_ptr() for p in params])
        keep.append(arr)
        d_s[r], d_th[r], d_out[r] = s.data_ptr(), th.data_ptr(), out.data_ptr()
        d_par[r] = ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p))
        d_st[r] = st.data_ptr() if want_status else None
        streams[r] = torch.cuda.current_stream(s.device).cuda_stream
        outs.append(out)
        stats.append(st)
    handles = (ctypes.c_void_p * n_ctx)(*[c.handle for c in ctxs])
    capi.check(lib.rimphony_batch_compute_multi_device(handles, n_ctx, kind, n_local, d_s, d_th, d_par, coeff_mask, 0, d_out,
                                                       d_st if want_status else None, None, streams, 1 if synchronize else 0),
               "rimphony_batch_compute_multi_device")
    return (outs, stats) if want_status else outs


class RcclComm:
    """An RCCL communicator made through the library's own own run-time-loaded librccl (rimphony_rccl_*): what a host that is not
    Python would use for the gather of the output table.  One per rank; rank 0 makes the 128-byte id."""

    def __init__(self, ctx, rank, world, unique_id):
        lib = capi.load()
        self.ctx, self.rank, self.world = ctx, rank, world
        self.handle = ctypes.c_void_p()
        capi.check(lib.rimphony_rccl_comm_create(ctx.handle, rank, world, unique_id, ctypes.byref(self.handle)),
                   "rimphony_rccl_comm_create")

    @staticmethod
    def available():
        return bool(capi.load().rimphony_rccl_available())

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        capi.check(capi.load().rimphony_rccl_unique_id(buf), "rimphony_rccl_unique_id")
        return buf

    def gather_table(self, shard, n_total, root=0, scratch=None):
        """shard: this rank's [m, 8] float64 device tensor (rows rank, rank + world, ... of the table); returns the
        [n_total, 8] table on the root, None elsewhere."""
        import torch
        table = torch.empty((n_total, 8), dtype=torch.float64, device=shard.device) if self.rank == root else None
        capi.check(capi.load().rimphony_rccl_gather_table(
            self.ctx.handle, self.handle, self.rank, self.world, root, n_total, ctypes.c_void_p(shard.data_ptr()),
            ctypes.c_void_p(table.data_ptr()) if table is not None else None,
            ctypes.c_void_p(scratch.data_ptr()) if scratch is not None else None,
            ctypes.c_void_p(torch.cuda.current_stream(shard.device).cuda_stream)), "rimphony_rccl_gather_table")
        return table

    def close(self):
        if self.handle:
            capi.load().rimphony_rccl_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ---------------------------------------------------------------------------------
# distribution builders + calculators, named as in the reference
# ---------------------------------------------------------------------------------

class FullSynchrotronCalculator:
    """lib.rs:230-247.  One parameter point; every call is a 1-point batch."""

    def __init__(self, kind, params, ctx=None):
        self.kind = kind
        self.params = [float(p) for p in params]
        self.ctx = ctx or default_context()

    def _run(self, s, theta, mask):
        out = self.ctx.compute_batch(self.kind, [s], [theta], [[p] for p in self.params], mask)
        return out[0]

    def compute_dimensionless(self, coeff, stokes, s, theta):
        coeff, stokes = Coefficient(coeff), Stokes(stokes)
        if coeff == Coefficient.Faraday and stokes == Stokes.I:
            return float("nan")          # lib.rs:239-240
        k = slot_of(coeff, stokes)
        return float(self._run(s, theta, 1 << k)[k])

    def compute_cgs(self, coeff, stokes, nu, b, n_e, theta):
        nu_c = ELECTRON_CHARGE * b / (TWO_PI * MASS_ELECTRON * SPEED_LIGHT)
        val = self.compute_dimensionless(coeff, stokes, nu / nu_c, theta)
        if Coefficient(coeff) == Coefficient.Emission:
            return val * n_e * nu
        return val * n_e / nu

    def compute_all_dimensionless(self, s, theta):
        return np.array(self._run(s, theta, SLOTS_ALL))

    def compute_all_cgs(self, nu, b, n_e, theta):
        nu_c = ELECTRON_CHARGE * b / (TWO_PI * MASS_ELECTRON * SPEED_LIGHT)
        v = self.compute_all_dimensionless(nu / nu_c, theta)
        scale = np.array([n_e * nu if c == Coefficient.Emission else n_e / nu for c, _ in SLOTS])
        return v * scale


class HighFrequencyApproximation:
    """SynchrotronCalculator over the closed-form high-frequency approximations
    (power_law.rs:131-170, thermal_juettner.rs:92-142): (Faraday, Q) and (Faraday, V); anything else is NaN."""

    def __init__(self, kind, params, ctx=None):
        if kind not in (POWER_LAW, THERMAL_JUETTNER):
            raise ValueError("the reference defines high-frequency approximations for power_law and thermal_juettner only")
        self.kind = kind
        self.params = [float(p) for p in params]
        self.ctx = ctx or default_context()

    def compute_dimensionless(self, coeff, stokes, s, theta):
        if int(coeff) != int(Coefficient.Faraday) or int(stokes) == int(Stokes.I):
            return float("nan")
        out = self.ctx.highfreq_batch(self.kind, np.array([s], dtype=np.float64), np.array([theta], dtype=np.float64),
                                      [np.array([p], dtype=np.float64) for p in self.params])
        return float(out[0, 0 if int(stokes) == int(Stokes.Q) else 1])


class _DistributionFunction:
    """The DistributionFunction trait (lib.rs:111-146).  `norm` mirrors the public field of the reference's structs:
    None until set, in which case calc_f uses the normalisation full_calculation would compute."""
    norm = None
    ctx = None

    def _kind_params(self):
        raise NotImplementedError

    def _calc(self, gamma, cos_xi):
        kind, params = self._kind_params()
        g, c = np.atleast_1d(np.asarray(gamma, dtype=np.float64)), np.atleast_1d(np.asarray(cos_xi, dtype=np.float64))
        g, c = np.broadcast_arrays(g, c)
        out = (self.ctx or default_context()).calc_f_batch(kind, params, np.ascontiguousarray(g), np.ascontiguousarray(c),
                                                           self.norm)
        return out if np.ndim(gamma) or np.ndim(cos_xi) else tuple(float(v[0]) for v in out)

    def calc_f(self, gamma, cos_xi):
        return self._calc(gamma, cos_xi)[0]

    def calc_f_derivatives(self, gamma, cos_xi):
        _, dfdg, dfdcx = self._calc(gamma, cos_xi)
        return dfdg, dfdcx


class PowerLawDistribution(_DistributionFunction):
    def _kind_params(self):
        return POWER_LAW, [self.p, self.gamma_min, self.gamma_max, self.gamma_cutoff]

    def __init__(self, p):
        self.p, self.gamma_min, self.gamma_max, self.gamma_cutoff = float(p), 1.0, 1e12, 1e10

    def gamma_limits(self, gamma_min, gamma_max, gamma_cutoff):
        self.gamma_min, self.gamma_max, self.gamma_cutoff = float(gamma_min), float(gamma_max), float(gamma_cutoff)
        return self

    def full_calculation(self, ctx=None):
        return FullSynchrotronCalculator(POWER_LAW, [self.p, self.gamma_min, self.gamma_max, self.gamma_cutoff], ctx)

    def high_freq_approximation(self, ctx=None):
        return HighFrequencyApproximation(POWER_LAW, [self.p, self.gamma_min, self.gamma_max, self.gamma_cutoff], ctx)


class ThermalJuettnerDistribution(_DistributionFunction):
    def _kind_params(self):
        return THERMAL_JUETTNER, [self.t]

    def __init__(self, t):
        self.t = float(t)

    def full_calculation(self, ctx=None):
        return FullSynchrotronCalculator(THERMAL_JUETTNER, [self.t], ctx)

    def high_freq_approximation(self, ctx=None):
        return HighFrequencyApproximation(THERMAL_JUETTNER, [self.t], ctx)


class PitchyPowerLawDistribution(_DistributionFunction):
    def _kind_params(self):
        return PITCHY_PL, [self.p, self.k, self.gamma_min, self.gamma_max, self.gamma_cutoff]

    def __init__(self, p, k):
        self.p, self.k = float(p), float(k)
        self.gamma_min, self.gamma_max, self.gamma_cutoff = 1.0, 1e12, 1e10

    def gamma_limits(self, gamma_min, gamma_max, gamma_cutoff):
        self.gamma_min, self.gamma_max, self.gamma_cutoff = float(gamma_min), float(gamma_max), float(gamma_cutoff)
        return self

    def full_calculation(self, ctx=None):
        return FullSynchrotronCalculator(
            PITCHY_PL, [self.p, self.k, self.gamma_min, self.gamma_max, self.gamma_cutoff], ctx)


class PitchyKappaDistribution(_DistributionFunction):
    def _kind_params(self):
        return PITCHY_KAPPA, [self.kappa, self.width, self.k, self._gamma_cutoff]

    def __init__(self, kappa, width, k):
        self.kappa, self.width, self.k, self._gamma_cutoff = float(kappa), float(width), float(k), 1e10

    def gamma_cutoff(self, gamma_cutoff):
        self._gamma_cutoff = float(gamma_cutoff)
        return self

    def full_calculation(self, ctx=None):
        return FullSynchrotronCalculator(PITCHY_KAPPA, [self.kappa, self.width, self.k, self._gamma_cutoff], ctx)
