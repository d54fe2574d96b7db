// rimphony-hip-sys/src/lib.rs -- raw FFI declarations for librimphony_hip.so.
//
// UNVERIFIED: the image this was written in has no Rust toolchain; this file has never been compiled.  It is a
// hand transcription of include/rimphony_hip.h (every `extern "C"` entry a binding needs; the unit seams that only
// the parity tests use are left out) and of the safe wrapper a maintainer of pkgw/rimphony would put on top
// (`HipContext`, `compute_batch`).  tests/test_host_side.py::test_rust_sys_crate_matches_header checks, without a
// Rust compiler, that every function named here is declared in the header with the same number of arguments.
//
// What each entry replaces in the reference (file:line relative to pkgw/rimphony):
//   rimphony_batch_compute*      N x (D::new(..).full_calculation(log) + compute_all_dimensionless(s, theta))
//                                src/lib.rs:178-191, src/power_law.rs:71-111 and the three other distributions
//   rimphony_batch_compute_multi the same over the GPUs of a node (row i -> context i mod n_ctx)
//   rimphony_batch_norm_device   full_calculation() alone
//   rimphony_highfreq_batch      high_freq_approximation() calculators, src/power_law.rs:117-170, thermal_juettner.rs:78-142
//   rimphony_calc_f_batch        DistributionFunction::calc_f / calc_f_derivatives, src/lib.rs:111-146
//   pkgw_bessel_j / pkgw_bessel_dj   the two symbols leung-bessel/src/lib.rs:36-42 already binds

#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_double, c_float, c_int, c_void};

#[repr(C)]
pub struct rimphony_ctx {
    _private: [u8; 0],
}

/// rimphony_work (include/rimphony_hip.h): device-counted work of the last batch call
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rimphony_work {
    pub samples: u64,
    pub passes: u64,
    pub inner_qags: u64,
    pub faraday_samples: u64,
    pub faraday_passes: u64,
    pub faraday_inner_qags: u64,
}

// distribution kinds: order of the SoA parameter arrays in the header
pub const RIMPHONY_POWER_LAW: c_int = 0; // p, gamma_min, gamma_max, gamma_cutoff
pub const RIMPHONY_THERMAL_JUETTNER: c_int = 1; // T
pub const RIMPHONY_PITCHY_PL: c_int = 2; // p, k, gamma_min, gamma_max, gamma_cutoff
pub const RIMPHONY_PITCHY_KAPPA: c_int = 3; // kappa, width, k, gamma_cutoff

// coeff_mask bits = output slots, in the order of compute_all_dimensionless (lib.rs:176-177)
pub const RIMPHONY_SLOT_J_I: u32 = 1 << 0;
pub const RIMPHONY_SLOT_ALPHA_I: u32 = 1 << 1;
pub const RIMPHONY_SLOT_J_Q: u32 = 1 << 2;
pub const RIMPHONY_SLOT_ALPHA_Q: u32 = 1 << 3;
pub const RIMPHONY_SLOT_J_V: u32 = 1 << 4;
pub const RIMPHONY_SLOT_ALPHA_V: u32 = 1 << 5;
pub const RIMPHONY_SLOT_RHO_Q: u32 = 1 << 6;
pub const RIMPHONY_SLOT_RHO_V: u32 = 1 << 7;
pub const RIMPHONY_SLOTS_ALL: u32 = 0xff;

// per-coefficient status bits
pub const RIMPHONY_ST_INNER_FAIL: i32 = 1;
pub const RIMPHONY_ST_OUTER_FAIL: i32 = 2;
pub const RIMPHONY_ST_CHUNK_CAP: i32 = 4;
pub const RIMPHONY_ST_STORE_FULL: i32 = 8;
pub const RIMPHONY_ST_NONFINITE: i32 = 16;
pub const RIMPHONY_ST_NORM_FAIL: i32 = 32;
pub const RIMPHONY_ST_NOT_COMPUTED: i32 = 64;

// error codes
pub const RIMPHONY_OK: c_int = 0;
pub const RIMPHONY_EINVAL: c_int = -1;
pub const RIMPHONY_EHIP: c_int = -2;
pub const RIMPHONY_ENOMEM: c_int = -3;
pub const RIMPHONY_ENODEVICE: c_int = -4;
pub const RIMPHONY_EBUSY: c_int = -5;
pub const RIMPHONY_ENOTSUP: c_int = -6;
pub const RIMPHONY_ERCCL: c_int = -7;

pub const RIMPHONY_PRECISION_F64: c_int = 0;
pub const RIMPHONY_PRECISION_F32_INTEGRAND: c_int = 1;

// (the `links = "rimphony_hip"` build script names the library and its directory)
extern "C" {
    pub fn rimphony_dist_nparams(dist_kind: c_int) -> c_int;
    pub fn rimphony_ctx_create(device: c_int, out: *mut *mut rimphony_ctx) -> c_int;
    pub fn rimphony_ctx_destroy(ctx: *mut rimphony_ctx);
    pub fn rimphony_ctx_shared_mode(ctx: *const rimphony_ctx) -> c_int;
    pub fn rimphony_ctx_device(ctx: *const rimphony_ctx, device: *mut c_int) -> c_int;
    pub fn rimphony_strerror(code: c_int) -> *const c_char;
    pub fn rimphony_last_error() -> *const c_char;
    pub fn rimphony_version() -> *const c_char;

    pub fn rimphony_last_work(ctx: *mut rimphony_ctx, out: *mut rimphony_work) -> c_int;
    pub fn rimphony_last_tail(ctx: *mut rimphony_ctx, out: *mut u64) -> c_int;
    pub fn rimphony_last_symphony_ms(ctx: *mut rimphony_ctx, ms: *mut c_float) -> c_int;
    pub fn rimphony_last_faraday_ms(ctx: *mut rimphony_ctx, ms: *mut c_float) -> c_int;

    /// device buffers, asynchronous on `stream` (a hipStream_t)
    pub fn rimphony_batch_compute_device(
        ctx: *mut rimphony_ctx, dist_kind: c_int, n: usize,
        d_s: *const c_double, d_theta: *const c_double, d_params: *const *const c_double,
        coeff_mask: u32, d_out: *mut c_double, d_status: *mut i32, stream: *mut c_void) -> c_int;
    /// host buffers: copies in, computes, copies out, synchronises
    pub fn rimphony_batch_compute(
        ctx: *mut rimphony_ctx, dist_kind: c_int, n: usize,
        s: *const c_double, theta: *const c_double, params: *const *const c_double,
        coeff_mask: u32, out: *mut c_double, status: *mut i32) -> c_int;
    pub fn rimphony_batch_compute_device_ex(
        ctx: *mut rimphony_ctx, dist_kind: c_int, n: usize,
        d_s: *const c_double, d_theta: *const c_double, d_params: *const *const c_double,
        coeff_mask: u32, precision: c_int,
        d_out: *mut c_double, d_status: *mut i32, d_work: *mut u64, stream: *mut c_void) -> c_int;
    pub fn rimphony_batch_compute_ex(
        ctx: *mut rimphony_ctx, dist_kind: c_int, n: usize,
        s: *const c_double, theta: *const c_double, params: *const *const c_double,
        coeff_mask: u32, precision: c_int, out: *mut c_double, status: *mut i32, work: *mut u64) -> c_int;
    /// ctxs[r] evaluates rows r, r + n_ctx, ...; the table does not depend on n_ctx
    pub fn rimphony_batch_compute_multi(
        ctxs: *const *mut rimphony_ctx, n_ctx: c_int, dist_kind: c_int, n: usize,
        s: *const c_double, theta: *const c_double, params: *const *const c_double,
        coeff_mask: u32, precision: c_int, out: *mut c_double, status: *mut i32, work: *mut u64) -> c_int;
    /// device buffers per context; asynchronous on `streams` unless `synchronize` != 0
    pub fn rimphony_batch_compute_multi_device(
        ctxs: *const *mut rimphony_ctx, n_ctx: c_int, dist_kind: c_int, n_local: *const usize,
        d_s: *const *const c_double, d_theta: *const *const c_double, d_params: *const *const *const c_double,
        coeff_mask: u32, precision: c_int, d_out: *const *mut c_double, d_status: *const *mut i32,
        d_work: *const *mut u64, streams: *const *mut c_void, synchronize: c_int) -> c_int;
    /// RCCL (dlopen'ed by the library): the gather of the interleaved table on a root rank
    pub fn rimphony_rccl_available() -> c_int;
    pub fn rimphony_rccl_unique_id(id128: *mut c_void) -> c_int;
    pub fn rimphony_rccl_comm_create(
        ctx: *mut rimphony_ctx, rank: c_int, world: c_int, id128: *const c_void, comm: *mut *mut c_void) -> c_int;
    pub fn rimphony_rccl_comm_destroy(comm: *mut c_void) -> c_int;
    pub fn rimphony_rccl_gather_table(
        ctx: *mut rimphony_ctx, comm: *mut c_void, rank: c_int, world: c_int, root: c_int, n_total: usize,
        d_shard: *const c_double, d_table: *mut c_double, d_scratch: *mut c_double, stream: *mut c_void) -> c_int;
    pub fn rimphony_status_histogram_device(
        ctx: *mut rimphony_ctx, n: usize, d_status: *const i32, hist: *mut u64, stream: *mut c_void) -> c_int;

    pub fn rimphony_batch_norm_device(
        ctx: *mut rimphony_ctx, dist_kind: c_int, n: usize,
        d_params: *const *const c_double, d_norm: *mut c_double, stream: *mut c_void) -> c_int;
    pub fn rimphony_bessel_batch_device(
        ctx: *mut rimphony_ctx, count: usize, d_n: *const c_double, d_x: *const c_double,
        d_j: *mut c_double, d_dj: *mut c_double, stream: *mut c_void) -> c_int;

    pub fn rimphony_highfreq_batch_device(
        ctx: *mut rimphony_ctx, dist_kind: c_int, n: usize,
        d_s: *const c_double, d_theta: *const c_double, d_params: *const *const c_double,
        d_out: *mut c_double, stream: *mut c_void) -> c_int;
    pub fn rimphony_highfreq_batch(
        ctx: *mut rimphony_ctx, dist_kind: c_int, n: usize,
        s: *const c_double, theta: *const c_double, params: *const *const c_double,
        out: *mut c_double) -> c_int;

    pub fn rimphony_calc_f_batch(
        ctx: *mut rimphony_ctx, dist_kind: c_int, params: *const c_double, norm_override: c_double, count: usize,
        gamma: *const c_double, cos_xi: *const c_double,
        f: *mut c_double, dfdg: *mut c_double, dfdcx: *mut c_double) -> c_int;

    // leung-bessel/src/lib.rs:36-42 binds these two names already (host functions: no GPU needed)
    pub fn pkgw_bessel_j(n: c_double, x: c_double) -> c_double;
    pub fn pkgw_bessel_dj(n: c_double, x: c_double) -> c_double;
}

// ---- the thin safe layer the main crate would use (src/lib.rs of pkgw/rimphony, beside lib.rs:150-210) ----

/// An owned context: one per GPU (its persistent grids fill the device).
pub struct HipContext {
    raw: *mut rimphony_ctx,
}

// the library serialises calls on one context (include/rimphony_hip.h, rimphony_ctx_create)
unsafe impl Send for HipContext {}
unsafe impl Sync for HipContext {}

fn error_text(rc: c_int) -> String {
    unsafe {
        let what = CStr::from_ptr(rimphony_strerror(rc)).to_string_lossy().into_owned();
        let detail = CStr::from_ptr(rimphony_last_error()).to_string_lossy().into_owned();
        if rc == RIMPHONY_EHIP && !detail.is_empty() {
            format!("{} ({})", what, detail)
        } else {
            what
        }
    }
}

impl HipContext {
    pub fn new(device: i32) -> Result<HipContext, String> {
        let mut raw: *mut rimphony_ctx = std::ptr::null_mut();
        let rc = unsafe { rimphony_ctx_create(device as c_int, &mut raw) };
        if rc != RIMPHONY_OK {
            return Err(error_text(rc));
        }
        Ok(HipContext { raw })
    }

    pub fn as_ptr(&self) -> *mut rimphony_ctx {
        self.raw
    }

    /// N x (full_calculation + compute_all_dimensionless): rows are
    /// [j_I, alpha_I, j_Q, alpha_Q, j_V, alpha_V, rho_Q, rho_V] exactly as
    /// SynchrotronCalculator::compute_all_dimensionless (lib.rs:178-191) returns them; a coefficient whose
    /// integration fails is NaN, as in the reference (symphony.rs:115-117, 127, 380).
    pub fn compute_batch(
        &self, kind: i32, s: &[f64], theta: &[f64], params: &[&[f64]], coeff_mask: u32,
    ) -> Result<Vec<[f64; 8]>, String> {
        let n = s.len();
        let np = unsafe { rimphony_dist_nparams(kind as c_int) };
        if np < 0 || params.len() != np as usize || theta.len() != n || params.iter().any(|p| p.len() != n) {
            return Err(error_text(RIMPHONY_EINVAL));
        }
        let pp: Vec<*const f64> = params.iter().map(|p| p.as_ptr()).collect();
        let mut out = vec![[f64::NAN; 8]; n];
        let rc = unsafe {
            rimphony_batch_compute(
                self.raw, kind as c_int, n, s.as_ptr(), theta.as_ptr(), pp.as_ptr(),
                coeff_mask, out.as_mut_ptr() as *mut f64, std::ptr::null_mut())
        };
        if rc != RIMPHONY_OK {
            return Err(error_text(rc));
        }
        Ok(out)
    }
}

impl Drop for HipContext {
    fn drop(&mut self) {
        unsafe { rimphony_ctx_destroy(self.raw) }
    }
}
