/* rimphony_hip.h -- C ABI of the MI355X-native batched synchrotron-coefficient
 * integrator (librimphony_hip.so).
 *
 * Scope: the per-parameter-point hot path of pkgw/rimphony, i.e. N times
 *     D::new(..).gamma_limits(..).full_calculation(log)        (power_law.rs:71-111,
 *         thermal_juettner.rs:45-72, pitchy_pl.rs:73-115, pitchy_kappa.rs:70-125)
 *     .compute_all_dimensionless(s, theta)                     (src/lib.rs:178-191)
 * evaluated on the GPU for a closed set of distribution functions.  The
 * reference has no batch/FFI interface for this path; its two existing FFI
 * seams are the leung-bessel extern block (leung-bessel/src/lib.rs:36-42) and
 * the GSL callback trampoline (src/gsl.rs:111-117).  A device kernel cannot
 * call back into a host closure, so the second seam is replaced by the
 * `dist_kind` + SoA parameter arrays below; the first is kept as a batched
 * entry point (rimphony_bessel_batch_device).  INTEGRATION.md shows the
 * `-sys` crate a maintainer would add on the Rust side.
 *
 * Conventions
 *   - plain pointers and sizes only; `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream); `d_*` arguments are DEVICE pointers.
 *   - return value: 0 on success, a negative RIMPHONY_E* code on API misuse or
 *     a HIP failure (rimphony_strerror()).  Numerical failure is never an
 *     error: as in the reference (symphony.rs:115-117,127,380) the affected
 *     coefficient is NaN, and the optional status array says why.
 *   - output slot order is that of lib.rs:176-177:
 *         [j_I, alpha_I, j_Q, alpha_Q, j_V, alpha_V, rho_Q, rho_V]
 *     Slots not selected in coeff_mask are written as NaN.
 *   - results depend only on the inputs of a point, never on batch size,
 *     launch geometry or which GPU evaluated it.
 *   - the caller owns every buffer it passes; the library owns its workspace
 *     through the opaque context and never returns memory to the caller.
 */
#ifndef RIMPHONY_HIP_H
#define RIMPHONY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum Stokes (lib.rs:74-87) and enum Coefficient (lib.rs:91-107) */
enum { RIMPHONY_STOKES_I = 0, RIMPHONY_STOKES_Q = 1, RIMPHONY_STOKES_V = 2 };
enum { RIMPHONY_EMISSION = 0, RIMPHONY_ABSORPTION = 1, RIMPHONY_FARADAY = 2 };

/* Distribution kinds; params are SoA arrays, one per parameter, in this order:
 *   POWER_LAW         p, gamma_min, gamma_max, gamma_cutoff          (power_law.rs:27-33, 82-87)
 *   THERMAL_JUETTNER  T                                              (thermal_juettner.rs:45-50)
 *   PITCHY_PL         p, k, gamma_min, gamma_max, gamma_cutoff       (pitchy_pl.rs:73-90)
 *   PITCHY_KAPPA      kappa, width, k, gamma_cutoff                  (pitchy_kappa.rs:70-85) */
enum {
    RIMPHONY_POWER_LAW = 0,
    RIMPHONY_THERMAL_JUETTNER = 1,
    RIMPHONY_PITCHY_PL = 2,
    RIMPHONY_PITCHY_KAPPA = 3
};
int rimphony_dist_nparams(int dist_kind);   /* 4, 1, 5, 4; negative for an unknown kind */

/* coeff_mask bits = output slots */
#define RIMPHONY_SLOT_J_I      (1u << 0)
#define RIMPHONY_SLOT_ALPHA_I  (1u << 1)
#define RIMPHONY_SLOT_J_Q      (1u << 2)
#define RIMPHONY_SLOT_ALPHA_Q  (1u << 3)
#define RIMPHONY_SLOT_J_V      (1u << 4)
#define RIMPHONY_SLOT_ALPHA_V  (1u << 5)
#define RIMPHONY_SLOT_RHO_Q    (1u << 6)
#define RIMPHONY_SLOT_RHO_V    (1u << 7)
#define RIMPHONY_SLOTS_ALL     0xffu

/* per-coefficient status bits (0 = clean) */
#define RIMPHONY_ST_INNER_FAIL  1   /* an inner QAG returned a GSL error -> NaN sample (symphony.rs:380) */
#define RIMPHONY_ST_OUTER_FAIL  2   /* an outer (n / Heyvaerts) QAG failed (symphony.rs:269 `?`)        */
#define RIMPHONY_ST_CHUNK_CAP   4   /* chunk-marching iteration cap hit                                  */
#define RIMPHONY_ST_STORE_FULL  8   /* LDS subinterval store exhausted                                   */
#define RIMPHONY_ST_NONFINITE   16  /* the coefficient is NaN                                            */
#define RIMPHONY_ST_NORM_FAIL   32  /* normalisation integral failed (the reference would panic)         */
#define RIMPHONY_ST_NOT_COMPUTED 64 /* slot not selected / not available                                 */

/* error codes */
#define RIMPHONY_OK         0
#define RIMPHONY_EINVAL    -1
#define RIMPHONY_EHIP      -2
#define RIMPHONY_ENOMEM    -3
#define RIMPHONY_ENODEVICE -4
#define RIMPHONY_EBUSY     -5   /* RIMPHONY_EXCLUSIVE=1 and another context already has the device */
#define RIMPHONY_ENOTSUP   -6   /* e.g. a `precision` this build does not implement; librccl not loadable */
#define RIMPHONY_ERCCL     -7   /* an RCCL call failed (rimphony_last_error() has RCCL's text) */

/* `precision` of the _ex / _multi entry points.  F64 is the reference's arithmetic (bit-identical to the
 * oracle) and the only precision offered.  F32_INTEGRAND (BASELINE configs[4]: the bodies of the elementary functions of
 * the Symphony integrand in single precision on the hardware transcendental unit, every difference, every product with
 * the harmonic number and every quadrature sum in fp64) returns RIMPHONY_ENOTSUP for EVERY distribution, as does any
 * other value: measured, it is slower AND lossier than F64 -- on pitchy_kappa (the distribution configs[4] names) 1.53 x
 * the fp64 time with 1.9 % new NaNs, because its non-smooth 1e-7 noise trips GSL's round-off detectors on integrals that
 * cancel; on the power-law and thermal distributions 0.90 / 0.96 of the fp64 time of the round-2 kernel it is built on,
 * i.e. 1.36 / 1.43 x the time of the fp64 default since that moved to the group kernel (round 3), with ~1e-6 p99
 * differences.  The kernels stay in the library behind a measurement hook -- a context created with
 * RIMPHONY_F32_VARIANT=1 in the environment accepts F32_INTEGRAND for POWER_LAW and THERMAL_JUETTNER
 * (tools/f32_variant.py) -- so that the statement above can be re-measured; configs[4]'s table itself runs in F64. */
#define RIMPHONY_PRECISION_F64            0
#define RIMPHONY_PRECISION_F32_INTEGRAND  1

typedef struct rimphony_ctx rimphony_ctx;

/* Create / destroy a context bound to one HIP device.  Fails with
 * RIMPHONY_ENODEVICE when no GPU is present (there is no CPU fallback).
 * Environment, read once here: RIMPHONY_NO_ASSIST=1 turns the cooperative tail of the kernels off (one
 * wavefront per task to the end; same results bit for bit, used for A/B measurements and by the tests).
 * One context per GPU is the supported configuration: its persistent grids fill the device.  The first context
 * on a device (in any process) takes an exclusive flock on /dev/shm/rimphony_hip.<pci bus id>.lock for its
 * lifetime; a context created while that lock is held runs in SHARED mode (quarter-size grids, cooperative tail
 * off: slower, same results) or, with RIMPHONY_EXCLUSIVE=1 in the environment, is refused with RIMPHONY_EBUSY.
 * Calls on one context are serialised and ordered on the device whatever streams they name; different contexts
 * may be used from different threads freely.
 * RIMPHONY_SYM_SOLO=1 runs the six Symphony coefficients one wave per (point, coefficient) as until round 2 instead of
 * the coefficients of a point in lock-step; RIMPHONY_FARADAY_GROUP=1 runs rho_Q and rho_V of a point in lock-step as well
 * (measured slower than one wave per coefficient, hence off).  Both for A/B measurements: the tables do not change.
 * RIMPHONY_EARLY_SQUAD=<n> sets how many waves of the Faraday kernel's grid serve the longest outer quadratures of a
 * launch from its first cycle instead of fetching tasks (0: none; default 64 for the power-law family, 256 for pitchy kappa,
 * 0 for the other two distributions; only on launches with at least four tasks per wave, never in shared mode),
 * RIMPHONY_EARLY_MIN=<n> the size from which an outer quadrature competes for them (16), RIMPHONY_FARADAY_ORDER=symphony
 * makes the Faraday launch visit the points in the Symphony launch's order; RIMPHONY_ROUNDS=0 makes a long outer
 * quadrature of the pitchy-kappa Faraday kernel evaluate one interval per batch (default: the children of up to four, their
 * rule sums filed ahead of qag.c's picks; the other distributions' kernels are built without rounds).  Scheduling only: the tables do not change.
 * RIMPHONY_OWNER_WAIT_US=<n> (test hook) shortens the 120 s an owner wave waits for its helpers before it recomputes
 * a published batch itself; results do not depend on it. */
int rimphony_ctx_create(int device, rimphony_ctx **out);
void rimphony_ctx_destroy(rimphony_ctx *ctx);
int rimphony_ctx_shared_mode(const rimphony_ctx *ctx);   /* 0: owns the device, 1: shared mode */
int rimphony_ctx_device(const rimphony_ctx *ctx, int *device);   /* the HIP device the context is bound to */
const char *rimphony_strerror(int code);
/* Text of the most recent failure on the calling thread (which HIP call failed and why); "" if none.  The library
 * never prints. */
const char *rimphony_last_error(void);
const char *rimphony_version(void);

/* Work counters of the most recent batch call on this context (device-side
 * counts, read back synchronously): integrand samples, wave-wide evaluation
 * passes, inner QAG calls. */
typedef struct {
    uint64_t samples;               /* Symphony kernel */
    uint64_t passes;
    uint64_t inner_qags;
    uint64_t faraday_samples;       /* Heyvaerts kernel */
    uint64_t faraday_passes;
    uint64_t faraday_inner_qags;
} rimphony_work;
int rimphony_last_work(rimphony_ctx *ctx, rimphony_work *out);

/* The tail of the most recent batch call (device-side, read back synchronously).  Per-task cost has a heavy tail and the
 * reference's chunk marching is sequential: the task with the longest CHAIN of batches (each waits for its slowest
 * gamma-integral / inner integral) bounds how early a launch can end, whatever the cooperative tail spreads out.
 *   out[0], out[1]   Symphony: batches of the heaviest coefficient, and its row (rows to 2^40)
 *   out[2], out[3]   Faraday: the same
 *   out[4]           Symphony group kernel: passes the coefficients would have executed one by one (out[4] / passes of
 *                    rimphony_last_work = how many coefficients an executed pass served on average)
 *   out[5]           ... rule sums filed ahead of a coefficient's own pick (the stash of symphony_group.h)
 *   out[6], out[7]   the same two for the Faraday pair (heyvaerts_group.h) */
int rimphony_last_tail(rimphony_ctx *ctx, uint64_t out[8]);

/* Duration of the most recent Symphony kernel launch of this context, measured
 * with HIP events recorded on the stream the kernel was launched on (waits for
 * the kernel to finish). */
int rimphony_last_symphony_ms(rimphony_ctx *ctx, float *ms);
/* Same for the Faraday (Heyvaerts) launch of the most recent batch call. */
int rimphony_last_faraday_ms(rimphony_ctx *ctx, float *ms);

/* Diagnostics: 16 heartbeat words in host-mapped memory, updated by the wave that
 * works on task number `task` (= point * nslots + slot index) of subsequent
 * batch calls while the kernel runs -- the counterpart of the reference drivers'
 * "write the parameters before computing so a hang can be reconstructed"
 * (examples/crank-out-pitchykappa.rs:193-200).  Words: [0] task+1, [1] batches,
 * [2] phase, [3] integrand passes, [4] inner QAG iteration, [5] chunks,
 * [6] n_start bits, [7] delta_n bits, [8] lane of the running gamma-integral,
 * [9] its n (bits), [10] 1 when the task has finished.  Tasks of the Heyvaerts kernel
 * are addressed as task | (1 << 62).  Symphony tasks are watched in the one-wave-per-coefficient kernel only
 * (RIMPHONY_SYM_SOLO=1 when the context is created); the group kernel does not write heartbeats, so for a Symphony
 * task of a context without that setting the call returns RIMPHONY_ENOTSUP instead of handing out words that would
 * never change. */
int rimphony_debug_heartbeat(rimphony_ctx *ctx, uint64_t task, uint64_t **host_words);

/* Diagnostics: the 16 raw device counter words of the most recent batch call: [0] task head,
 * [1..3] Symphony samples / passes / inner QAGs, [4] Faraday task head, [5..7] Faraday work,
 * [8] batches published on the assist board, [9] requests evaluated by helper waves,
 * [10] requests of published batches evaluated by their owner, [11] owner wait (100 MHz ticks),
 * [12] helper polls, [13] helper visits that found every request already claimed, [14] ticks spent
 * evaluating requests, [15] the task with the most batches as (batches << 24) | point index.  Words 8..15 are only counted by a library built with
 * -DRIM_COOP_DIAG (tools/ab_assist.py); they are 0 otherwise.  `out` must hold 32 words (a -DRIM_PROF build
 * returns its 32 region timers instead, tools/region_profile.py). */
int rimphony_debug_counters(rimphony_ctx *ctx, uint64_t out[32]);

/* The batched compute(): N x (full_calculation + compute_all_dimensionless).
 *   d_s, d_theta   [n]                device
 *   d_params       host array of rimphony_dist_nparams(kind) DEVICE pointers, each [n]
 *   d_out          [n][8] row-major   device
 *   d_status       [n][8] int32 or NULL
 * Asynchronous on `stream`. */
int rimphony_batch_compute_device(rimphony_ctx *ctx, int dist_kind, size_t n,
                                  const double *d_s, const double *d_theta,
                                  const double *const *d_params, uint32_t coeff_mask,
                                  double *d_out, int32_t *d_status, void *stream);

/* Same with HOST buffers (copies in, computes, copies out, synchronises). */
int rimphony_batch_compute(rimphony_ctx *ctx, int dist_kind, size_t n,
                           const double *s, const double *theta,
                           const double *const *params, uint32_t coeff_mask,
                           double *out, int32_t *status);

/* The same two with the remaining arguments of SURVEY 8b's batch ABI:
 *   precision   RIMPHONY_PRECISION_*
 *   d_work/work optional [n][8] uint64: integrand samples spent on each coefficient (0 for unselected slots) --
 *               the per-point counterpart of the reference's per-call trace lines (symphony.rs:69-75, 217-223,
 *               272-275) and of the crank-out drivers' per-row `time_ms(meta)` column
 *               (examples/crank-out-pitchypl.rs:167-173): cost in units that do not depend on what else ran. */
int rimphony_batch_compute_device_ex(rimphony_ctx *ctx, int dist_kind, size_t n,
                                     const double *d_s, const double *d_theta,
                                     const double *const *d_params, uint32_t coeff_mask, int precision,
                                     double *d_out, int32_t *d_status, uint64_t *d_work, void *stream);
int rimphony_batch_compute_ex(rimphony_ctx *ctx, int dist_kind, size_t n,
                              const double *s, const double *theta,
                              const double *const *params, uint32_t coeff_mask, int precision,
                              double *out, int32_t *status, uint64_t *work);

/* Multi-GPU batch (SURVEY 8b `n_devices`, 8e): HOST buffers, one context per device in ctxs[0..n_ctx-1].  Row i is
 * evaluated by ctxs[i mod n_ctx] (interleaved sharding; one host thread per context; no exchange between devices
 * while they compute) and lands in row i of out / status / work.  The table does not depend on n_ctx.  This is the
 * in-process form; the one-process-per-GPU form with an RCCL gather is rimphony_amd/sharding.py + bench.py. */
int rimphony_batch_compute_multi(rimphony_ctx *const *ctxs, int n_ctx, int dist_kind, size_t n,
                                 const double *s, const double *theta, const double *const *params,
                                 uint32_t coeff_mask, int precision, double *out, int32_t *status, uint64_t *work);

/* The same with DEVICE buffers: ctxs[r] (bound to its own device) evaluates the n_local[r] rows its arrays hold --
 * d_s[r], d_theta[r], d_params[r][k], d_out[r] ([n_local[r]][8]), optionally d_status[r], d_work[r], on streams[r] (or the
 * null stream if `streams` is NULL).  All launches are issued from the calling thread before anything is waited for; with
 * `synchronize` != 0 the call returns when every device has finished, otherwise it is asynchronous on the given streams.
 * Which rows a context holds is the caller's choice; rimphony_rccl_gather_table assumes the interleaved convention
 * (row i of the table on rank i mod world).  Replaces N concurrent calls of lib.rs:178-191's loop body. */
int rimphony_batch_compute_multi_device(rimphony_ctx *const *ctxs, int n_ctx, int dist_kind, const size_t *n_local,
                                        const double *const *d_s, const double *const *d_theta,
                                        const double *const *const *d_params, uint32_t coeff_mask, int precision,
                                        double *const *d_out, int32_t *const *d_status, uint64_t *const *d_work,
                                        void *const *streams, int synchronize);

/* The one collective of the path, for a host that is not Python (north_star: "RCCL gather over xGMI for the output
 * table"; the Python form is rimphony_amd/sharding.py).  librccl.so.1 is dlopen'ed on the first of these calls
 * (RIMPHONY_RCCL_LIB overrides the name); the library does not link against it.  One process per GPU:
 *   rimphony_rccl_available()        1 if librccl could be loaded, else 0 (rimphony_last_error() says why)
 *   rimphony_rccl_unique_id(id)      rank 0: fill 128 bytes, to be handed to the other ranks by the caller's own means
 *   rimphony_rccl_comm_create(..)    ncclCommInitRank on the context's device; *comm is an ncclComm_t
 *   rimphony_rccl_gather_table(..)   every rank: d_shard = its [ceil((n_total - rank) / world)][8] rows of the interleaved
 *                                    table (device); root: d_table [n_total][8] (device) receives row i from rank i mod
 *                                    world.  One grouped send / receive per rank (the root sends to itself too, so a world
 *                                    of one runs the same RCCL calls) and an un-interleave kernel, all on `stream`.
 *                                    d_scratch: n_total * 8 doubles on the root (ignored elsewhere); NULL = allocated
 *                                    and freed inside, and the call then returns synchronised.
 *   rimphony_rccl_comm_destroy(comm)
 * A comm created elsewhere (the caller's own RCCL binding) is accepted as well: it is only passed through. */
int rimphony_rccl_available(void);
int rimphony_rccl_unique_id(void *id128);
int rimphony_rccl_comm_create(rimphony_ctx *ctx, int rank, int world, const void *id128, void **comm);
int rimphony_rccl_comm_destroy(void *comm);
int rimphony_rccl_gather_table(rimphony_ctx *ctx, void *comm, int rank, int world, int root, size_t n_total,
                               const double *d_shard, double *d_table, double *d_scratch, void *stream);

/* Status histogram of a computed table: hist[slot * 8 + b] = rows whose status word of `slot` has bit b set
 * (b = 0..6, the RIMPHONY_ST_* bits in order), hist[slot * 8 + 7] = rows with status 0.  Synchronous. */
int rimphony_status_histogram_device(rimphony_ctx *ctx, size_t n, const int32_t *d_status, uint64_t hist[64], void *stream);

/* full_calculation() alone: the normalisation constant of each point
 * (power_law.rs:93-103 etc.); NaN where the integral failed. */
int rimphony_batch_norm_device(rimphony_ctx *ctx, int dist_kind, size_t n,
                               const double *const *d_params, double *d_norm, void *stream);

/* leung-bessel seam (leung-bessel/src/lib.rs:36-42), batched:
 * d_j[i] = pkgw_bessel_j(d_n[i], d_x[i]), d_dj[i] = pkgw_bessel_dj(d_n[i], d_x[i]);
 * either output may be NULL. */
int rimphony_bessel_batch_device(rimphony_ctx *ctx, size_t count, const double *d_n, const double *d_x,
                                 double *d_j, double *d_dj, void *stream);

/* Diagnostic seam: FullSynchrotronCalculator::diagnostic_symphony_gamma_integrand
 * (lib.rs:278-285), batched over (n, gamma) pairs for ONE parameter point given
 * as host scalars.  params: host array of the kind's parameters. */
int rimphony_gamma_integrand_batch_device(rimphony_ctx *ctx, int dist_kind, const double *params,
                                          int coeff, int stokes, double s, double theta,
                                          size_t count, const double *d_n, const double *d_gamma,
                                          double *d_out, void *stream);

/* Diagnostic seam: diagnostic_symphony_gamma_integral (lib.rs:266-272), batched
 * over orders n for one parameter point; negative_lobe selects the Stokes-V lobe. */
int rimphony_gamma_integral_batch_device(rimphony_ctx *ctx, int dist_kind, const double *params,
                                         int coeff, int stokes, int negative_lobe, double s, double theta,
                                         size_t count, const double *d_n, double *d_out, void *stream);

/* Unit seam: one leaf function of rimphony_amd/csrc/detmath.h over arrays, as the kernels evaluate it
 * (the parity contract is that gcc/x86-64 and hipcc/gfx950 give the same bits for these).
 * op: 0 exp, 1 log, 2 log10, 3 pow(x, y), 4 sqrt, 5 log10_region, 6 lgamma (x > 0), 7 sin, 8 cos,
 * 9 rim_div_by(x, y, 1/y), 10 cbrt (positive normal x), 11 1/Gamma(x) (-8.5 < x < 9.5), 12 rim_third_powers(x)[(int) y],
 * 13 x^(-1/4) (positive normal x), 14 rim_powexp_normal(x, y, -x/1000).  d_y is only read by ops 3, 9, 12 and 14. */
int rimphony_detmath_batch_device(rimphony_ctx *ctx, int op, size_t n, const double *d_x, const double *d_y,
                                  double *d_out, void *stream);

/* The reference's own scalar FFI seam, leung-bessel/src/lib.rs:36-42
 *   extern { fn pkgw_bessel_j(n: c_double, x: c_double) -> c_double; fn pkgw_bessel_dj(...) -> c_double; }
 * exported under the same names so that crate can link this library in place of leung-bessel/src/bessel.c.
 * HOST code (the host build of the device function): no HIP call, works without a GPU, reentrant; returns the
 * same bits as rimphony_bessel_batch_device.  Failure -> NaN (the seam's convention, bessel.c:327-333, 382-388). */
double pkgw_bessel_j(double n, double x);
double pkgw_bessel_dj(double n, double x);

/* High-frequency closed-form Faraday coefficients: `high_freq_approximation()` of
 * PowerLawDistribution (power_law.rs:117-170; d_params = {p, gamma_min, ...}) and of
 * ThermalJuettnerDistribution (thermal_juettner.rs:78-142; d_params = {T}).  Other kinds: RIMPHONY_EINVAL
 * (the reference has none).  d_out is [n][2] = {rho_Q, rho_V}, dimensionless.  As in the reference no
 * check is made that the parameters lie where the approximation is good. */
int rimphony_highfreq_batch_device(rimphony_ctx *ctx, int dist_kind, size_t n, const double *d_s, const double *d_theta,
                                   const double *const *d_params, double *d_out, void *stream);
/* The same with host buffers (synchronous). */
int rimphony_highfreq_batch(rimphony_ctx *ctx, int dist_kind, size_t n, const double *s, const double *theta,
                            const double *const *params, double *out);

/* Diagnostic seam: diagnostic_symphony_n_integral (lib.rs:254-260 -> symphony.rs:298-307), the QAG over the
 * harmonic number n in [n_lo[i], n_hi[i]] of the gamma-integral, for one parameter point.  A GSL error of the
 * reference (its Err) is NaN here. */
int rimphony_n_integral_batch_device(rimphony_ctx *ctx, int dist_kind, const double *params,
                                     int coeff, int stokes, int negative_lobe, double s, double theta,
                                     size_t count, const double *d_n_lo, const double *d_n_hi, double *d_out, void *stream);

/* Unit seam for gsl::deriv_central (gsl.rs:233-257 -> gsl_deriv_central: 5-point central difference with one step-size
 * refinement) as n_integration drives it (symphony.rs:238-240): d_out[i] = the accepted derivative estimate of the
 * gamma-integral with respect to the harmonic number at d_n_start[i], step h = 1e-10 n_start, computed by the
 * derivative-probe phases of the coefficient's state machine (the code the product kernels run). */
int rimphony_deriv_probe_batch_device(rimphony_ctx *ctx, int dist_kind, const double *params,
                                      int coeff, int stokes, int negative_lobe, double s, double theta,
                                      size_t count, const double *d_n_start, double *d_out, void *stream);

/* Diagnostic seam: diagnostic_symphony_gamma_contribution (lib.rs:288-296 -> symphony.rs:491-567), the
 * contribution of all harmonics at fixed gamma (cgs-scaled like a coefficient), for one parameter point. */
int rimphony_gamma_contribution_batch_device(rimphony_ctx *ctx, int dist_kind, const double *params, int coeff, int stokes,
                                             double s, double theta, size_t count, const double *d_gamma, double *d_out,
                                             void *stream);

/* The DistributionFunction trait (lib.rs:111-146): calc_f(gamma, cos_xi) and calc_f_derivatives(gamma, cos_xi) of ONE
 * distribution (host `params`, kind-specific as above) over device arrays of count (gamma, cos_xi) pairs.  Any of
 * d_f, d_dfdg, d_dfdcx may be NULL.  norm_override: NaN = the distribution's own normalisation (what
 * full_calculation computes); any other value is used as the normalisation constant -- the reference's derivative
 * tests set it to 1 (pitchy_pl.rs:216-217, pitchy_kappa.rs:149-150). */
int rimphony_calc_f_batch_device(rimphony_ctx *ctx, int dist_kind, const double *params, double norm_override,
                                 size_t count, const double *d_gamma, const double *d_cos_xi,
                                 double *d_f, double *d_dfdg, double *d_dfdcx, void *stream);
/* the same with host buffers (synchronous) */
int rimphony_calc_f_batch(rimphony_ctx *ctx, int dist_kind, const double *params, double norm_override, size_t count,
                          const double *gamma, const double *cos_xi, double *f, double *dfdg, double *dfdcx);

/* Unit seams of the Heyvaerts (Faraday) path for ONE parameter point (host `params`), stokes = RIMPHONY_STOKES_Q (the
 * "h" elements, rho_Q) or _V (the "f" elements, rho_V):
 *   element: the inner integrand -- qr != 0: h_qr / f_qr_element at sigma = d_fixed[i], pomega = d_v[i]
 *            (heyvaerts.rs:302-373, 400-447); qr == 0: h_nr / f_nr_element at pomega = d_fixed[i], sigma = d_v[i]
 *            (heyvaerts.rs:379-394, 453-468);
 *   outer:   the outer integrand = one inner integral -- qr_outer_integrand(sigma = d_u[i]) / nr_outer_integrand(pomega =
 *            d_u[i]) (heyvaerts.rs:213-250, 262-296); NaN where the inner QAG fails. */
int rimphony_hey_element_batch_device(rimphony_ctx *ctx, int dist_kind, const double *params, int stokes, double s, double theta,
                                      int qr, size_t count, const double *d_fixed, const double *d_v, double *d_out, void *stream);
int rimphony_hey_outer_batch_device(rimphony_ctx *ctx, int dist_kind, const double *params, int stokes, double s, double theta,
                                    int qr, size_t count, const double *d_u, double *d_out, void *stream);

/* Self-test seam for the wavefront QAG (gsl.rs:156-207 semantics) on built-in
 * integrands made of + - * / sqrt only, so CPU and GPU agree bit for bit:
 *   family 0: 1 / (1 + ((x - p0) * p1)^2)         family 1: sqrt(|x - p0|) * p1
 *   family 2: x^2 * (p0 + x * p1)                  family 3: 1 / sqrt(|x - p0| + p1)
 *   family 4: p1 * |frac(x * p0) - 1/2|  (triangle wave: forces long subinterval lists)
 * Arrays of `count` problems; outputs result, abserr, status, size. */
int rimphony_qag_selftest_device(rimphony_ctx *ctx, size_t count, const int32_t *d_family,
                                 const double *d_p0, const double *d_p1, const double *d_a, const double *d_b,
                                 double epsabs, double epsrel, int32_t limit,
                                 double *d_result, double *d_abserr, int32_t *d_qstatus, int32_t *d_size,
                                 void *stream);

#ifdef __cplusplus
}
#endif
#endif
