/* rimo.h -- CPU oracle for the rimphony hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory restates, in plain C, the per-parameter-point algorithm of
 * pkgw/rimphony (symphony.rs, heyvaerts.rs, gsl.rs + the GSL routines they
 * call, the four distribution functions, and leung-bessel/src/bessel.c) so
 * that the HIP kernels can be checked against it.  Nothing in the product
 * (rimphony_amd/, include/) may call, link or import anything from here; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Pinning: the Rust crate cannot be built here (no
 * cargo/rustc, no GSL, special-fun un-vendored) and bessel.c needs the absent
 * <gsl/gsl_sf_bessel.h>, so there is no oracle/_ref.  The oracle is pinned by
 * the reference's own fixtures: tests/symphony-powerlaw.txt (200 x 6
 * coefficients, 1 %), the Faraday known answers in power_law.rs:209-240 and
 * thermal_juettner.rs:174-210 (1 %), normalisation = 1, pitchy_pl(k=0) ==
 * power_law, the finite-difference derivative checks and the n<30 Bessel
 * smoke values -- see tests/test_oracle_*.py.  pitchy_kappa coefficients have
 * no absolute fixture in the reference: parity unpinned for those.
 *
 * Arithmetic: two builds of the same sources.
 *   liboracle.so       elementary functions from rimphony_amd/csrc/detmath.h
 *                      and GK31 sums in the fixed 32-leaf tree order the
 *                      wavefront kernel uses -> comparable BIT FOR BIT with
 *                      the GPU.
 *   liboracle_libm.so  (-DRIMO_LIBM) glibc libm and GSL's sequential GK31
 *                      summation order, i.e. as close to the Rust/GSL binary
 *                      as can be had here; used to show that the two choices
 *                      above move results only at rounding level.
 */
#ifndef RIMO_H
#define RIMO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* lib.rs:74-107 */
enum { RIMO_STOKES_I = 0, RIMO_STOKES_Q = 1, RIMO_STOKES_V = 2 };
enum { RIMO_EMISSION = 0, RIMO_ABSORPTION = 1, RIMO_FARADAY = 2 };

/* distribution kinds and their parameter vectors (same order as the C ABI) */
enum {
    RIMO_POWER_LAW = 0,        /* p, gamma_min, gamma_max, gamma_cutoff        power_law.rs:27-33   */
    RIMO_THERMAL_JUETTNER = 1, /* T                                             thermal_juettner.rs:23-26 */
    RIMO_PITCHY_PL = 2,        /* p, k, gamma_min, gamma_max, gamma_cutoff      pitchy_pl.rs:22-29  */
    RIMO_PITCHY_KAPPA = 3      /* kappa, width, k, gamma_cutoff                 pitchy_kappa.rs:28-35 */
};
#define RIMO_MAX_PARAMS 6

typedef struct {
    int kind;
    double par[RIMO_MAX_PARAMS];
    double inv_gamma_cutoff;
    double inv_kappa_width;
    double neg_inverse_t;
    double norm;
} rimo_dist;

/* GSL status codes used by the path (gsl_errno.h values) */
enum { RIMO_SUCCESS = 0, RIMO_EFAILED = 5, RIMO_EMAXITER = 11, RIMO_EBADTOL = 13, RIMO_EROUND = 18, RIMO_ESING = 21 };

/* work counters: the algorithmic-work figures of SURVEY.md 8(d) */
typedef struct {
    uint64_t integrand_evals;   /* gamma_integrand / heyvaerts element samples */
    uint64_t gk_evals;          /* 31-point rule applications (inner integrals) */
    uint64_t inner_qag_calls;   /* gamma_integral / inner sigma|pomega QAGs      */
    uint64_t outer_gk_evals;    /* 31-point rule applications of outer QAGs     */
    uint64_t outer_qag_calls;   /* n-chunks / outer Heyvaerts QAGs              */
    uint64_t deriv_calls;
    uint64_t max_inner_size;    /* largest subinterval count seen               */
    uint64_t max_outer_size;
    uint64_t bessel_calls;
    uint64_t norm_evals;
    /* Faraday (Heyvaerts) sample mix, for the per-sample flop model of bench.py / DESIGN.md */
    uint64_t hey_nr_samples;    /* non-resonant element samples                                   */
    uint64_t hey_qr_i_samples;  /* quasi-resonant samples on the I_{+-1/3}, I_{+-2/3} branch (g < 10) */
    uint64_t hey_qr_jy_samples; /* quasi-resonant samples on the J_nu / Y_nu branch (g >= 10)        */
    uint64_t hey_series_terms;  /* terms summed by the ascending Bessel series                      */
    uint64_t hey_series_calls;
} rimo_counters;

typedef double (*rimo_fn)(double x, void *ctx);

/* --- gsl.rs:58-70, 156-257 + the GSL algorithms behind them ------------- */
typedef struct {
    size_t limit, size, nrmax, i, maximum_level;
    double *alist, *blist, *rlist, *elist;
    size_t *order, *level;
} rimo_workspace;
rimo_workspace *rimo_workspace_alloc(size_t limit);   /* gsl_integration_workspace_alloc */
void rimo_workspace_free(rimo_workspace *w);

/* gsl_integration_qag(..., key = 3 (GK31), ...): returns a GSL status; the Rust
 * wrapper turns any non-zero status into Err (gsl.rs:198-199). */
int rimo_qag(rimo_fn f, void *ctx, double a, double b, double epsabs, double epsrel,
             size_t limit, rimo_workspace *w, double *result, double *abserr, uint64_t *gk_evals);
/* convenience for tests: allocates its own workspace */
int rimo_qag_gk31(rimo_fn f, void *ctx, double a, double b, double epsabs, double epsrel,
                  size_t limit, double *result, double *abserr, size_t *size_out, uint64_t *gk_evals);
void rimo_qk31(rimo_fn f, void *ctx, double a, double b,
               double *result, double *abserr, double *resabs, double *resasc);
int rimo_deriv_central(rimo_fn f, void *ctx, double x, double h, double *result, double *abserr);
double rimo_hyperg_2F1_at_1(double a, double b, double c);

/* --- leung-bessel ------------------------------------------------------ */
double rimo_bessel_j(double n, double x);   /* pkgw_bessel_j  bessel.c:318-376 */
double rimo_bessel_dj(double n, double x);  /* pkgw_bessel_dj bessel.c:379-405 */
double rimo_bessel_jn_int(int n, double x); /* stands in for gsl_sf_bessel_Jn  */

/* --- special functions behind heyvaerts.rs (special-fun crate in the reference) --- */
double rimo_gamma_real(double z);
double rimo_bessel_i(double nu, double x);     /* x.besseli(nu)  */
double rimo_bessel_jnu(double nu, double x);   /* x.besselj(nu), ascending series (small x) */
double rimo_bessel_ynu(double nu, double x);   /* x.bessely(nu), reflection formula */
/* { J_sigma, Y_sigma, J_(sigma-1), Y_(sigma-1) } at x as the quasi-resonant elements of this flavour obtain them */
void rimo_bessel_jy_set(double sigma, double x, double out[4]);

/* --- distributions ----------------------------------------------------- */
int rimo_dist_init(rimo_dist *d, int kind, const double *params); /* new()+limits+full_calculation(): 0 or GSL status */
double rimo_calc_f(const rimo_dist *d, double gamma, double cos_xi);
void rimo_calc_f_derivatives(const rimo_dist *d, double gamma, double cos_xi, double *dfdg, double *dfdcx);

/* --- calculators -------------------------------------------------------- */
double rimo_symphony(const rimo_dist *d, int coeff, int stokes, double s, double theta, rimo_counters *c);
double rimo_heyvaerts(const rimo_dist *d, int coeff, int stokes, double s, double theta, rimo_counters *c);
double rimo_compute_dimensionless(const rimo_dist *d, int coeff, int stokes, double s, double theta, rimo_counters *c);
void rimo_compute_all_dimensionless(const rimo_dist *d, double s, double theta, double out[8], rimo_counters *c);
double rimo_compute_cgs(const rimo_dist *d, int coeff, int stokes, double nu, double b, double n_e, double theta);

double rimo_hey_element(const rimo_dist *d, int stokes, double s, double theta, int qr, double fixed, double v);
/* diagnostic: one OUTER-integrand sample = one inner integral (heyvaerts.rs:213-250 nr_outer_integrand at pomega = u,
 * 262-296 qr_outer_integrand at sigma = u); NaN where the inner QAG fails, as `.unwrap_or(NAN)` does upstream */
double rimo_hey_outer_integrand(const rimo_dist *d, int stokes, double s, double theta, int qr, double u);

/* diagnostics (lib.rs:254-298) */
double rimo_gamma_integrand(const rimo_dist *d, int coeff, int stokes, double s, double theta, double n, double gamma);
double rimo_gamma_contribution(const rimo_dist *d, int coeff, int stokes, double s, double theta, double gamma);
int rimo_n_integral(const rimo_dist *d, int coeff, int stokes, int negative_lobe, double s, double theta,
                    double n_lo, double n_hi, double *value);
double rimo_gamma_integral(const rimo_dist *d, int coeff, int stokes, int negative_lobe, double s, double theta, double n);
/* gsl::deriv_central as n_integration calls it (symphony.rs:238-240): d(gamma_integral)/dn at n_start, h = 1e-10 n_start */
double rimo_symphony_deriv_probe(const rimo_dist *d, int coeff, int stokes, int negative_lobe, double s, double theta, double n_start);

/* batch: N x (full_calculation + selected coefficients); params SoA [nparams][n];
 * out [n][8] row-major in the order of lib.rs:176-177; counters may be NULL
 * (else one struct summed over the batch). nthreads<=1: serial. */
int rimo_batch(int kind, size_t n, const double *s, const double *theta, const double *const *params,
               uint32_t coeff_mask, double *out, rimo_counters *counters, int nthreads);

/* high-frequency closed forms (power_law.rs:133-170, thermal_juettner.rs:94-142): out = {rho_Q, rho_V} */
int rimo_highfreq(int kind, const double *params, double s, double theta, double out[2]);
void rimo_bessel_k012(double x, double k[3]);

int rimo_batch_norm(int kind, size_t n, const double *const *params, double *norm);

/* QAG self-test on integrands built from + - * / sqrt only (see include/rimphony_hip.h) */
int rimo_qag_selftest(int family, double p0, double p1, double a, double b, double epsabs, double epsrel,
                      size_t limit, double *result, double *abserr, size_t *size_out);

const char *rimo_build_flavour(void);
/* Investigation knob (tools/faraday_tail_attribution.py only): in the attribution build (liboracle_attr.so, `make attr`)
 * bit k set = reformulation k of rimo_math.h's RIMO_ATTR_* list evaluated in its LITERAL form; 0 = the deterministic
 * flavour, RIMO_ATTR_ALL = the literal flavour.  Process-global; a no-op in every other build. */
void rimo_set_attr_mask(unsigned mask);

/* Investigation knobs (tools/ only; every test and the bench run with the defaults = the reference's values):
 * tolerances and truncation parameters of symphony.rs / heyvaerts.rs that the reference hard-codes, and this
 * restatement's own caps on the marching loops.  NULL restores the defaults.  Process-global, not thread-safe to
 * change while a batch runs.  exact_bessel != 0 replaces the Leung approximations by glibc's jn() for integer orders
 * (only meaningful for the discrete harmonics: what an exact J_n would change). */
typedef struct {
    double epsrel_gamma;    /* 1e-3   symphony.rs:375-380                    */
    double epsrel_n;        /* 1e-3   symphony.rs:264-269                    */
    double tail_tolerance;  /* 1e5    symphony.rs:202, 225                   */
    int n_discrete;         /* 30     symphony.rs:96 (N_MAX)                 */
    int max_chunks;         /* 4096   this restatement's cap   */
    int hey_max_steps;      /* 4096   the same for heyvaerts.rs:102,134,161  */
} rimo_tuning;
void rimo_set_tuning(const rimo_tuning *t);
/* Investigation knob (tools/nan_pattern.py only): count Symphony integrand samples below DBL_MIN as 0.  Default off. */
void rimo_set_flush_subnormal_samples(int on);
/* Investigation knob (tools/sample_overlap.py only): record a 64-bit key of (harmonic, gamma) for every Symphony
 * integrand sample the calling thread evaluates, up to cap keys; NULL switches it off. */
void rimo_set_sample_log(uint64_t *buf, size_t cap);
size_t rimo_sample_log_count(void);
/* Investigation knob (tools/lockstep_sim.py only): the raw bit patterns (n, gamma) of every Symphony integrand sample of
 * the calling thread, in evaluation order, up to cap_pairs pairs; NULL switches it off. */
void rimo_set_sample_raw_log(uint64_t *buf, size_t cap_pairs);
size_t rimo_sample_raw_count(void);
void rimo_get_tuning(rimo_tuning *t);
/* Investigation knobs (tools/nan_rootcause.py only), thread-local: a log of the quadratures that ended in a GSL error
 * (level 0 = gamma-integral, symphony.rs:375-380; level 1 = n-chunk, symphony.rs:264-269), and a per-bisection trace of
 * the next rimo_qag call (14 doubles per iteration: a_i b_i r_i e_i area1 error1 resasc1 area2 error2 resasc2 rt1 rt2
 * errsum tolerance; row 0 holds the first rule application: a b result0 abserr0 resabs0 resasc0). */
typedef struct { double n, a, b, result, abserr; int lobe, status, size, level; } rimo_fail_rec;
void rimo_set_fail_log(rimo_fail_rec *buf, size_t cap);
size_t rimo_fail_log_count(void);
void rimo_set_qag_trace(double *buf, size_t cap_rows);
size_t rimo_qag_trace_rows(void);

#ifdef __cplusplus
}
#endif
#endif
