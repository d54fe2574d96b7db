/* rimo_symphony.c -- oracle restatement of src/symphony.rs and the dispatch /
 * scaling layer of src/lib.rs (TEST INFRASTRUCTURE ONLY, see rimo.h).
 *
 *   CalculationState::compute   symphony.rs:66-187
 *   n_integration               symphony.rs:196-295
 *   gamma_integral              symphony.rs:312-389
 *   gamma_integrand             symphony.rs:398-479
 *   compute_cgs / compute_all   lib.rs:163-209,  dispatch lib.rs:236-247
 */
#include <stdlib.h>
#include <string.h>
#include "rimo.h"
#include "rimo_math.h"

/* lib.rs:55-67 */
#define TWO_PI (2. * RIM_PI)
#define MASS_ELECTRON 9.1093826e-28
#define SPEED_LIGHT 2.99792458e10
#define ELECTRON_CHARGE 4.80320680e-10

/* hard cap on n-chunks: the reference's while loop (symphony.rs:225) has none and
 * relies on pkgw_bessel_dj going NaN at n >= 1e15; the kernels share this cap */
#define RIMO_MAX_CHUNKS 4096

static const rimo_tuning TUNING_DEFAULT = { 1e-3, 1e-3, 1e5, 30, RIMO_MAX_CHUNKS, 4096 };
static rimo_tuning g_tuning = { 1e-3, 1e-3, 1e5, 30, RIMO_MAX_CHUNKS, 4096 };
void rimo_set_tuning(const rimo_tuning *t) { g_tuning = t ? *t : TUNING_DEFAULT; }
void rimo_get_tuning(rimo_tuning *t) { *t = g_tuning; }

typedef struct {
    const rimo_dist *d;
    int coeff, stokes;
    double s, cos_observer_angle, sin_observer_angle;
    int negative_lobe;           /* StokesVSwitch */
    rimo_workspace *gamma_ws;
    rimo_counters *c;
    double cur_n;                /* n of the gamma integral being evaluated */
} sym_state;

static double gamma_integrand(sym_state *st, double gamma, double n)
{
    const double s = st->s;
    const double cos_th = st->cos_observer_angle, sin_th = st->sin_observer_angle;

    const double beta = m_sqrt(1. - 1. / (gamma * gamma));
    const double cos_xi = (s * gamma - n) / (s * gamma * beta * cos_th);
    const double sin_xi = m_sqrt(1. - cos_xi * cos_xi);
    const double m = (cos_th - beta * cos_xi) / sin_th;
    const double big_n = beta * sin_xi;

    double gamma_sin_xi;
    if (beta < 0.1) {
        gamma_sin_xi = gamma * sin_xi;
    } else {
        const double bc = beta * cos_th;
        const double beta2_costh2 = bc * bc;
        const double s_on_r = 2. * n / (s * (beta2_costh2 - 1.));
        const double r = 1. - 1. / beta2_costh2;
        gamma_sin_xi = m_sqrt(r * (gamma * (gamma + s_on_r)) - (n * n / (s * s * beta2_costh2)));
    }

    const double z = s * beta * sin_th * gamma_sin_xi;

    const double mj = m * rimo_bessel_j(n, z);
    const double njp = big_n * rimo_bessel_dj(n, z);

    double pol_term;
    switch (st->stokes) {
    case RIMO_STOKES_I: pol_term = mj * mj + njp * njp; break;
    case RIMO_STOKES_Q: pol_term = mj * mj - njp * njp; break;
    default: pol_term = 2. * mj * njp; break;
    }

    double f_term;
    if (st->coeff == RIMO_EMISSION) {
        f_term = rimo_calc_f(st->d, gamma, cos_xi);
    } else {
        double dfdg, dfdcx;
        rimo_calc_f_derivatives(st->d, gamma, cos_xi, &dfdg, &dfdcx);
        const double dfdcx_factor = (beta * cos_th - cos_xi) / (gamma - 1. / gamma);
        f_term = dfdg + dfdcx_factor * dfdcx;
    }

    if (st->c) { st->c->integrand_evals++; st->c->bessel_calls += 3; }
    return gamma * gamma * pol_term * f_term;
}

/* investigation knob (tools/nan_pattern.py): integrand samples in the subnormal range count as 0 */
static int g_flush_subnormal_samples = 0;
void rimo_set_flush_subnormal_samples(int on) { g_flush_subnormal_samples = on; }

/* investigation knob (tools/sample_overlap.py): log a key of every integrand sample (harmonic, gamma) of this thread */
static __thread uint64_t *t_sample_log = 0;
static __thread size_t t_sample_cap = 0, t_sample_count = 0;
void rimo_set_sample_log(uint64_t *buf, size_t cap) { t_sample_log = buf; t_sample_cap = cap; t_sample_count = 0; }
size_t rimo_sample_log_count(void) { return t_sample_count; }

/* investigation knob (rimo.h): log of the quadratures of this thread that ended in a GSL error */
static __thread rimo_fail_rec *t_fail_log = 0;
static __thread size_t t_fail_cap = 0, t_fail_count = 0;
void rimo_set_fail_log(rimo_fail_rec *buf, size_t cap) { t_fail_log = buf; t_fail_cap = cap; t_fail_count = 0; }
size_t rimo_fail_log_count(void) { return t_fail_count; }
static void fail_log(double n, double a, double b, double result, double abserr, int lobe, int status, size_t size, int level)
{
    if (t_fail_log && t_fail_count < t_fail_cap) {
        rimo_fail_rec *r = t_fail_log + t_fail_count;
        r->n = n; r->a = a; r->b = b; r->result = result; r->abserr = abserr;
        r->lobe = lobe; r->status = status; r->size = (int) size; r->level = level;
    }
    if (t_fail_log) t_fail_count++;
}

/* investigation knob (tools/lockstep_sim.py): the raw (n, gamma) bit patterns of every sample, in evaluation order */
static __thread uint64_t *t_sample_raw = 0;
static __thread size_t t_raw_cap = 0, t_raw_count = 0;
void rimo_set_sample_raw_log(uint64_t *buf, size_t cap_pairs) { t_sample_raw = buf; t_raw_cap = cap_pairs; t_raw_count = 0; }
size_t rimo_sample_raw_count(void) { return t_raw_count; }

static double gamma_integrand_cb(double g, void *ctx)
{
    sym_state *st = (sym_state *) ctx;
    if (t_sample_raw) {
        if (t_raw_count < t_raw_cap) {
            memcpy(&t_sample_raw[2 * t_raw_count], &st->cur_n, 8);
            memcpy(&t_sample_raw[2 * t_raw_count + 1], &g, 8);
        }
        t_raw_count++;
    }
    if (t_sample_log) {
        uint64_t a, b;
        memcpy(&a, &st->cur_n, 8); memcpy(&b, &g, 8);
        if (t_sample_count < t_sample_cap) t_sample_log[t_sample_count] = a * 0x9E3779B97F4A7C15ull ^ (b + (b << 31) + (b >> 17));
        t_sample_count++;
    }
    const double v = gamma_integrand(st, g, st->cur_n);
    if (g_flush_subnormal_samples && m_fabs(v) < RIM_DBL_MIN) return 0.;
    return v;
}

static double gamma_integral(sym_state *st, double n)
{
    const double s = st->s;
    const double acos_th = m_fabs(st->cos_observer_angle);
    const double sin2 = st->sin_observer_angle * st->sin_observer_angle;
    const double nos = n / s;
    const double root = m_sqrt(nos * nos - sin2);

    const double gamma_minus = (nos - acos_th * root) / sin2;
    const double gamma_plus = (nos + acos_th * root) / sin2;
    const double gamma_peak = 0.5 * (gamma_plus + gamma_minus);

    const double rel_width = (s < 1e6) ? 1. : m_exp(-0.27 * m_log(n) - 0.1);

    const double gamma_minus_high = gamma_peak - (gamma_peak - gamma_minus) * rel_width;
    const double gamma_plus_high = gamma_peak - (gamma_peak - gamma_plus) * rel_width;

    double gamma0, gamma1;
    if (st->stokes == RIMO_STOKES_V) {
        if (!st->negative_lobe) { gamma0 = gamma_peak; gamma1 = gamma_plus_high; }
        else { gamma0 = gamma_minus_high; gamma1 = gamma_peak; }
    } else {
        gamma0 = gamma_minus_high;
        gamma1 = gamma_plus_high;
    }

    double result, abserr;
    uint64_t nev = 0;
    st->cur_n = n;
    int status = rimo_qag(gamma_integrand_cb, st, gamma0, gamma1, 0., g_tuning.epsrel_gamma, 5000, st->gamma_ws,
                          &result, &abserr, &nev);
    if (st->c) {
        st->c->gk_evals += nev;
        st->c->inner_qag_calls++;
        if (st->gamma_ws->size > st->c->max_inner_size) st->c->max_inner_size = st->gamma_ws->size;
    }
    if (status) fail_log(n, gamma0, gamma1, result, abserr, st->negative_lobe, status, st->gamma_ws->size, 0);
    return status ? RIM_NAN : result;
}

static double gamma_integral_cb(double n, void *ctx)
{
    return gamma_integral((sym_state *) ctx, n);
}

/* Err(...) is reported through *failed */
static double n_integration(sym_state *st, double n_start, int *failed)
{
    double ans = 0., contrib = 0., delta_n = 1e5, incr_step_factor = 10.;
    const double DERIV_TOL = 1e-5, TOLERANCE = g_tuning.tail_tolerance;
    rimo_workspace *n_ws = rimo_workspace_alloc(1000);
    int chunks = 0;

    *failed = 0;

    if (st->s < 10.) {
        delta_n = 1.;
        incr_step_factor = 2.;
    }

    while (m_fabs(contrib) >= m_fabs(ans / TOLERANCE)) {
        double deriv, derr;

        if (++chunks > g_tuning.max_chunks) { *failed = 1; break; }

        rimo_deriv_central(gamma_integral_cb, st, n_start, 1e-10 * n_start, &deriv, &derr);
        if (st->c) st->c->deriv_calls++;

        if (deriv == 0. || (contrib != 0. && m_fabs(deriv / contrib) < DERIV_TOL))
            delta_n *= incr_step_factor;

        if (delta_n < n_start / incr_step_factor)
            delta_n *= incr_step_factor;

        {
            double abserr;
            uint64_t nev = 0;
            int status = rimo_qag(gamma_integral_cb, st, n_start, n_start + delta_n, 0., g_tuning.epsrel_n, 1000, n_ws,
                                  &contrib, &abserr, &nev);
            if (st->c) {
                st->c->outer_gk_evals += nev;
                st->c->outer_qag_calls++;
                if (n_ws->size > st->c->max_outer_size) st->c->max_outer_size = n_ws->size;
            }
            if (status) {
                fail_log(n_start, n_start, n_start + delta_n, contrib, abserr, st->negative_lobe, status, n_ws->size, 1);
                *failed = 1;
                break;
            }
        }

        ans += contrib;
        n_start += delta_n;

        if (n_start > 1e13)
            incr_step_factor = 1.;
    }

    rimo_workspace_free(n_ws);
    return ans;
}

/* Rust `x as i64`: truncate toward zero, saturating, NaN -> 0 */
static int64_t sat_i64(double x)
{
    if (!(x == x)) return 0;
    if (x >= 9223372036854775807.0) return INT64_MAX;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t) x;
}

static void sym_init(sym_state *st, const rimo_dist *d, int coeff, int stokes, double s, double theta, rimo_counters *c)
{
    st->d = d;
    st->coeff = coeff;
    st->stokes = stokes;
    st->s = s;
    m_sincos(theta, &st->sin_observer_angle, &st->cos_observer_angle);
    st->negative_lobe = 1;
    st->gamma_ws = rimo_workspace_alloc(5000);
    st->c = c;
    st->cur_n = 0.;
}

double rimo_symphony(const rimo_dist *d, int coeff, int stokes, double s, double theta, rimo_counters *c)
{
    const double N_MAX = (double) g_tuning.n_discrete;
    sym_state st;
    double ans = 0.;
    int failed;

    sym_init(&st, d, coeff, stokes, s, theta, c);

    const double n_minus = s * m_fabs(st.sin_observer_angle);
    const int64_t n_lo = sat_i64(n_minus + 1.);
    const int64_t n_hi = sat_i64(n_minus + 1. + N_MAX);

    for (int64_t n = n_lo; n < n_hi; n++) {
        st.negative_lobe = 0;
        ans += gamma_integral(&st, (double) n);
        if (stokes == RIMO_STOKES_V) {
            st.negative_lobe = 1;
            ans += gamma_integral(&st, (double) n);
        }
    }

    if (!rim_isfinite(ans)) { ans = RIM_NAN; goto done; }

    {
        const double n_start = rim_floor(n_minus + 1. + N_MAX);
        double contrib;

        st.negative_lobe = 0;
        contrib = n_integration(&st, n_start, &failed);
        if (failed) contrib = RIM_NAN;
        ans += contrib;
        if (!rim_isfinite(ans)) { ans = RIM_NAN; goto done; }

        if (stokes == RIMO_STOKES_V) {
            st.negative_lobe = 1;
            contrib = n_integration(&st, n_start, &failed);
            if (failed) contrib = RIM_NAN;
            ans += contrib;
        }
        if (!rim_isfinite(ans)) { ans = RIM_NAN; goto done; }
    }

    {
        const double tpe = TWO_PI * ELECTRON_CHARGE;
        const double acos_th = m_fabs(st.cos_observer_angle);
        if (coeff == RIMO_EMISSION)
            ans = ans * ((tpe * tpe) / (SPEED_LIGHT * acos_th));
        else
            ans = ans * (-1. * (tpe * tpe) / (2. * MASS_ELECTRON * SPEED_LIGHT * acos_th));
    }

done:
    rimo_workspace_free(st.gamma_ws);
    return ans;
}

double rimo_gamma_integrand(const rimo_dist *d, int coeff, int stokes, double s, double theta, double n, double gamma)
{
    sym_state st;
    sym_init(&st, d, coeff, stokes, s, theta, NULL);
    double r = gamma_integrand(&st, gamma, n);
    rimo_workspace_free(st.gamma_ws);
    return r;
}

double rimo_gamma_integral(const rimo_dist *d, int coeff, int stokes, int negative_lobe, double s, double theta, double n)
{
    sym_state st;
    sym_init(&st, d, coeff, stokes, s, theta, NULL);
    st.negative_lobe = negative_lobe;
    double r = gamma_integral(&st, n);
    rimo_workspace_free(st.gamma_ws);
    return r;
}

/* diagnostic_symphony_n_integral (lib.rs:254-260 -> symphony.rs:298-307): one QAG over n in [n_lo, n_hi] of the
 * gamma-integral (limit 1000, eps_rel 1e-3).  Returns the GSL status; *value is only meaningful for 0. */
int rimo_n_integral(const rimo_dist *d, int coeff, int stokes, int negative_lobe, double s, double theta,
                    double n_lo, double n_hi, double *value)
{
    sym_state st;
    sym_init(&st, d, coeff, stokes, s, theta, NULL);
    st.negative_lobe = negative_lobe;
    rimo_workspace *n_ws = rimo_workspace_alloc(1000);
    double abserr;
    uint64_t nev = 0;
    const int status = rimo_qag(gamma_integral_cb, &st, n_lo, n_hi, 0., 1e-3, 1000, n_ws, value, &abserr, &nev);
    rimo_workspace_free(n_ws);
    rimo_workspace_free(st.gamma_ws);
    return status;
}

/* unit seam for gsl::deriv_central as n_integration calls it (symphony.rs:238-240): the derivative estimate of the
 * gamma-integral with respect to n at n_start, h = 1e-10 n_start */
double rimo_symphony_deriv_probe(const rimo_dist *d, int coeff, int stokes, int negative_lobe, double s, double theta, double n_start)
{
    sym_state st;
    double deriv = 0., derr = 0.;
    sym_init(&st, d, coeff, stokes, s, theta, NULL);
    st.negative_lobe = negative_lobe;
    rimo_deriv_central(gamma_integral_cb, &st, n_start, 1e-10 * n_start, &deriv, &derr);
    rimo_workspace_free(st.gamma_ws);
    return deriv;
}

/* diagnostic_symphony_gamma_contribution (lib.rs:288-296 -> symphony.rs:491-567): the contribution of all
 * harmonics n at fixed gamma -- the Symphony double integral with the order of integration reversed. */
typedef struct { sym_state *st; double gamma; } contrib_ctx;
static double contrib_cb(double n, void *ctx)
{
    contrib_ctx *c = (contrib_ctx *) ctx;
    return gamma_integrand(c->st, c->gamma, n);
}

double rimo_gamma_contribution(const rimo_dist *d, int coeff, int stokes, double s, double theta, double gamma)
{
    const int64_t FULLY_DISCRETE_THRESHOLD = 1000, N_DISCRETE = 30;
    sym_state st;
    sym_init(&st, d, coeff, stokes, s, theta, NULL);
    const double delta = m_fabs(st.cos_observer_angle) * m_sqrt(gamma * gamma - 1.);
    const int64_t n_minus = sat_i64(s * (gamma - delta) + 1.);
    const int64_t n_plus = sat_i64(s * (gamma + delta));
    double ans = 0.;
    if (n_plus - n_minus < FULLY_DISCRETE_THRESHOLD) {
        for (int64_t n = n_minus; n < n_plus + 1; n++) ans += gamma_integrand(&st, gamma, (double) n);
    } else {
        for (int64_t n = n_minus; n < n_minus + N_DISCRETE + 1; n++) ans += gamma_integrand(&st, gamma, (double) n);
        contrib_ctx c = { &st, gamma };
        rimo_workspace *ws = rimo_workspace_alloc(5000);
        double contrib, abserr;
        uint64_t nev = 0;
        const int status = rimo_qag(contrib_cb, &c, (double) (n_minus + N_DISCRETE + 1), (double) n_plus, 0., 1e-3, 5000, ws,
                                    &contrib, &abserr, &nev);
        rimo_workspace_free(ws);
        ans += status ? RIM_NAN : contrib;
    }
    rimo_workspace_free(st.gamma_ws);
    if (!rim_isfinite(ans)) return RIM_NAN;
    const double tpe = TWO_PI * ELECTRON_CHARGE;
    const double acos_th = m_fabs(st.cos_observer_angle);
    if (coeff == RIMO_EMISSION) return ans * ((tpe * tpe) / (SPEED_LIGHT * acos_th));
    return ans * (-1. * (tpe * tpe) / (2. * MASS_ELECTRON * SPEED_LIGHT * acos_th));
}

/* ---- lib.rs dispatch and scaling --------------------------------------- */

double rimo_compute_dimensionless(const rimo_dist *d, int coeff, int stokes, double s, double theta, rimo_counters *c)
{
    if (coeff == RIMO_FARADAY) {
        if (stokes == RIMO_STOKES_I) return RIM_NAN;
        return rimo_heyvaerts(d, coeff, stokes, s, theta, c);
    }
    return rimo_symphony(d, coeff, stokes, s, theta, c);
}

static const int SLOT_COEFF[8] = { RIMO_EMISSION, RIMO_ABSORPTION, RIMO_EMISSION, RIMO_ABSORPTION,
                                   RIMO_EMISSION, RIMO_ABSORPTION, RIMO_FARADAY, RIMO_FARADAY };
static const int SLOT_STOKES[8] = { RIMO_STOKES_I, RIMO_STOKES_I, RIMO_STOKES_Q, RIMO_STOKES_Q,
                                    RIMO_STOKES_V, RIMO_STOKES_V, RIMO_STOKES_Q, RIMO_STOKES_V };

void rimo_compute_all_dimensionless(const rimo_dist *d, double s, double theta, double out[8], rimo_counters *c)
{
    for (int k = 0; k < 8; k++)
        out[k] = rimo_compute_dimensionless(d, SLOT_COEFF[k], SLOT_STOKES[k], s, theta, c);
}

double rimo_compute_cgs(const rimo_dist *d, int coeff, int stokes, double nu, double b, double n_e, double theta)
{
    const double nu_c = ELECTRON_CHARGE * b / (TWO_PI * MASS_ELECTRON * SPEED_LIGHT);
    const double val = rimo_compute_dimensionless(d, coeff, stokes, nu / nu_c, theta, NULL);
    if (coeff == RIMO_EMISSION) return val * n_e * nu;
    return val * n_e / nu;
}

/* ---- batch driver -------------------------------------------------------- */

#ifdef _OPENMP
#include <omp.h>
#endif

static void counters_add(rimo_counters *a, const rimo_counters *b)
{
    a->integrand_evals += b->integrand_evals;
    a->gk_evals += b->gk_evals;
    a->inner_qag_calls += b->inner_qag_calls;
    a->outer_gk_evals += b->outer_gk_evals;
    a->outer_qag_calls += b->outer_qag_calls;
    a->deriv_calls += b->deriv_calls;
    if (b->max_inner_size > a->max_inner_size) a->max_inner_size = b->max_inner_size;
    if (b->max_outer_size > a->max_outer_size) a->max_outer_size = b->max_outer_size;
    a->bessel_calls += b->bessel_calls;
    a->norm_evals += b->norm_evals;
    a->hey_nr_samples += b->hey_nr_samples;
    a->hey_qr_i_samples += b->hey_qr_i_samples;
    a->hey_qr_jy_samples += b->hey_qr_jy_samples;
    a->hey_series_terms += b->hey_series_terms;
    a->hey_series_calls += b->hey_series_calls;
}

int rimo_batch(int kind, size_t n, const double *s, const double *theta, const double *const *params,
               uint32_t coeff_mask, double *out, rimo_counters *counters, int nthreads)
{
    static const int NPAR[4] = { 4, 1, 5, 4 };
    if (kind < 0 || kind > 3) return -1;
    const int np = NPAR[kind];
    rimo_counters total;
    memset(&total, 0, sizeof total);
    if (nthreads < 1) nthreads = 1;

#pragma omp parallel num_threads(nthreads)
    {
        rimo_counters local;
        memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 1)
        for (long long i = 0; i < (long long) n; i++) {
            double par[RIMO_MAX_PARAMS];
            rimo_dist d;
            for (int k = 0; k < np; k++) par[k] = params[k][i];
            int st = rimo_dist_init(&d, kind, par);
            for (int k = 0; k < 8; k++) {
                double v = RIM_NAN;
                if (coeff_mask & (1u << k)) {
                    if (!st)
                        v = rimo_compute_dimensionless(&d, SLOT_COEFF[k], SLOT_STOKES[k], s[i], theta[i],
                                                       counters ? &local : NULL);
                }
                out[i * 8 + k] = v;
            }
        }
#pragma omp critical
        counters_add(&total, &local);
    }
    if (counters) *counters = total;
    return 0;
}

/* ---- QAG self-test integrands (+ - * / sqrt only): same families as the
 * rimphony_qag_selftest_device entry point of the product ------------------- */

typedef struct { int family; double p0, p1; } selftest_ctx;

static double selftest_integrand(double x, void *ctx)
{
    const selftest_ctx *c = (const selftest_ctx *) ctx;
    switch (c->family) {
    case 0: { const double u = (x - c->p0) * c->p1; return 1. / (1. + u * u); }
    case 1: return m_sqrt(m_fabs(x - c->p0)) * c->p1;
    case 2: return x * x * (c->p0 + x * c->p1);
    case 3: return 1. / m_sqrt(m_fabs(x - c->p0) + c->p1);
    default: { const double u = x * c->p0; return c->p1 * m_fabs((u - rim_floor(u)) - 0.5); }
    }
}

int rimo_qag_selftest(int family, double p0, double p1, double a, double b, double epsabs, double epsrel,
                      size_t limit, double *result, double *abserr, size_t *size_out)
{
    selftest_ctx c = { family, p0, p1 };
    return rimo_qag_gk31(selftest_integrand, &c, a, b, epsabs, epsrel, limit, result, abserr, size_out, NULL);
}

/* full_calculation() alone, batched (for parity tests of the norm kernel) */
int rimo_batch_norm(int kind, size_t n, const double *const *params, double *norm)
{
    static const int NPAR[4] = { 4, 1, 5, 4 };
    if (kind < 0 || kind > 3) return -1;
    for (size_t i = 0; i < n; i++) {
        double par[RIMO_MAX_PARAMS];
        rimo_dist d;
        for (int k = 0; k < NPAR[kind]; k++) par[k] = params[k][i];
        int st = rimo_dist_init(&d, kind, par);
        norm[i] = st ? RIM_NAN : d.norm;
    }
    return 0;
}
