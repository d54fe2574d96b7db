/* rimo_heyvaerts.c -- oracle restatement of src/heyvaerts.rs (TEST INFRASTRUCTURE
 * ONLY, see rimo.h): Faraday conversion rho_Q ("h") and rotation rho_V ("f") in
 * the Heyvaerts et al. (2013) formalism.
 *
 *   compute_dimensionless / CalculationState::compute   heyvaerts.rs:60-191
 *   fill_coord_vars                                     heyvaerts.rs:194-201
 *   nr_outer_integral / nr_outer_integrand              heyvaerts.rs:204-250
 *   qr_outer_integral / qr_outer_integrand              heyvaerts.rs:253-296
 *   h_qr / h_nr / f_qr / f_nr elements                  heyvaerts.rs:302-468
 *   dfdsigma                                            heyvaerts.rs:472-493
 *
 * Third-party arithmetic: the Bessel functions I_{+-1/3}, I_{+-2/3}, J_nu, Y_nu come
 * from the un-vendored `special-fun` crate (Cargo.toml:20; believed to bind Cephes
 * iv/jv/yv -- source unavailable, version unpinned).  They are restated here from
 * their defining ascending series, which is all this path needs:
 *   - the I branch is taken for g < 10 (heyvaerts.rs:33,330,358,434);
 *   - the J/Y branch (g >= 10) is only reachable when the quasi-resonant pomega
 *     range is limited by sqrt(sigma^2 - sigma0^2) rather than by the physical
 *     limit, i.e. for sigma < 3^{3/2}/... ~ 3 (heyvaerts.rs:263-265), so orders lie in
 *     (-1, 3) and x is small: the ascending series converge in a few terms.
 * Y_nu uses the reflection formula (as Cephes yv does for non-integer order).
 * These functions are pinned only through the four 1 % Faraday known answers of
 * the reference (power_law.rs:209-240, thermal_juettner.rs:174-210) and checked
 * against scipy in tests/test_oracle_heyvaerts.py.
 */
#include <stdlib.h>
#include "rimo.h"
#include "rimo_math.h"

#define TWO_PI (2. * RIM_PI)
#define MASS_ELECTRON 9.1093826e-28
#define SPEED_LIGHT 2.99792458e10
#define ELECTRON_CHARGE 4.80320680e-10

/* heyvaerts.rs:28-33 */
static const double FOUR_OVER_SQRT_27 = 0.769800358919501;
static const double INVERSE_C = 1. / SPEED_LIGHT;
static const double INVERSE_SQRT_3 = 0.5773502691896257;
static const double SQRT_8_OVER_3 = 0.9428090415820635;
static const double THREE_TWO_THIRDS = 2.080083823051904;
static const double G_APPROXIMATION_CUTOFF = 10.;

#define HEY_MAX_STEPS 4096   /* cap on each chunk-marching loop (unbounded in the reference) */

/* Rust f64::min / f64::max: a NaN operand is ignored */
static double rust_min(double a, double b) { if (a != a) return b; if (b != b) return a; return a < b ? a : b; }
static double rust_max(double a, double b) { if (a != a) return b; if (b != b) return a; return a > b ? a : b; }

/* ---- special functions --------------------------------------------------- */

/* Gamma(z) for real z.  Deterministic flavour (shared with the kernels' gamma_real, dev_heyvaerts.h): shift above 16,
 * Stirling, divide back (poles give +-inf).  Literal flavour: the C library's long double Gamma function rounded to
 * double -- the stand-in for the Gamma function inside special-fun's Bessel routines, whose source is absent; a library
 * Gamma is good to an ulp or two.  (Until round 3 the literal flavour formed exp(lgamma(w)) / prod, whose relative error
 * is |lgamma(w)| ulps -- 15 to 27 ulp at the four arguments 1 + nu the quasi-resonant elements need, always with the
 * same sign; the difference I_(-nu) - I_nu of heyvaerts.rs:337, 365 amplifies that by up to 3e8 at g -> 10, and
 * tools/faraday_tail_attribution.py showed that this one quantity owned the whole tail of rho_Q / rho_V beyond 1e-6
 * between the kernels and the literal vectors: profiles/r4_faraday_tail_attribution.txt.) */
double rimo_gamma_real(double z)
{
    if (RIMO_LIT(RIMO_ATTR_ELEM))
        return (double) tgammal((long double) z);
    double prod = 1., w = z;
    int guard = 0;
    while (w < 16. && guard < 64) { prod = prod * w; w = w + 1.; guard++; }
    return rim_exp(rim_lgamma_stirling(w)) / prod;
}

/* work counters of the calling thread's Heyvaerts evaluation (null outside one) */
static __thread rimo_counters *t_ctr = 0;

/* sum_k (sign q)^k / (k! (nu+1)_k),  q = (x/2)^2 */
static double ascending_series(double nu, double q, double sign)
{
    double term = 1., sum = 1.;
    int k;
    for (k = 1; k <= 500; k++) {
        term = term * (sign * q / (k * (k + nu)));
        sum = sum + term;
        if (m_fabs(term) < 1e-17 * m_fabs(sum)) break;
    }
    if (t_ctr) { t_ctr->hey_series_terms += (uint64_t) (k <= 500 ? k : 500); t_ctr->hey_series_calls += 1; }
    return sum;
}

/* the same series for the four fixed orders of the quasi-resonant elements as the kernels sum it (dev_heyvaerts.h
 * rim_iseries4): Horner's rule on the correctly rounded coefficients 1 / (k! (nu + 1)_k) (iseries_table.h), from a
 * degree chosen by q alone */
#include "../rimphony_amd/csrc/iseries_table.h"
static const double ISERIES[4 * (RIM_ISERIES_NMAX + 1)] = RIM_ISERIES_TABLE;
static double iseries_horner(int j, double q)
{
    static const double QMIN[6] = { RIM_ISERIES_Q21, RIM_ISERIES_Q18, RIM_ISERIES_Q15, RIM_ISERIES_Q12, RIM_ISERIES_Q9, RIM_ISERIES_Q6 };
    double s = 0.;
    int terms = 7;
    for (int b = 0; b < 6; b++) {          /* blocks of three terms, degree 24 downwards */
        if (q >= QMIN[b]) {
            const int k = 24 - 3 * b;
            s = rim_fma(s, q, ISERIES[4 * k + j]);
            s = rim_fma(s, q, ISERIES[4 * (k - 1) + j]);
            s = rim_fma(s, q, ISERIES[4 * (k - 2) + j]);
            terms += 3;
        }
    }
    for (int k = 6; k >= 0; k--) s = rim_fma(s, q, ISERIES[4 * k + j]);
    if (t_ctr) { t_ctr->hey_series_terms += (uint64_t) terms; t_ctr->hey_series_calls += 1; }
    return s;
}

double rimo_bessel_i(double nu, double x)   /* receiver.besseli(nu): I_nu(x), x > 0 */
{
    const double h = 0.5 * x;
    /* the four orders of the quasi-resonant elements: the deterministic flavour takes (x/2)^nu from one cube root and
     * 1 / Gamma(1 + nu) as a constant, and sums the series by Horner's rule, as the kernels do (rim_third_powers,
     * detmath.h; dev_heyvaerts.h rim_iseries4); the two are separate switches of the attribution build */
    static const double ORD[4] = { 2. / 3., -2. / 3., 1. / 3., -1. / 3. };
    static const double RGAM[4] = { RIM_RGAMMA_5_3, RIM_RGAMMA_1_3, RIM_RGAMMA_4_3, RIM_RGAMMA_2_3 };
    for (int j = 0; j < 4; j++)
        if (nu == ORD[j]) {
            double prefactor;
            if (RIMO_LIT(RIMO_ATTR_THIRD)) {
                prefactor = m_pow(h, nu) / rimo_gamma_real(nu + 1.);
            } else {
                double pref[4];
                rim_third_powers(h, pref);
                prefactor = pref[j] * RGAM[j];
            }
            return prefactor * (RIMO_LIT(RIMO_ATTR_SERIES) ? ascending_series(nu, h * h, 1.) : iseries_horner(j, h * h));
        }
    return m_pow(h, nu) / rimo_gamma_real(nu + 1.) * ascending_series(nu, h * h, 1.);
}

double rimo_bessel_jnu(double nu, double x)  /* J_nu(x), x > 0, small x */
{
    const double h = 0.5 * x;
    if (!RIMO_LIT(RIMO_ATTR_RGAMMA)) {
        /* the deterministic flavour multiplies by 1 / Gamma from the short-range series where it applies, as the kernels do */
        const double z = nu + 1.;
        const double rg = RIM_RGAMMA_NEAR(z) ? rim_rgamma_near(z) : 1. / rimo_gamma_real(z);
        return m_pow(h, nu) * rg * ascending_series(nu, h * h, -1.);
    }
    return m_pow(h, nu) / rimo_gamma_real(nu + 1.) * ascending_series(nu, h * h, -1.);
}

double rimo_bessel_ynu(double nu, double x)  /* Y_nu(x) by reflection */
{
    double sn, cs;
    /* exact integer orders: the reflection formula is 0/0; step off by 2^-26 relative (documented) */
    if (nu == rim_floor(nu)) nu = nu + 1.4901161193847656e-08 * (m_fabs(nu) > 1. ? m_fabs(nu) : 1.);
    m_sincos(RIM_PI * nu, &sn, &cs);
    return (cs * rimo_bessel_jnu(nu, x) - rimo_bessel_jnu(-nu, x)) / sn;
}

/* J_sigma, Y_sigma, J_(sigma-1), Y_(sigma-1) at x for a non-integer 0 < sigma < 8 as the kernels evaluate them on the J/Y
 * branch since round 4 (dev_heyvaerts.h bessel_jy_fast, operation for operation): one power h^sigma and two reciprocals for
 * the four prefactors, rim_rgamma_quad for the four 1 / Gamma, one sincos, the four ascending series in one loop as
 * numerator / denominator pairs with one division each at the end. */
static void jy_fast(double sigma, double x, int want_ym1, double *js, double *ys, double *jm1, double *ym1)
{
    const double h = 0.5 * x, q = h * h;
    double ll;
    const double lh = rim_log_dd(h, &ll);
    const double p = rim_pow_from_log(lh, ll, sigma);
    const double hinv = 1. / h, pinv = 1. / p;
    double rg[4];
    rim_rgamma_quad(sigma, rg);
    double sn, cs;
    rim_sincos(RIM_PI * sigma, &sn, &cs);
    const double nu0 = sigma, nu1 = sigma - 1., nu2 = -sigma, nu3 = 1. - sigma;
    double n0 = 1., n1 = 1., n2 = 1., n3 = 1., d0 = 1., d1 = 1., d2 = 1., d3 = 1., pw = 1.;
    const double mq = -q;
    int k;
    for (k = 1; k <= 80; k++) {
        const double kd = (double) k;
        pw = pw * mq;
        const double e0 = kd * (kd + nu0), e1 = kd * (kd + nu1), e2 = kd * (kd + nu2), e3 = kd * (kd + nu3);
        n0 = rim_fma(n0, e0, pw); d0 = d0 * e0;
        n1 = rim_fma(n1, e1, pw); d1 = d1 * e1;
        n2 = rim_fma(n2, e2, pw); d2 = d2 * e2;
        n3 = rim_fma(n3, e3, pw); d3 = d3 * e3;
        const double t = m_fabs(pw);
        if (t < 1e-17 * m_fabs(n0) && t < 1e-17 * m_fabs(n1) && t < 1e-17 * m_fabs(n2) && t < 1e-17 * m_fabs(n3)) break;
    }
    if (t_ctr) { t_ctr->hey_series_terms += (uint64_t) (4 * (k <= 80 ? k : 80)); t_ctr->hey_series_calls += 4; }
    const double j_s = (p * rg[0]) * (n0 / d0);
    const double j_sm1 = ((p * hinv) * rg[1]) * (n1 / d1);
    const double j_ms = (pinv * rg[3]) * (n2 / d2);
    *js = j_s;
    *jm1 = j_sm1;
    *ys = (cs * j_s - j_ms) / sn;
    if (want_ym1) {
        const double j_1ms = ((h * pinv) * rg[2]) * (n3 / d3);
        *ym1 = (cs * j_sm1 + j_1ms) / sn;
    } else {
        *ym1 = 0.;
    }
}

/* the Bessel values of the J/Y branch of a quasi-resonant element (heyvaerts.rs:335-336, 359-363, 437-441) */
static void jy_set(double sigma, double x, int want_ym1, double *js, double *ys, double *jm1, double *ym1)
{
    if (!RIMO_LIT(RIMO_ATTR_JYFAST) && sigma > 0. && sigma < 8. && sigma != rim_floor(sigma)) {
        jy_fast(sigma, x, want_ym1, js, ys, jm1, ym1);
        return;
    }
    *js = rimo_bessel_jnu(sigma, x);
    *jm1 = rimo_bessel_jnu(sigma - 1., x);
    *ys = rimo_bessel_ynu(sigma, x);
    *ym1 = want_ym1 ? rimo_bessel_ynu(sigma - 1., x) : 0.;
}

/* test seam: { J_sigma, Y_sigma, J_(sigma-1), Y_(sigma-1) } at x as the elements of this flavour obtain them */
void rimo_bessel_jy_set(double sigma, double x, double out[4]) { jy_set(sigma, x, 1, out, out + 1, out + 2, out + 3); }

/* ---- the calculation ------------------------------------------------------- */

typedef struct {
    const rimo_dist *d;
    int stokes;
    double s, cos_observer_angle, sin_observer_angle, sigma0, sigma0_sq;
    double dinv;             /* 1 / (sigma0 sin(theta)): the deterministic flavour multiplies by it, as the kernels do */
    /* set inside integrands */
    double sigma, pomega, x, gamma, mu;
    /* inner integral context */
    double fixed;            /* pomega (NR) or sigma (QR) held fixed by the inner integral */
    rimo_workspace *ows, *iws;
    rimo_counters *c;
} hey_state;

static void fill_coord_vars(hey_state *st, double sigma, double pomega)
{
    st->sigma = sigma;
    st->pomega = pomega;
    st->x = m_sqrt(sigma * sigma - pomega * pomega - st->sigma0_sq);
    if (RIMO_LIT(RIMO_ATTR_DINV))
        st->gamma = (sigma - pomega * st->cos_observer_angle) / (st->sigma0 * st->sin_observer_angle);
    else
        st->gamma = (sigma - pomega * st->cos_observer_angle) * st->dinv;
    st->mu = (sigma * st->cos_observer_angle - pomega)
        / (st->sigma0 * st->sin_observer_angle * m_sqrt(st->gamma * st->gamma - 1.));
}

/* g of the quasi-resonant elements (heyvaerts.rs:309, 406) */
static double qr_g(const hey_state *st)
{
    if (RIMO_LIT(RIMO_ATTR_GFORM))
        return SQRT_8_OVER_3 * m_pow15(st->sigma - st->x) / m_sqrt(st->x);
    /* the deterministic flavour, as the kernels (dev_heyvaerts.h hey_shared): one root, and the quotient (sigma - x) / x */
    return SQRT_8_OVER_3 * (st->sigma - st->x) * m_sqrt((st->sigma - st->x) / st->x);
}

static double dfdsigma(const hey_state *st)
{
    double dfdg, dfdcxi;
    rimo_calc_f_derivatives(st->d, st->gamma, st->mu, &dfdg, &dfdcxi);

    const double g_term = RIMO_LIT(RIMO_ATTR_DINV) ? dfdg / (st->sigma0 * st->sin_observer_angle) : dfdg * st->dinv;
    double mu_term;
    if (dfdcxi == 0.) {
        mu_term = 0.;
    } else {
        const double q = st->sigma - st->pomega * st->cos_observer_angle;
        const double r = st->pomega - st->sigma * st->cos_observer_angle;
        const double t = st->sigma0 * st->sin_observer_angle;
        const double u = q * q - t * t;
        const double dcxi_dsigma = (q * u * st->cos_observer_angle + u * r + r * (t * t))
            / (m_pow15(u) * q);
        mu_term = dcxi_dsigma * dfdcxi;
    }
    return g_term + mu_term;
}

static double h_qr_element(const hey_state *st)
{
    const double po_sq = st->pomega * st->pomega;
    const double smxox = (st->sigma - st->x) / st->x;
    const double g = qr_g(st);
    double y;

    double js = 0., ys = 0., jm1 = 0., ym1 = 0.;
    if (!(g < G_APPROXIMATION_CUTOFF)) jy_set(st->sigma, st->x, 1, &js, &ys, &jm1, &ym1);
    if (g < G_APPROXIMATION_CUTOFF) {
        const double plus = rimo_bessel_i(2. / 3., g);
        const double minus = rimo_bessel_i(-2. / 3., g);
        y = FOUR_OVER_SQRT_27 * (smxox * smxox) * (minus - plus) * (minus + plus);
    } else {
        const double jvp = jm1 - st->sigma * js / st->x;
        const double yvp = ym1 - st->sigma * ys / st->x;
        y = jvp * yvp;
    }
    const double t1 = RIM_PI * RIM_PI * (st->x * st->x) * y;

    if (g < G_APPROXIMATION_CUTOFF) {
        const double plus = rimo_bessel_i(1. / 3., g);
        const double minus = rimo_bessel_i(-1. / 3., g);
        y = 0.5 * FOUR_OVER_SQRT_27 * smxox * (minus - plus) * (minus + plus);
    } else {
        y = -js * ys;
    }
    const double t2 = RIM_PI * RIM_PI * (st->pomega * st->pomega) * y;

    const double t3 = -RIM_PI * (2. * po_sq + st->sigma0_sq) / m_sqrt(po_sq + st->sigma0_sq);

    const double dfds = dfdsigma(st);
    return INVERSE_C * (t1 + t2 + t3) * dfds;
}

static double h_nr_element(const hey_state *st)
{
    const double s_sq = st->sigma * st->sigma;
    const double x_sq = st->x * st->x;
    const double ssqmxsq = s_sq - x_sq;
    const double ratio = s_sq / ssqmxsq;
    const double a1 = 1. / 8. - 5. / 24. * s_sq / ssqmxsq;
    const double a2 = 3. / 128. - 77. / 576. * s_sq / ssqmxsq + 385. / 3456. * (ratio * ratio);
    const double xa1p = -5. / 12. * s_sq * x_sq / (ssqmxsq * ssqmxsq);
    const double t1 = (6. * a2 - a1 * a1 + xa1p) / m_sqrt(ssqmxsq) + a1 * x_sq / m_pow15(ssqmxsq)
        - (x_sq * x_sq) / m_pow25(ssqmxsq) / 8.;
    const double t2 = (6. * a2 - a1 * a1) / m_pow15(ssqmxsq);
    const double u1 = 2. * t1 - st->sigma0_sq * t2;
    const double dfds = dfdsigma(st);
    return RIM_PI * INVERSE_C * u1 * dfds;
}

static double f_qr_element(const hey_state *st)
{
    const double g = qr_g(st);
    double y;
    if (g < G_APPROXIMATION_CUTOFF) {
        y = INVERSE_SQRT_3
            * g
            * (rimo_bessel_i(-2. / 3., g) - rimo_bessel_i(2. / 3., g))
            * (rimo_bessel_i(-1. / 3., g) + rimo_bessel_i(1. / 3., g));
    } else {
        double js, ys, jm1, ym1;
        jy_set(st->sigma, st->x, 0, &js, &ys, &jm1, &ym1);
        const double jvp = jm1 - st->sigma * js / st->x;
        y = -st->x * jvp * ys;
    }
    const double dfds = dfdsigma(st);
    return -TWO_PI * INVERSE_C * st->pomega * (RIM_PI * y - 1.) * dfds;
}

static double f_nr_element(const hey_state *st)
{
    const double s_sq = st->sigma * st->sigma;
    const double x_sq = st->x * st->x;
    const double ssqmxsq = s_sq - x_sq;
    const double ratio = s_sq / ssqmxsq;
    const double a1 = 1. / 8. - 5. / 24. * s_sq / ssqmxsq;
    const double a2 = 3. / 128. - 77. / 576. * s_sq / ssqmxsq + 385. / 3456. * (ratio * ratio);
    const double xa1p = -5. / 12. * s_sq * x_sq / (ssqmxsq * ssqmxsq);
    const double z =
        0.5 * x_sq / m_pow15(ssqmxsq)
        + (6. * a2 + xa1p - a1 * a1) / ssqmxsq
        + 1.5 * a1 * x_sq / (ssqmxsq * ssqmxsq);
    const double dfds = dfdsigma(st);
    return -2. * RIM_PI * INVERSE_C * z * st->pomega * dfds;
}

/* The deterministic flavour of the two non-resonant elements, operation for operation what the kernels evaluate
 * (dev_heyvaerts.h nr_common / h_nr_element / f_nr_element): every division by a power of sigma^2 - x^2 as a product of
 * powers of u = 1 / sqrt(sigma^2 - x^2).  h_nr_element / f_nr_element above stay the literal restatement (the libm
 * flavour runs them; tools/check_literals.py compares both forms with heyvaerts.rs:379-394, 453-468). */
static void nr_common_det(const hey_state *st, double *a1, double *a2, double *xa1p, double *x_sq, double *u, double *u2)
{
    const double s_sq = st->sigma * st->sigma;
    *x_sq = st->x * st->x;
    *u = 1. / m_sqrt(s_sq - *x_sq);
    *u2 = *u * *u;
    const double ratio = s_sq * *u2;
    *a1 = 1. / 8. - 5. / 24. * ratio;
    *a2 = 3. / 128. - 77. / 576. * ratio + 385. / 3456. * (ratio * ratio);
    *xa1p = -5. / 12. * (s_sq * *x_sq) * (*u2 * *u2);
}

static double h_nr_element_det(const hey_state *st)
{
    double a1, a2, xa1p, x_sq, u, u2;
    nr_common_det(st, &a1, &a2, &xa1p, &x_sq, &u, &u2);
    const double u3 = u2 * u;
    const double t1 = (6. * a2 - a1 * a1 + xa1p) * u + a1 * x_sq * u3
        - (x_sq * x_sq) * (u3 * u2) * 0.125;
    const double t2 = (6. * a2 - a1 * a1) * u3;
    const double u1 = 2. * t1 - st->sigma0_sq * t2;
    const double dfds = dfdsigma(st);
    return RIM_PI * INVERSE_C * u1 * dfds;
}

static double f_nr_element_det(const hey_state *st)
{
    double a1, a2, xa1p, x_sq, u, u2;
    nr_common_det(st, &a1, &a2, &xa1p, &x_sq, &u, &u2);
    const double z =
        0.5 * x_sq * (u2 * u)
        + (6. * a2 + xa1p - a1 * a1) * u2
        + 1.5 * a1 * x_sq * (u2 * u2);
    const double dfds = dfdsigma(st);
    return -2. * RIM_PI * INVERSE_C * z * st->pomega * dfds;
}

/* inner integrands: the integration variable is sigma (NR) or pomega (QR) */
static double nr_inner_cb(double sigma, void *ctx)
{
    hey_state *st = (hey_state *) ctx;
    fill_coord_vars(st, sigma, st->fixed);
    if (st->c) { st->c->integrand_evals++; st->c->hey_nr_samples++; }
    if (RIMO_LIT(RIMO_ATTR_NR))
        return st->stokes == RIMO_STOKES_Q ? h_nr_element(st) : f_nr_element(st);
    return st->stokes == RIMO_STOKES_Q ? h_nr_element_det(st) : f_nr_element_det(st);
}

static double qr_inner_cb(double pomega, void *ctx)
{
    hey_state *st = (hey_state *) ctx;
    fill_coord_vars(st, st->fixed, pomega);
    if (st->c) {
        st->c->integrand_evals++;
        const double g = qr_g(st);
        if (g < G_APPROXIMATION_CUTOFF) st->c->hey_qr_i_samples++; else st->c->hey_qr_jy_samples++;
    }
    return st->stokes == RIMO_STOKES_Q ? h_qr_element(st) : f_qr_element(st);
}

static double inner_qag(hey_state *st, rimo_fn f, double a, double b)
{
    double result, abserr;
    uint64_t nev = 0;
    int status = rimo_qag(f, st, a, b, 0., 1e-3, 4096, st->iws, &result, &abserr, &nev);
    if (st->c) {
        st->c->gk_evals += nev;
        st->c->inner_qag_calls++;
        if (st->iws->size > st->c->max_inner_size) st->c->max_inner_size = st->iws->size;
    }
    return status ? RIM_NAN : result;
}

static double nr_outer_integrand(double pomega, void *ctx)
{
    hey_state *st = (hey_state *) ctx;
    const double sigma_min = m_sqrt(pomega * pomega + st->sigma0_sq);
    const double sigma_max = INVERSE_SQRT_3 * m_pow15(sigma_min);
    if (sigma_max <= sigma_min)
        return 0.;
    st->fixed = pomega;
    return inner_qag(st, nr_inner_cb, sigma_min, sigma_max);
}

static double qr_outer_integrand(double sigma, void *ctx)
{
    hey_state *st = (hey_state *) ctx;
    const double pomega_max_phys = m_sqrt(THREE_TWO_THIRDS * m_pow43(sigma) - st->sigma0_sq);
    const double pomega_max_qr = m_sqrt(sigma * sigma - st->sigma0_sq);
    const double pomega_max = rust_min(pomega_max_phys, pomega_max_qr);
    st->fixed = sigma;
    return inner_qag(st, qr_inner_cb, -pomega_max, pomega_max);
}

static double outer_integral(hey_state *st, rimo_fn f, double a, double b)
{
    double result, abserr;
    uint64_t nev = 0;
    int status = rimo_qag(f, st, a, b, 0., 1e-3, 4096, st->ows, &result, &abserr, &nev);
    if (st->c) {
        st->c->outer_gk_evals += nev;
        st->c->outer_qag_calls++;
        if (st->ows->size > st->c->max_outer_size) st->c->max_outer_size = st->ows->size;
    }
    return status ? RIM_NAN : result;
}

static double deriv_of(hey_state *st, rimo_fn f, double x)
{
    double r, e;
    rimo_deriv_central(f, st, x, 1e-6, &r, &e);
    if (st->c) st->c->deriv_calls++;
    return r;
}

/* The quasi-resonant marching loop (heyvaerts.rs:156-185) never tests keep_going while qr_val is 0: when every chunk
 * contributes exactly 0 it marches for ever.  For the thermal distribution that happens to cold, high-frequency points:
 * d f / d gamma = norm exp(-gamma / T) (-1 / T) has underflowed to exactly 0 and gamma only grows.  The step cap of this
 * restatement turns "never returns" into NaN; where the loop is PROVABLY endless the cap is applied at once, as the kernels
 * do (dev_heyvaerts.h hey_point_endless / hey_qr_is_endless -- the same test, both flavours): every sample of every later
 * chunk has gamma >= (sigma - 3^(1/3) sigma^(2/3)) / (sigma0 sin theta) >= 760 T, and exp() is exactly 0 below -745.2. */
static int qr_is_endless(const hey_state *st, double sigma_low)
{
    if (st->d->kind != RIMO_THERMAL_JUETTNER) return 0;
    const double g0 = -760. / st->d->neg_inverse_t;
    if (!(g0 > 0. && g0 < 1e300) || !(sigma_low >= 1.)) return 0;
    const double c = rim_cbrt_normal(sigma_low);
    return (sigma_low - 1.4422495703074083 * (c * c)) * st->dinv >= g0;
}

double rimo_heyvaerts(const rimo_dist *d, int coeff, int stokes, double s, double theta, rimo_counters *c)
{
    hey_state st;
    const double TOL = 1e-5, DELTA_SCALE_FACTOR = 5.;
    double result = RIM_NAN;
    int steps;

    (void) coeff;
    if (stokes != RIMO_STOKES_Q && stokes != RIMO_STOKES_V) return RIM_NAN;
    rimo_tuning tuning;
    rimo_get_tuning(&tuning);
    const int hey_max_steps = tuning.hey_max_steps;

    st.d = d;
    st.stokes = stokes;
    st.s = s;
    m_sincos(theta, &st.sin_observer_angle, &st.cos_observer_angle);
    st.sigma0 = s * st.sin_observer_angle;
    st.sigma0_sq = st.sigma0 * st.sigma0;
    st.dinv = 1. / (st.sigma0 * st.sin_observer_angle);
    st.sigma = st.pomega = st.x = st.gamma = st.mu = RIM_NAN;
    st.fixed = RIM_NAN;
    st.ows = rimo_workspace_alloc(4096);
    st.iws = rimo_workspace_alloc(4096);
    st.c = c;
    t_ctr = c;

    double pomega_left = -3. * st.sigma0;
    double pomega_right = 3. * st.sigma0;
    double delta_left = pomega_right;
    double delta_right = pomega_right;
    int keep_going = 1;

    double nr_val = outer_integral(&st, nr_outer_integrand, pomega_left, pomega_right);
    if (rim_isnan(nr_val)) goto done;

    steps = 0;
    while (keep_going) {
        if (++steps > hey_max_steps) goto done;
        if (nr_val != 0.) {
            const double rel_deriv = deriv_of(&st, nr_outer_integrand, pomega_right);
            if (rel_deriv == 0. || m_fabs(1. / (rel_deriv * delta_right)) > DELTA_SCALE_FACTOR)
                delta_right *= DELTA_SCALE_FACTOR;
        }
        const double contrib = outer_integral(&st, nr_outer_integrand, pomega_right, pomega_right + delta_right);
        if (rim_isnan(contrib)) goto done;
        if (nr_val != 0.)
            keep_going = m_fabs(contrib / nr_val) > TOL;
        nr_val += contrib;
        pomega_right += delta_right;
    }

    keep_going = 1;
    steps = 0;
    while (keep_going) {
        if (++steps > hey_max_steps) goto done;
        const double rel_deriv = deriv_of(&st, nr_outer_integrand, pomega_left);
        if (rel_deriv == 0. || m_fabs(1. / (rel_deriv * delta_left)) > DELTA_SCALE_FACTOR)
            delta_left *= DELTA_SCALE_FACTOR;
        const double contrib = outer_integral(&st, nr_outer_integrand, pomega_left - delta_left, pomega_left);
        if (rim_isnan(contrib)) goto done;
        keep_going = m_fabs(contrib / nr_val) > TOL;
        nr_val += contrib;
        pomega_left -= delta_left;
    }

    {
        double qr_val = 0.;
        double sigma_low = rust_max(st.sigma0, INVERSE_SQRT_3 * m_pow(st.sigma0, 1.5));
        double delta_sigma = st.sigma0;
        keep_going = 1;
        steps = 0;
        while (keep_going) {
            if (++steps > hey_max_steps) goto done;
            if (qr_val != 0.) {
                const double rel_deriv = deriv_of(&st, qr_outer_integrand, sigma_low);
                if (rel_deriv == 0. || m_fabs(1. / (rel_deriv * delta_sigma)) > DELTA_SCALE_FACTOR) {
                    if (delta_sigma < 1e6 * st.sigma0)
                        delta_sigma *= DELTA_SCALE_FACTOR;
                }
            }
            else if (qr_is_endless(&st, sigma_low)) {
                goto done;          /* provably endless: the step cap at once (see qr_is_endless) */
            }
            const double contrib = outer_integral(&st, qr_outer_integrand, sigma_low, sigma_low + delta_sigma);
            if (rim_isnan(contrib)) goto done;
            if (qr_val != 0.)
                keep_going = m_fabs(contrib / qr_val) > TOL;
            qr_val += contrib;
            sigma_low += delta_sigma;
        }

        const double ssin = s * st.sin_observer_angle;
        result = 2. * (ELECTRON_CHARGE * ELECTRON_CHARGE) * (nr_val + qr_val) / (MASS_ELECTRON * (ssin * ssin));
    }

done:
    t_ctr = 0;
    rimo_workspace_free(st.ows);
    rimo_workspace_free(st.iws);
    return result;
}

/* diagnostic: one outer-integrand sample (an inner integral) */
double rimo_hey_outer_integrand(const rimo_dist *d, int stokes, double s, double theta, int qr, double u)
{
    hey_state st;
    st.d = d;
    st.stokes = stokes;
    st.s = s;
    m_sincos(theta, &st.sin_observer_angle, &st.cos_observer_angle);
    st.sigma0 = s * st.sin_observer_angle;
    st.sigma0_sq = st.sigma0 * st.sigma0;
    st.dinv = 1. / (st.sigma0 * st.sin_observer_angle);
    st.sigma = st.pomega = st.x = st.gamma = st.mu = RIM_NAN;
    st.fixed = RIM_NAN;
    st.ows = NULL;
    st.iws = rimo_workspace_alloc(4096);
    st.c = NULL;
    const double v = qr ? qr_outer_integrand(u, &st) : nr_outer_integrand(u, &st);
    rimo_workspace_free(st.iws);
    return v;
}

/* diagnostic: one inner-integrand sample (qr: quasi-resonant element at fixed sigma, variable
 * pomega; else non-resonant element at fixed pomega, variable sigma) */
double rimo_hey_element(const rimo_dist *d, int stokes, double s, double theta, int qr, double fixed, double v)
{
    hey_state st;
    st.d = d;
    st.stokes = stokes;
    st.s = s;
    m_sincos(theta, &st.sin_observer_angle, &st.cos_observer_angle);
    st.sigma0 = s * st.sin_observer_angle;
    st.sigma0_sq = st.sigma0 * st.sigma0;
    st.dinv = 1. / (st.sigma0 * st.sin_observer_angle);
    st.fixed = fixed;
    st.c = NULL;
    return qr ? qr_inner_cb(v, &st) : nr_inner_cb(v, &st);
}
