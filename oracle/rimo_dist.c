/* rimo_dist.c -- oracle restatement of the four electron distribution
 * functions and their per-instance normalisation (TEST INFRASTRUCTURE ONLY).
 *
 *   power_law.rs:36-111, thermal_juettner.rs:29-72, pitchy_pl.rs:32-116,
 *   pitchy_kappa.rs:38-126.
 *
 * One deviation, recorded in DESIGN.md: the thermal normalisation is a
 * semi-infinite gsl_integration_qagiu (QAGS + epsilon extrapolation,
 * thermal_juettner.rs:58-62, eps_rel 1e-5).  GSL is absent; the same integral
 * int_1^inf g sqrt(g^2-1) exp(-g/T) dg is evaluated here with the QAG/GK31
 * machinery already on the path after the substitution g = 1 + u^2 (removes
 * the square-root endpoint singularity), eps_rel 1e-10, truncated where the
 * integrand has decayed below 1e-18 of its peak.  It agrees with the closed
 * form T K_2(1/T) (tests/test_oracle_dist.py), i.e. to better than the
 * reference's own 1e-5 request.
 */
#include "rimo.h"
#include "rimo_math.h"

#define TWO_PI (2. * RIM_PI)

static double pl_norm_integrand(double g, void *ctx)
{
    const rimo_dist *d = (const rimo_dist *) ctx;
    const double p = (d->kind == RIMO_POWER_LAW) ? d->par[0] : d->par[0];
    return m_pow(g, -p) * m_exp(-g * d->inv_gamma_cutoff);
}

static double kappa_norm_integrand(double g, void *ctx)
{
    const rimo_dist *d = (const rimo_dist *) ctx;
    const double kappa = d->par[0];
    return g * m_sqrt(g * g - 1.) *
        m_pow(1. + (g - 1.) * d->inv_kappa_width, -(kappa + 1.)) *
        m_exp(-g * d->inv_gamma_cutoff);
}

static double tj_norm_integrand(double u, void *ctx)
{
    const rimo_dist *d = (const rimo_dist *) ctx;
    const double u2 = u * u;
    const double g = 1. + u2;
    /* g sqrt(g^2-1) exp(-g/T) dg,  g = 1+u^2, sqrt(g^2-1) = u sqrt(u^2+2), dg = 2u du */
    return g * (u * m_sqrt(u2 + 2.)) * m_exp(d->neg_inverse_t * g) * (2. * u);
}

int rimo_dist_init(rimo_dist *d, int kind, const double *params)
{
    double integral = 0, abserr = 0;
    int st = 0;
    rimo_workspace *ws;

    d->kind = kind;
    for (int i = 0; i < RIMO_MAX_PARAMS; i++) d->par[i] = 0.;
    d->inv_gamma_cutoff = 0.;
    d->inv_kappa_width = 0.;
    d->neg_inverse_t = 0.;
    d->norm = RIM_NAN;

    ws = rimo_workspace_alloc(1000);

    switch (kind) {
    case RIMO_POWER_LAW: /* p, gamma_min, gamma_max, gamma_cutoff */
        for (int i = 0; i < 4; i++) d->par[i] = params[i];
        d->inv_gamma_cutoff = 1. / params[3];
        st = rimo_qag(pl_norm_integrand, d, d->par[1], d->par[2], 0., 1e-8, 1000, ws, &integral, &abserr, NULL);
        if (!st) d->norm = 1. / (2. * TWO_PI * integral);
        break;
    case RIMO_THERMAL_JUETTNER: { /* T */
        d->par[0] = params[0];
        d->neg_inverse_t = -1. / params[0];
        /* exp(-(1+u^2)/T) < 1e-18 exp(-1/T) u^-4 ... for u^2 > 60 T is ample */
        const double umax = m_sqrt(60. * params[0] + 4.);
        st = rimo_qag(tj_norm_integrand, d, 0., umax, 0., 1e-10, 1000, ws, &integral, &abserr, NULL);
        if (!st) d->norm = 1. / (2. * TWO_PI * integral);
        break;
    }
    case RIMO_PITCHY_PL: { /* p, k, gamma_min, gamma_max, gamma_cutoff */
        for (int i = 0; i < 5; i++) d->par[i] = params[i];
        d->inv_gamma_cutoff = 1. / params[4];
        const double pa_integral = rimo_hyperg_2F1_at_1(0.5, -0.5 * d->par[1], 1.5);
        st = rimo_qag(pl_norm_integrand, d, d->par[2], d->par[3], 0., 1e-8, 1000, ws, &integral, &abserr, NULL);
        if (!st) d->norm = 1. / (2. * TWO_PI * pa_integral * integral);
        break;
    }
    case RIMO_PITCHY_KAPPA: { /* kappa, width, k, gamma_cutoff */
        for (int i = 0; i < 4; i++) d->par[i] = params[i];
        d->inv_kappa_width = 1. / (params[0] * params[1]);
        d->inv_gamma_cutoff = 1. / params[3];
        const double pa_integral = rimo_hyperg_2F1_at_1(0.5, -0.5 * d->par[2], 1.5);
        const double g_cut = 1. / d->inv_gamma_cutoff;
        st = rimo_qag(kappa_norm_integrand, d, 1., 1e3 * g_cut, 0., 1e-8, 1000, ws, &integral, &abserr, NULL);
        if (!st) d->norm = 1. / (2. * TWO_PI * pa_integral * integral);
        break;
    }
    default:
        st = 4;
    }

    rimo_workspace_free(ws);
    return st;
}

/* gamma^y exp(e), the energy factor of the three power-law-like distributions (power_law.rs:51-56, pitchy_pl.rs:74-80,
 * pitchy_kappa.rs:66-70 write powf(..) * exp(..)).  The literal flavour multiplies the two library values; the
 * deterministic flavour -- the one the kernels are compared with bit for bit -- evaluates one exponential of the
 * double-double sum (rim_powexp_normal, detmath.h), as the kernels do. */
#define POWEXP(x, y, e) (RIMO_LIT(RIMO_ATTR_POWEXP) ? m_pow(x, y) * m_exp(e) : rim_powexp_normal(x, y, e))

/* (1 + (gamma - 1) / (kappa width))^-(kappa + 1) exp(-gamma / gamma_cutoff) */
static double kappa_gamma_term(const rimo_dist *d, double gamma)
{
    const double base = 1. + (gamma - 1.) * d->inv_kappa_width;
    const double y = -(d->par[0] + 1.), e = -gamma * d->inv_gamma_cutoff;
    if (!RIMO_LIT(RIMO_ATTR_POWEXP)) {
        /* the kernels' domain test for the restricted function (dev_symphony.h dist_prepare) */
        const double kw = d->par[0] * d->par[1];
        if (rim_isfinite(d->par[0]) && kw > 1e-100 && kw < 1e100)
            return rim_powexp_normal(base, y, e);
    }
    return m_pow(base, y) * m_exp(e);
}

double rimo_calc_f(const rimo_dist *d, double gamma, double cos_xi)
{
    switch (d->kind) {
    case RIMO_POWER_LAW: {
        const double p = d->par[0];
        if (gamma < d->par[1] || gamma > d->par[2])
            return 0.;
        const double beta = m_sqrt(1. - 1. / (gamma * gamma));
        if (RIMO_LIT(RIMO_ATTR_POWEXP))
            return d->norm * m_pow(gamma, -p) * m_exp(-gamma * d->inv_gamma_cutoff) / (gamma * gamma * beta);
        return d->norm * POWEXP(gamma, -p, -gamma * d->inv_gamma_cutoff) / (gamma * gamma * beta);
    }
    case RIMO_THERMAL_JUETTNER:
        return d->norm * m_exp(d->neg_inverse_t * gamma);
    case RIMO_PITCHY_PL: {
        const double p = d->par[0], k = d->par[1];
        if (gamma < d->par[2] || gamma > d->par[3])
            return 0.;
        const double sin_xi = m_sqrt(1. - cos_xi * cos_xi);
        const double pa_term = m_pow(sin_xi, k);
        const double beta = m_sqrt(1. - 1. / (gamma * gamma));
        const double gamma_term = POWEXP(gamma, -p, -gamma * d->inv_gamma_cutoff);
        return d->norm * pa_term * gamma_term / (gamma * gamma * beta);
    }
    case RIMO_PITCHY_KAPPA: {
        const double k = d->par[2];
        const double sin_xi = m_sqrt(1. - cos_xi * cos_xi);
        const double pa_term = m_pow(sin_xi, k);
        const double gamma_term = kappa_gamma_term(d, gamma);
        return d->norm * pa_term * gamma_term;
    }
    }
    return RIM_NAN;
}

void rimo_calc_f_derivatives(const rimo_dist *d, double gamma, double cos_xi, double *dfdg, double *dfdcx)
{
    switch (d->kind) {
    case RIMO_POWER_LAW: {
        if (gamma < d->par[1] || gamma > d->par[2]) { *dfdg = 0.; *dfdcx = 0.; return; }
        const double p_plus_1 = d->par[0] + 1.;
        const double g2_minus_1 = gamma * gamma - 1.;
        if (RIMO_LIT(RIMO_ATTR_POWEXP))
            *dfdg = -d->norm * m_pow(gamma, -p_plus_1) / m_sqrt(g2_minus_1) *
                m_exp(-gamma * d->inv_gamma_cutoff) *
                (p_plus_1 / gamma + gamma / g2_minus_1 + d->inv_gamma_cutoff);
        else
            *dfdg = -d->norm * POWEXP(gamma, -p_plus_1, -gamma * d->inv_gamma_cutoff) / m_sqrt(g2_minus_1) *
                (p_plus_1 / gamma + gamma / g2_minus_1 + d->inv_gamma_cutoff);
        *dfdcx = 0.;
        return;
    }
    case RIMO_THERMAL_JUETTNER:
        *dfdg = d->norm * m_exp(d->neg_inverse_t * gamma) * d->neg_inverse_t;
        *dfdcx = 0.;
        return;
    case RIMO_PITCHY_PL: {
        const double p = d->par[0], k = d->par[1];
        if (gamma < d->par[2] || gamma > d->par[3]) { *dfdg = 0.; *dfdcx = 0.; return; }
        const double sin_xi = m_sqrt(1. - cos_xi * cos_xi);
        const double pa_term = m_pow(sin_xi, k);
        const double beta = m_sqrt(1. - 1. / (gamma * gamma));
        const double gamma_term = POWEXP(gamma, -p, -gamma * d->inv_gamma_cutoff);
        const double f = d->norm * pa_term * gamma_term / (gamma * gamma * beta);
        *dfdg = -f * ((p + 1.) / gamma + gamma / (gamma * gamma - 1.) + d->inv_gamma_cutoff);
        *dfdcx = -f * k * cos_xi / (sin_xi * sin_xi);
        return;
    }
    case RIMO_PITCHY_KAPPA: {
        const double kappa = d->par[0], width = d->par[1], k = d->par[2];
        const double sin_xi = m_sqrt(1. - cos_xi * cos_xi);
        const double pa_term = m_pow(sin_xi, k);
        const double gamma_term = kappa_gamma_term(d, gamma);
        const double f = d->norm * pa_term * gamma_term;
        *dfdg = -f * ((kappa + 1.) / (kappa * width + gamma - 1.) + d->inv_gamma_cutoff);
        *dfdcx = -f * k * cos_xi / (sin_xi * sin_xi);
        return;
    }
    }
    *dfdg = RIM_NAN;
    *dfdcx = RIM_NAN;
}
