// dev_symphony.h -- per-sample device functions of the Symphony path:
// the four distribution functions and gamma_integrand.
//
//   DistributionFunction::calc_f / calc_f_derivatives
//       power_law.rs:36-62, thermal_juettner.rs:29-39, pitchy_pl.rs:32-64,
//       pitchy_kappa.rs:38-62
//   CalculationState::gamma_integrand      symphony.rs:398-479
//
// The reference reaches these through a GSL callback trampoline per sample
// (gsl.rs:111-117); here they are inlined into the wavefront quadrature, one
// sample per lane.  Operation order matches oracle/rimo_dist.c and
// oracle/rimo_symphony.c exactly (bit-for-bit parity is tested).
#ifndef RIM_DEV_SYMPHONY_H
#define RIM_DEV_SYMPHONY_H

#include "dev_bessel.h"

namespace rim {

enum { DIST_POWER_LAW = 0, DIST_THERMAL_JUETTNER = 1, DIST_PITCHY_PL = 2, DIST_PITCHY_KAPPA = 3 };
enum { STOKES_I = 0, STOKES_Q = 1, STOKES_V = 2 };
enum { COEFF_EMISSION = 0, COEFF_ABSORPTION = 1, COEFF_FARADAY = 2 };

// Distribution parameters of one point (wave-uniform).  par[] follows the C ABI:
//   power_law {p, gmin, gmax, gcut}; thermal {T}; pitchy_pl {p, k, gmin, gmax, gcut};
//   pitchy_kappa {kappa, width, k, gcut}.
struct DistParams {
    double par[5];
    double inv_gamma_cutoff;
    double inv_kappa_width;
    double neg_inverse_t;
    double norm;
};

template <int KIND>
RIM_DEV void dist_prepare(DistParams &d, double norm)
{
    d.inv_gamma_cutoff = 0.;
    d.inv_kappa_width = 0.;
    d.neg_inverse_t = 0.;
    d.norm = norm;
    if (KIND == DIST_POWER_LAW) d.inv_gamma_cutoff = 1. / d.par[3];
    if (KIND == DIST_THERMAL_JUETTNER) d.neg_inverse_t = -1. / d.par[0];
    if (KIND == DIST_PITCHY_PL) d.inv_gamma_cutoff = 1. / d.par[4];
    if (KIND == DIST_PITCHY_KAPPA) {
        d.inv_kappa_width = 1. / (d.par[0] * d.par[1]);
        d.inv_gamma_cutoff = 1. / d.par[3];
        // par[4] (unused by this distribution in the C ABI) flags the points whose kappa term may use the
        // restricted power function: a finite exponent and 1 + (gamma - 1) / (kappa width) a positive normal
        // number for every gamma >= 1.  Anything else -- kappa = +-inf (1^inf), kappa width < 0 (negative base) --
        // takes the general rim_pow, as the reference's powf would treat it (tools/hostile_sweep.py).
        const double kw = d.par[0] * d.par[1];
        d.par[4] = (rim_isfinite(d.par[0]) && kw > 1e-100 && kw < 1e100) ? 1. : 0.;
    }
}

// (1 + (gamma - 1) / (kappa width))^-(kappa + 1) exp(-gamma / gamma_cutoff) of the pitchy-kappa distribution
template <int PREC = 0>
RIM_DEV double kappa_gamma_term(const DistParams &d, double gamma)
{
    const double base = 1. + (gamma - 1.) * d.inv_kappa_width;
    const double y = -(d.par[0] + 1.);
    if (rim_bits(d.par[4]) != 0) return RimMath<PREC>::powexp_normal(base, y, -gamma * d.inv_gamma_cutoff);     // wave-uniform test
    return rim_pow(base, y) * RimMath<PREC>::exp(-gamma * d.inv_gamma_cutoff);
}

template <int KIND, int PREC = 0>
RIM_DEV double calc_f(const DistParams &d, double gamma, double cos_xi)
{
    typedef RimMath<PREC> M;
    if (KIND == DIST_POWER_LAW) {
        if (gamma < d.par[1] || gamma > d.par[2]) return 0.;
        RIM_HIT(19);
        const double beta = rim_sqrt(1. - rim_div_moderate(1., gamma * gamma));
        return d.norm * M::powexp_normal(gamma, -d.par[0], -gamma * d.inv_gamma_cutoff) / (gamma * gamma * beta);
    } else if (KIND == DIST_THERMAL_JUETTNER) {
        return d.norm * M::exp(d.neg_inverse_t * gamma);
    } else if (KIND == DIST_PITCHY_PL) {
        if (gamma < d.par[2] || gamma > d.par[3]) return 0.;
        const double sin_xi = rim_sqrt(1. - cos_xi * cos_xi);
        const double pa_term = M::pow(sin_xi, d.par[1]);
        const double beta = rim_sqrt(1. - rim_div_moderate(1., gamma * gamma));
        const double gamma_term = M::powexp_normal(gamma, -d.par[0], -gamma * d.inv_gamma_cutoff);
        return d.norm * pa_term * gamma_term / (gamma * gamma * beta);
    } else {
        const double sin_xi = rim_sqrt(1. - cos_xi * cos_xi);
        const double pa_term = M::pow(sin_xi, d.par[2]);
        const double gamma_term = kappa_gamma_term<PREC>(d, gamma);
        return d.norm * pa_term * gamma_term;
    }
}

template <int KIND, int PREC = 0>
RIM_DEV void calc_f_derivatives(const DistParams &d, double gamma, double cos_xi, double &dfdg, double &dfdcx)
{
    typedef RimMath<PREC> M;
    if (KIND == DIST_POWER_LAW) {
        if (gamma < d.par[1] || gamma > d.par[2]) { dfdg = 0.; dfdcx = 0.; return; }
        RIM_HIT(20);
        const double p_plus_1 = d.par[0] + 1.;
        const double g2_minus_1 = gamma * gamma - 1.;
        dfdg = -d.norm * M::powexp_normal(gamma, -p_plus_1, -gamma * d.inv_gamma_cutoff) / rim_sqrt(g2_minus_1) *
            (p_plus_1 / gamma + gamma / g2_minus_1 + d.inv_gamma_cutoff);
        dfdcx = 0.;
    } else if (KIND == DIST_THERMAL_JUETTNER) {
        dfdg = d.norm * M::exp(d.neg_inverse_t * gamma) * d.neg_inverse_t;
        dfdcx = 0.;
    } else if (KIND == DIST_PITCHY_PL) {
        if (gamma < d.par[2] || gamma > d.par[3]) { dfdg = 0.; dfdcx = 0.; return; }
        const double p = d.par[0], k = d.par[1];
        const double sin_xi = rim_sqrt(1. - cos_xi * cos_xi);
        const double pa_term = M::pow(sin_xi, k);
        const double beta = rim_sqrt(1. - rim_div_moderate(1., gamma * gamma));
        const double gamma_term = M::powexp_normal(gamma, -p, -gamma * d.inv_gamma_cutoff);
        const double f = d.norm * pa_term * gamma_term / (gamma * gamma * beta);
        dfdg = -f * ((p + 1.) / gamma + gamma / (gamma * gamma - 1.) + d.inv_gamma_cutoff);
        dfdcx = -f * k * cos_xi / (sin_xi * sin_xi);
    } else {
        const double kappa = d.par[0], width = d.par[1], k = d.par[2];
        const double sin_xi = rim_sqrt(1. - cos_xi * cos_xi);
        const double pa_term = M::pow(sin_xi, k);
        const double gamma_term = kappa_gamma_term<PREC>(d, gamma);
        const double f = d.norm * pa_term * gamma_term;
        dfdg = -f * ((kappa + 1.) / (kappa * width + gamma - 1.) + d.inv_gamma_cutoff);
        dfdcx = -f * k * cos_xi / (sin_xi * sin_xi);
    }
}

// Observer/point data shared by all samples of a coefficient (wave-uniform).
struct SymPoint {
    double s, cos_th, sin_th;
    int coeff, stokes;
};

// Order data for one gamma-integral: J_n and J_{n+1} at fixed n (wave-uniform).
// The two LeungOrder records live in memory the caller provides -- LDS in the wave kernels -- and
// are read where they are used: held in registers they would cost 56 SGPRs for the whole integrand
// (the kernel then spills SGPRs into VGPR lanes inside the hot loop).
struct SymOrder {
    double n;
    bool small;         // integer n < 30: Miller recurrence instead of the Leung expansions
    bool np1_small;     // n + 1 < 30
    bool dj_nan;        // n >= 1e15 (bessel.c:382-388)
    const LeungOrder *o;  // [2]: orders n and n + 1 (valid when >= 30)
};

RIM_DEV SymOrder sym_order(double n, LeungOrder *store)
{
    SymOrder so;
    so.n = n;
    so.small = n < 30.;
    so.np1_small = (n + 1.) < 30.;
    so.dj_nan = n >= 1e15;
    store[0] = LeungOrder();
    store[1] = LeungOrder();
    if (!so.small) store[0] = leung_order(n);
    if (!so.np1_small) store[1] = leung_order(n + 1.);
    so.o = store;
    return so;
}

// J_n(z) and J'_n(z) as the reference's pkgw_bessel_j / pkgw_bessel_dj pair would
// return them.  The two Leung evaluations (orders n and n+1) run through one
// loop body so the expansions are inlined once.
template <int PREC = 0>
RIM_DEV void sym_bessel_pair(const SymOrder &so, double z, double &jn, double &djn)
{
    const double n = so.n;
    double jv0 = 0., jv1 = 0.;
    if (!so.np1_small) {
        // which expansions do the two orders need at this z?  (order n only if it is >= 30)
        RIM_PROF_T(t_sel);
        double pos0 = 0., pos1 = 0.;
        int c0 = 0;
        if (!so.small) c0 = leung_select_code(so.o[0], z, pos0);
        const int c1 = leung_select_code(so.o[1], z, pos1);
        RIM_PROF_ADD(3, t_sel);
        RIM_LANES(30, true);
        RIM_LANES(26, ((c0 | c1) & LSEL_DEBYE) != 0);
        RIM_LANES(27, (c0 & LSEL_MEISSEL) != 0);
        RIM_LANES(28, (c1 & LSEL_MEISSEL) != 0);
        RIM_LANES(29, (c1 & LSEL_BLEND) != 0);
        // Debye for both orders in one go (they share everything but x - n), Meissel order by order
        RIM_PROF_T(t_deb);
        double deb0 = 0., deb1 = 0.;
        if ((c0 | c1) & LSEL_DEBYE) debye_eps_pair<PREC>(so.o[0].n, so.o[1].n, z, &deb0, &deb1);
        RIM_PROF_ADD(4, t_deb);
        RIM_PROF_T(t_mei);
        double mei0 = 0., mei1 = 0.;
        if (c0 & LSEL_MEISSEL) mei0 = meissel_first<PREC, true>(so.o[0], z);
        if (c1 & LSEL_MEISSEL) mei1 = meissel_first<PREC, true>(so.o[1], z);
        RIM_PROF_ADD(5, t_mei);
        if (!so.small) jv0 = leung_combine_code(c0, pos0, deb0, mei0);
        jv1 = leung_combine_code(c1, pos1, deb1, mei1);
    }
    if (so.small) {
        // the reference returns NaN for non-integer n < 30 (bessel.c:327-331)
        const int n_int = (int) n;
        if (!(n >= 0 && z >= 0) || (double) n_int != n) {
            jv0 = RIM_NAN; jv1 = RIM_NAN;
        } else {
            double a, b;
            RIM_HIT(16);
            RIM_PROF_T(t_mil);
            jn_int_pair(n_int, z, &a, &b);
            RIM_PROF_ADD(8, t_mil);
            jv0 = a;
            if (so.np1_small) jv1 = b;
        }
    }
    jn = jv0;
    const double jnp1 = jv1;
    if (so.dj_nan) { djn = RIM_NAN; return; }
    if (z == 0.) {
        if (n >= 2.) djn = 0.;
        else if (n == 0.) djn = -jnp1;
        else djn = n * jn / RIM_DBL_MIN - jnp1;
        return;
    }
    djn = n * jn / z - jnp1;
}

// gamma_integrand (symphony.rs:398-479) in three pieces, so that the coefficients of one parameter point can share
// the part that depends on (s, theta, n, gamma) only -- the kinematics and the Bessel pair, symphony.rs:406-442 -- and
// differ in the polarisation term (:444-448) and the distribution term (:455-463): symphony_group.h.
struct GiShared {
    double gamma, beta, cos_xi;
    double mj, njp;             // M J_n(z), N J'_n(z)
};

template <int PREC = 0>
RIM_DEV GiShared gamma_integrand_shared(double s, double cos_th, double sin_th, const SymOrder &so, double gamma)
{
    const double n = so.n;

    const double beta = rim_sqrt(1. - rim_div_moderate(1., gamma * gamma));
    // (numerators that can be exactly 0 give the IEEE signed zero through the bare sequence too)
    const double cos_xi = rim_div_moderate(s * gamma - n, s * gamma * beta * cos_th);
    const double sin_xi = rim_sqrt(1. - cos_xi * cos_xi);
    const double m = rim_div_moderate(cos_th - beta * cos_xi, sin_th);
    const double big_n = beta * sin_xi;

    double gamma_sin_xi;
    if (beta < 0.1) {
        RIM_HIT(14);
        gamma_sin_xi = gamma * sin_xi;
    } else {
        RIM_HIT(15);
        const double bc = beta * cos_th;
        const double beta2_costh2 = bc * bc;
        const double s_on_r = rim_div_moderate(2. * n, s * (beta2_costh2 - 1.));
        const double r = 1. - rim_div_moderate(1., beta2_costh2);
        gamma_sin_xi = rim_sqrt(r * (gamma * (gamma + s_on_r)) - rim_div_moderate(n * n, s * s * beta2_costh2));
    }

    const double z = s * beta * sin_th * gamma_sin_xi;

    double jn, djn;
    { RIM_PROF_T(t_cal); RIM_PROF_ADD(10, t_cal); }     // empty region: the timers' own cost
    RIM_PROF_T(t_bes);
    sym_bessel_pair<PREC>(so, z, jn, djn);
    RIM_PROF_ADD(2, t_bes);
    GiShared sh;
    sh.gamma = gamma; sh.beta = beta; sh.cos_xi = cos_xi;
    sh.mj = m * jn;
    sh.njp = big_n * djn;
    return sh;
}

RIM_DEV double gamma_integrand_pol_term(int stokes, double mj, double njp)
{
    if (stokes == STOKES_I) return mj * mj + njp * njp;
    if (stokes == STOKES_Q) return mj * mj - njp * njp;
    return 2. * mj * njp;
}

template <int KIND, int PREC = 0>
RIM_DEV double gamma_integrand_f_term(int coeff, const DistParams &d, double cos_th, const GiShared &sh)
{
    const double gamma = sh.gamma, beta = sh.beta, cos_xi = sh.cos_xi;
    if (coeff == COEFF_EMISSION) {
        RIM_HIT(24);
        return calc_f<KIND, PREC>(d, gamma, cos_xi);
    }
    double dfdg, dfdcx;
    RIM_HIT(25);
    calc_f_derivatives<KIND, PREC>(d, gamma, cos_xi, dfdg, dfdcx);
    if (KIND == DIST_POWER_LAW || KIND == DIST_THERMAL_JUETTNER) {
        // dfdcx is the constant +0 (isotropic distributions): dfdcx_factor * dfdcx is a zero with the sign of the
        // factor (a NaN only where the sample is a NaN through cos_xi anyway), and for gamma > 0 the factor
        // (beta cos_th - cos_xi) / (gamma - 1 / gamma) has the sign of (beta cos_th - cos_xi) (gamma - 1): the
        // same bits -- signed zeros of f_term included -- without the two divisions.
        return dfdg + ((beta * cos_th - cos_xi) * dfdcx) * (gamma - 1.);
    }
    const double dfdcx_factor = (beta * cos_th - cos_xi) / (gamma - 1. / gamma);
    return dfdg + dfdcx_factor * dfdcx;
}

// calc_f AND calc_f_derivatives of one sample (the emission and absorption members of a group, symphony_group.h): the
// values calc_f / calc_f_derivatives return, bit for bit, with what the two have in common computed once -- the
// double-double logarithm of gamma of the power-law energy factor (rim_powexp_from_log), the exponential of the thermal
// distribution, and for the anisotropic distributions f itself, which calc_f_derivatives forms by calc_f's own expression
// (pitchy_pl.rs:56-61, pitchy_kappa.rs:53-58).
template <int KIND>
RIM_DEV void calc_f_both(const DistParams &d, double gamma, double cos_xi, double &f, double &dfdg, double &dfdcx)
{
    if (KIND == DIST_POWER_LAW) {
        f = 0.; dfdg = 0.; dfdcx = 0.;
        if (gamma < d.par[1] || gamma > d.par[2]) return;
        RIM_HIT(19); RIM_HIT(20);
        double ll;
        const double lh = rim_log_dd_normal(gamma, &ll);
        const double e = -gamma * d.inv_gamma_cutoff;
        const double beta = rim_sqrt(1. - rim_div_moderate(1., gamma * gamma));
        f = d.norm * rim_powexp_from_log(lh, ll, -d.par[0], e) / (gamma * gamma * beta);
        const double p_plus_1 = d.par[0] + 1.;
        const double g2_minus_1 = gamma * gamma - 1.;
        dfdg = -d.norm * rim_powexp_from_log(lh, ll, -p_plus_1, e) / rim_sqrt(g2_minus_1) *
            (p_plus_1 / gamma + gamma / g2_minus_1 + d.inv_gamma_cutoff);
    } else if (KIND == DIST_THERMAL_JUETTNER) {
        f = d.norm * RimMath<0>::exp(d.neg_inverse_t * gamma);
        dfdg = f * d.neg_inverse_t;
        dfdcx = 0.;
    } else if (KIND == DIST_PITCHY_PL) {
        f = 0.; dfdg = 0.; dfdcx = 0.;
        if (gamma < d.par[2] || gamma > d.par[3]) return;
        const double p = d.par[0], k = d.par[1];
        const double sin_xi = rim_sqrt(1. - cos_xi * cos_xi);
        const double pa_term = RimMath<0>::pow(sin_xi, k);
        const double beta = rim_sqrt(1. - rim_div_moderate(1., gamma * gamma));
        const double gamma_term = RimMath<0>::powexp_normal(gamma, -p, -gamma * d.inv_gamma_cutoff);
        f = d.norm * pa_term * gamma_term / (gamma * gamma * beta);
        dfdg = -f * ((p + 1.) / gamma + gamma / (gamma * gamma - 1.) + d.inv_gamma_cutoff);
        dfdcx = -f * k * cos_xi / (sin_xi * sin_xi);
    } else {
        const double kappa = d.par[0], width = d.par[1], k = d.par[2];
        const double sin_xi = rim_sqrt(1. - cos_xi * cos_xi);
        const double pa_term = RimMath<0>::pow(sin_xi, k);
        const double gamma_term = kappa_gamma_term<0>(d, gamma);
        f = d.norm * pa_term * gamma_term;
        dfdg = -f * ((kappa + 1.) / (kappa * width + gamma - 1.) + d.inv_gamma_cutoff);
        dfdcx = -f * k * cos_xi / (sin_xi * sin_xi);
    }
}

// the emission and the absorption f_term of one sample (symphony.rs:455-463) from calc_f_both
template <int KIND>
RIM_DEV void gamma_integrand_f_terms(const DistParams &d, double cos_th, const GiShared &sh, double &f_em, double &f_ab)
{
    const double gamma = sh.gamma, beta = sh.beta, cos_xi = sh.cos_xi;
    double dfdg, dfdcx;
    RIM_HIT(24); RIM_HIT(25);
    calc_f_both<KIND>(d, gamma, cos_xi, f_em, dfdg, dfdcx);
    if (KIND == DIST_POWER_LAW || KIND == DIST_THERMAL_JUETTNER) {
        f_ab = dfdg + ((beta * cos_th - cos_xi) * dfdcx) * (gamma - 1.);       // (gamma_integrand_f_term says why)
    } else {
        const double dfdcx_factor = (beta * cos_th - cos_xi) / (gamma - 1. / gamma);
        f_ab = dfdg + dfdcx_factor * dfdcx;
    }
}

template <int KIND, int PREC = 0>
RIM_DEV double gamma_integrand(const SymPoint &pt, const DistParams &d, const SymOrder &so, double gamma)
{
    const GiShared sh = gamma_integrand_shared<PREC>(pt.s, pt.cos_th, pt.sin_th, so, gamma);
    RIM_PROF_T(t_f);
    const double pol_term = gamma_integrand_pol_term(pt.stokes, sh.mj, sh.njp);
    const double f_term = gamma_integrand_f_term<KIND, PREC>(pt.coeff, d, pt.cos_th, sh);
    RIM_PROF_ADD(6, t_f);
    return gamma * gamma * pol_term * f_term;
}

}  // namespace rim
#endif
