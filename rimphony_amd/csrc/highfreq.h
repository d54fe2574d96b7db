/* highfreq.h -- the reference's closed-form high-frequency Faraday approximations (C99 / C++ / HIP).
 *
 *   power law   HighFrequencyApproximation::faraday_q / faraday_v   src/power_law.rs:144-169
 *               (Huang & Shcherbakov 2011, eq. 51)
 *   thermal     HighFrequencyApproximation::faraday_q / faraday_v   src/thermal_juettner.rs:105-141
 *               (Heyvaerts et al. 2013, eq. 43), which need K_0, K_1, K_2 at 1/T
 *
 * The reference takes K_nu from the un-vendored `special-fun` crate (`inverse_t.besselk(n)`); here
 * K_0 and K_1 come from Temme's method (series below x = 2, Steed's continued fraction CF2 above), the
 * scheme of Numerical Recipes' `bessik` at nu = 0, and K_2 from the recurrence K_2 = K_0 + (2/x) K_1.
 * Written with + - * / sqrt fma and the detmath.h functions only, so the oracle (which includes this
 * header) and the HIP kernel agree bit for bit; tests/test_highfreq.py pins both against scipy.special.kv
 * and against an independent numpy transcription of the reference's formulas.
 */
#ifndef RIM_HIGHFREQ_H
#define RIM_HIGHFREQ_H

#include "detmath.h"

#define RIM_HF_EULER_GAMMA 0.57721566490153286061
#define RIM_HF_ELECTRON_CHARGE 4.80320680e-10
#define RIM_HF_MASS_ELECTRON 9.1093826e-28
#define RIM_HF_SPEED_LIGHT 2.99792458e10

/* K_0(x), K_1(x), K_2(x) for x > 0 */
RIM_FN void rim_bessel_k012(double x, double *k0, double *k1, double *k2)
{
    double rk0, rk1;
    if (!(x > 0.)) { *k0 = RIM_NAN; *k1 = RIM_NAN; *k2 = RIM_NAN; return; }
    if (x < 2.) {
        /* Temme's series at nu = 0: gam1 = -gamma_E, gam2 = 1, 1/Gamma(1 +- nu) = 1 */
        const double b = 0.5 * x;
        const double d0 = -rim_log(b);
        double ff = d0 - RIM_HF_EULER_GAMMA;
        double sum = ff;
        double p = 0.5, q = 0.5, c = 1.;
        const double d = b * b;
        double sum1 = p;
        for (int i = 1; i <= 500; i++) {
            const double di = (double) i;
            ff = (di * ff + p + q) / (di * di);
            c = c * (d / di);
            p = p / di;
            q = q / di;
            const double del = c * ff;
            sum = sum + del;
            const double del1 = c * (p - di * ff);
            sum1 = sum1 + del1;
            if (rim_fabs(del) < rim_fabs(sum) * 1e-17) break;
        }
        rk0 = sum;
        rk1 = sum1 * (2. / x);
    } else {
        /* Steed's algorithm for the continued fraction CF2 at nu = 0 */
        double b = 2. * (1. + x);
        double d = 1. / b;
        double h = d, delh = d;
        double q1 = 0., q2 = 1.;
        const double a1 = 0.25;
        double q = a1, c = a1, a = -a1;
        double s = 1. + q * delh;
        for (int i = 2; i <= 10000; i++) {
            a = a - (double) (2 * (i - 1));
            c = -a * c / (double) i;
            const double qnew = (q1 - b * q2) / a;
            q1 = q2;
            q2 = qnew;
            q = q + c * qnew;
            b = b + 2.;
            d = 1. / (b + a * d);
            delh = (b * d - 1.) * delh;
            h = h + delh;
            const double dels = q * delh;
            s = s + dels;
            if (rim_fabs(dels / s) < 1e-17) break;
        }
        h = a1 * h;
        rk0 = rim_sqrt(RIM_PI / (2. * x)) * rim_exp(-x) / s;
        rk1 = rk0 * (x + 0.5 - h) / x;
    }
    *k0 = rk0;
    *k1 = rk1;
    *k2 = rk0 + (2. / x) * rk1;
}

/* power_law.rs:150-156 */
RIM_FN double rim_hf_powerlaw_faraday_q(double p, double gamma_min, double s, double sin_th)
{
    return 0.0085 *
        2. / (p - 2.) *
        (rim_pow(s / (sin_th * (gamma_min * gamma_min)), (p - 2.) / 2.) - 1.) *
        (p - 1.) / rim_pow(gamma_min, 1. - p) *
        rim_pow(sin_th / s, (p + 2.) / 2.);
}

/* power_law.rs:163-169 */
RIM_FN double rim_hf_powerlaw_faraday_v(double p, double gamma_min, double s, double sin_th)
{
    return 0.017
        * (rim_log(gamma_min) * (p - 1.))
        / ((p + 1.) * (gamma_min * gamma_min))
        / s
        * sin_th;
}

/* thermal_juettner.rs:111-124 and :131-141; t = dimensionless temperature */
RIM_FN void rim_hf_thermal_faraday(double t, double s, double sin_th, double cos_th, double *rho_q, double *rho_v)
{
    const double nu_plasma_factor = 2. * RIM_HF_ELECTRON_CHARGE * RIM_HF_ELECTRON_CHARGE / RIM_HF_MASS_ELECTRON;
    const double neg_inverse_t = -1. / t;          /* ThermalJuettnerDistribution::new */
    const double inverse_t = -neg_inverse_t;
    const double tt = 1. / inverse_t;
    double k0, k1, k2;
    rim_bessel_k012(inverse_t, &k0, &k1, &k2);
    *rho_q = nu_plasma_factor
        * (sin_th * sin_th)
        * (k1 + 6. * tt * k2)
        / (2. * RIM_HF_SPEED_LIGHT * (s * s) * k2);
    *rho_v = nu_plasma_factor
        * cos_th
        * k0
        / (RIM_HF_SPEED_LIGHT * s * k2);
}

#endif
