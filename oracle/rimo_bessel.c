/* rimo_bessel.c -- oracle restatement of the Leung Bessel evaluator
 * (TEST INFRASTRUCTURE ONLY, see rimo.h).
 *
 * Follows leung-bessel/src/bessel.c of the reference:
 *   exp_factor               bessel.c:22-51
 *   BesselJ_Meissel_First    bessel.c:94-149   (Chishtie et al. 2005 series)
 *   BesselJ_Debye_Eps_Exp    bessel.c:159-213
 *   pkgw_bessel_j            bessel.c:318-376  (region selection + linear blend)
 *   pkgw_bessel_dj           bessel.c:379-405
 *
 * Deliberately reproduced quirks: the Meissel-1 prefactor uses (n+1) Gamma(n)
 * in place of Gamma(n+1) (bessel.c:123-124,145), so J is low by n/(n+1) on
 * that branch; the 0.6666666 slope literal; n >= 1e15 -> NaN in the
 * derivative; x > 1e55 -> NaN in the Debye expansion; non-integer n < 30 -> NaN.
 *
 * BesselJ_Meissel_Second (bessel.c:57-88, x > n) evaluates acos and the cosine of
 * its phase in long double; the libm build does the same, the deterministic build
 * (and the HIP library) use fp64 functions of detmath.h -- a phase error of about
 * n * 1e-15.  The hot path never reaches that expansion (hard
 * part 2: z/n < 1 always); it completes the pkgw_bessel_j seam.
 *
 * Horner steps are fused multiply-adds in the default build (one rounding
 * instead of two; the reference's C is unfused) -- the HIP kernels use the
 * same fused steps so both sides round identically.
 *
 * Integer orders n < 30 go to gsl_sf_bessel_Jn in the reference
 * (bessel.c:327-333).  GSL is absent; J_n(x) is evaluated here by Miller's
 * backward recurrence normalised with 1 = J_0 + 2 sum J_2k (accurate to a few
 * ulp, like GSL's own CF1+recurrence scheme).
 */
#include "rimo.h"
#include "rimo_math.h"
#include "../rimphony_amd/csrc/leung_table.h"

static double exp_factor(const double f_factor, const double f_exp)
{
    double fabs_exp;

    if (f_factor == 0.)
        return 0.;

    fabs_exp = m_fabs(f_exp);

    if (fabs_exp < 1e-3) {
        /* 8th-order Taylor polynomial of exp, integer coefficients 8!/k! */
        const double x = f_exp;
        double q = 8 + x;
        q = m_fma(q, x, 56);
        q = m_fma(q, x, 336);
        q = m_fma(q, x, 1680);
        q = m_fma(q, x, 6720);
        q = m_fma(q, x, 20160);
        q = m_fma(q, x, 40320);
        return f_factor * (1 + (q * x / 40320.));
    }

    if (fabs_exp > 690.) {
        const double sign_f = (f_factor < 0) ? -1. : 1.;
        const double log_f = m_log(m_fabs(f_factor));
        if (log_f * f_exp < 0.)
            return sign_f * m_exp(log_f + f_exp);
        return f_factor * m_exp(f_exp);
    }

    return f_factor * m_exp(f_exp);
}

/* coefficient rows of the Meissel "first" expansion: sum_k A_k(t1) U^(k+1), A_k
 * polynomials in t1 = (x/n)^2 (ascending coefficients), all over 10321920 */
static const double MF_A0[] = { 860160, 1290240 };
static const double MF_A1[] = { 0, -2580480, -645120 };
static const double MF_A2[] = { -28672, 2709504, 6547968, 672000 };
static const double MF_A3[] = { 0, -2580480, -23224320, -18708480, -1048320 };
static const double MF_A4[] = { -8192, -2519040, -60518400, -151828480, -61254720, -2163168 };
static const double MF_A5[] = { 0, 2580480, 138700800, 800163840, 940423680, 228049920, 5537280 };
static const double MF_A6[] = { 6144, -2644992, -299351808, -3405435264., -8653594320., -5897669400., -954875250, -16907985 };
static const double MF_A7[] = { 0, 2580480, 625766400, 12841758720., 60631119360., 86387857920., 38435160960., 4450158720., 59968440 };
static const double *const MF_ROWS[8] = { MF_A0, MF_A1, MF_A2, MF_A3, MF_A4, MF_A5, MF_A6, MF_A7 };
static const int MF_DEG[8] = { 1, 2, 3, 4, 5, 6, 7, 8 };

static double horner(const double *c, int deg, double t)
{
    double q = c[deg];
    for (int i = deg - 1; i >= 0; i--)
        q = m_fma(q, t, c[i]);
    return q;
}

static double meissel_first(const double n, const double x)
{
    double exp_val;

    const double z = x / n;
    const double eps = (n - x) / n;
    const double ninv = 1. / n;
    double Z, U, factor;
    if (RIMO_LIT(RIMO_ATTR_BESSEL)) {
        Z = m_sqrt(eps * (1 + z));
        U = 1. / (n * Z * Z * Z);
        /* Gamma(n+1) replaced by (n+1) Gamma(n), as in the reference */
        factor = 1. / ((n + 1.) * m_sqrt(Z));
    } else {
        /* the deterministic flavour takes the three roots from one inverse fourth root, as the kernels do (detmath.h) */
        double inv_z3, inv_sqrt_z;
        rim_meissel_roots(eps * (1 + z), &Z, &inv_z3, &inv_sqrt_z);
        U = ninv * inv_z3;
        factor = (1. / (n + 1.)) * inv_sqrt_z;
    }
    const double t1 = z * z;
    const double t2 = ninv * ninv;

    /* V_n sum, part 1: Horner in U over the rows A_7 .. A_0 */
    double v = horner(MF_ROWS[7], MF_DEG[7], t1);
    for (int k = 6; k >= 0; k--)
        v = m_fma(v, U, horner(MF_ROWS[k], MF_DEG[k], t1));
    const double Vsum1 = (U * v) / 0.10321920e8;

    /* V_n sum, part 2 */
    const double Vsum2 = -(ninv * m_fma(m_fma(m_fma(3, t2, -4), t2, -14), t2, 420)) / 0.5040e4;


    if (eps < 1e-4 && n > 1e3) {
        const double t3 = t2 * t2;
        const double loggamma_exp = (ninv * (-420 + 14 * t2 - 4 * t3 + 3 * t3 * t2)) / 0.5040e4;
        double q = m_fma(0.139204065e9, eps, 0.160692840e9);
        q = m_fma(q, eps, 0.190139040e9);
        q = m_fma(q, eps, 0.233192960e9);
        q = m_fma(q, eps, 0.303114240e9);
        q = m_fma(q, eps, 0.442810368e9);
        q = m_fma(q, eps, 0.984023040e9);
        const double exp2 = -n * m_sqrt(2. * eps) * eps * q / 0.1476034560e10;
        exp_val = 0.5 * m_log(0.5 * n / RIM_PI) + loggamma_exp + exp2 - Vsum1 - Vsum2;
    } else {
        double invZp1;
        if (Z < 1.e-3) {
            double q = 1 - Z;
            q = m_fma(q, Z, -1);
            q = m_fma(q, Z, 1);
            q = m_fma(q, Z, -1);
            q = m_fma(q, Z, 1);
            q = m_fma(q, Z, -1);
            q = m_fma(q, Z, 1);
            invZp1 = q;
        } else {
            invZp1 = 1. / (1. + Z);
        }
        exp_val = n * (m_log(x * invZp1) - (1 - Z)) - Vsum1 - Vsum2 - m_lgamma(n);
    }

    return exp_factor(factor, exp_val);
}

/* Debye "epsilon" expansion around x = n: a degree-15 polynomial in ez = x - n
 * whose coefficients D_0..D_13(+ez^2 term) are short Horner chains in
 * z = x^(1/3), t3 = z^2, t4 = x z and x, with the At[] constants. */
static double debye_eps_exp(const double n, const double x)
{
    if (x > 1.e55)
        return RIM_NAN;

    const double ez = x - n;
    const double z = m_cbrt(x);     /* bessel.c:180: pow(x, 1./3.) */
    const double t3 = z * z;
    const double t4 = x * z;
    const double t10 = t4 * t4;
    const double t146 = t10 * t10;
    const double K38 = 810485676000000. * RIM_AT4;

    double d[14], q;

    q = 19451656224000000. * RIM_AT0;
    q = m_fma(q, t4, 69470200800000. * RIM_AT4);
    q = m_fma(q, t3, -5403237840000. * RIM_AT6);
    d[0] = q * t10 * z;

    q = 19451656224000000. * RIM_AT1;
    q = m_fma(q, t3, -1296777081600000. * RIM_AT3);
    q = m_fma(q, t4, -8027667648000. * RIM_AT7);
    q = m_fma(q, t3, 8027667648000. * RIM_AT9);
    q = m_fma(q, z, -36011689560. * RIM_AT10);
    q = m_fma(q, t3, 3012121710. * RIM_AT12);
    q = m_fma(q, z, 4707059994. * RIM_AT13);
    d[1] = m_fma(q, t3, -401283384. * RIM_AT15);

    q = m_fma(-K38, t3, 67540473000000. * RIM_AT6);
    q = m_fma(q, t4, 484040056500. * RIM_AT10);
    q = m_fma(q, t3, -41423013450. * RIM_AT12);
    d[2] = q * x;

    q = 3241942704000000. * RIM_AT3;
    q = m_fma(q, t4, 29331862560000. * RIM_AT7);
    q = m_fma(q, t3, -2594411820000. * RIM_AT9);
    q = m_fma(q, t4, -19964735910. * RIM_AT13);
    d[3] = m_fma(q, t3, 1748257220. * RIM_AT15);

    q = m_fma(K38, t3, -94556662200000. * RIM_AT6);
    q = m_fma(q, t4, -860873013000. * RIM_AT10);
    q = m_fma(q, t3, 78248884350. * RIM_AT12);
    d[4] = q * x;

    q = m_fma(-21612951360000. * RIM_AT7, t3, 2283511230000. * RIM_AT9);
    q = m_fma(q, t4, 20997160275. * RIM_AT13);
    d[5] = m_fma(q, t3, -1938419560. * RIM_AT15);

    q = 27016189200000. * RIM_AT6;
    q = m_fma(q, t4, 459918459000. * RIM_AT10);
    q = m_fma(q, t3, -47153256150. * RIM_AT12);
    d[6] = q * x;

    q = m_fma(3859455600000. * RIM_AT7, t3, -643242600000. * RIM_AT9);
    q = m_fma(q, t4, -8397889500. * RIM_AT13);
    d[7] = m_fma(q, t3, 849093050. * RIM_AT15);

    q = m_fma(-88445857500. * RIM_AT10, t3, 11448186750. * RIM_AT12);
    d[8] = q * x;

    q = 53603550000. * RIM_AT9;
    q = m_fma(q, t4, 1474097625. * RIM_AT13);
    d[9] = m_fma(q, t3, -173573400. * RIM_AT15);

    q = m_fma(5360355000. * RIM_AT10, t3, -1161410250. * RIM_AT12);
    d[10] = q * x;

    d[11] = m_fma(-113704500. * RIM_AT13, t3, 17481100. * RIM_AT15);
    d[12] = (40608750. * RIM_AT12) * x;
    d[13] = m_fma(14875. * RIM_AT15, ez * ez, m_fma(3123750. * RIM_AT13, t3, -833000. * RIM_AT15));

    q = d[13];
    for (int k = 12; k >= 0; k--)
        q = m_fma(q, ez, d[k]);

    return q / (RIM_PI * t146 * 0.58354968672000000e17);
}

/* ---- integer order, n < 30 -------------------------------------------- */

/* Starting order of the backward recurrence: depends on x only, so one pass
 * serves every order 0..31 with identical bits. */
static int miller_start(double x)
{
    const double m = (x > 32.) ? x : 32.;
    int N = (int) (m + 24. + 4.5 * m_sqrt(m));
    return N + (N & 1);      /* even */
}

/* J_n(x) for integer 0 <= n < 30 and x > 5e4, beyond the reach of the backward recurrence above: Hankel's */
/* asymptotic expansion (x >> n^2 here, the terms fall off like (n^2 / 2x)^k).  GSL's gsl_sf_bessel_Jn covers */
/* these arguments in the reference (bessel.c:327-333); the integrator never needs them (x < n + 1 <= 30). */
static double jn_hankel(int n, double x)
{
    const double mu = 4. * (double) n * (double) n;
    const double inv8x = 1. / (8. * x);
    double term = 1., p = 1., q = 0.;
    for (int k = 1; k <= 40; k++) {
        const double odd = (double) (2 * k - 1);
        const double next = term * ((mu - odd * odd) * inv8x / (double) k);
        if (m_fabs(next) >= m_fabs(term) && k > 1) break;      /* the asymptotic series has started to diverge */
        term = next;
        /* terms alternate between Q (k odd) and P (k even), with signs + - - + + - - + ... */
        if (k & 1) q = q + (((k >> 1) & 1) ? -term : term);
        else p = p + (((k >> 1) & 1) ? -term : term);
        if (m_fabs(term) < 1e-18) break;
    }
    /* chi = x - (2n + 1) pi / 4: rotate (cos x, sin x) by an exact multiple of pi/4 */
    double sx, cx;
    m_sincos(x, &sx, &cx);
    const double r = 0.70710678118654752440;
    double cphi, sphi;
    switch ((2 * n + 1) & 7) {
    case 1: cphi = r; sphi = r; break;
    case 3: cphi = -r; sphi = r; break;
    case 5: cphi = -r; sphi = -r; break;
    default: cphi = r; sphi = -r; break;
    }
    const double cchi = cx * cphi + sx * sphi;
    const double schi = sx * cphi - cx * sphi;
    return m_sqrt(2. / (RIM_PI * x)) * (p * cchi - q * schi);
}

double rimo_bessel_jn_int(int n, double x)
{
    if (n < 0 || !(x >= 0)) return RIM_NAN;
    if (x == 0.) return n == 0 ? 1. : 0.;
    if (x > 5.0e4) return jn_hankel(n, x);     /* beyond the backward recurrence (hot path has x < n + 1 <= 30) */

    /* leading term of the power series suffices when (x/2)^2/(n+1) < 2^-55 */
    if (x * x < 1.0e-16 * (n + 1)) {
        double r = 1.;
        const double hx = 0.5 * x;
        for (int k = 1; k <= n; k++)
            r = r * (hx / k);
        return r;
    }

    const int N = miller_start(x);
    const double tox = 2. / x;
    double jp1 = 0., j = 1e-300, sum = 0., want = 0.;
    /* j_{k-1} = (2k/x) j_k - j_{k+1} */
    for (int k = N; k >= 1; k--) {
        const double jm1 = m_fma(k * tox, j, -jp1);
        jp1 = j;
        j = jm1;
        /* now j = J_{k-1} (unnormalised) */
        if (m_fabs(j) > 1e250) {
            j *= 1e-250; jp1 *= 1e-250; sum *= 1e-250; want *= 1e-250;
        }
        if (k - 1 == n) want = j;
        if (((k - 1) & 1) == 0 && k - 1 > 0) sum += 2. * j;
    }
    sum += j;   /* J_0 */
    return want / sum;
}

/* ---- pkgw_bessel_j / pkgw_bessel_dj ------------------------------------ */

static const double MINUS_ETA_A_INTERCEPT = 0.174857;
static const double MINUS_ETA_B_INTERCEPT = 0.295966;
static const double PLUS_ETA_A_INTERCEPT = 0.151550;
static const double PLUS_ETA_B_INTERCEPT = 0.438914;

/* Meissel's "second" expansion (bessel.c:57-88), x > n.  Not on the hot path; completes pkgw_bessel_j.
 * The reference computes acos and the cosine of the phase in long double (kept in the libm build);
 * the deterministic build uses the fp64 functions of detmath.h, as the HIP library does. */
static double meissel_second(const double n, const double x)
{
    const double z = x / n;
    const double eps = (x - n) / n;
    const double Z = m_sqrt(eps * (1 + z));
    const double U = 1. / (n * Z * Z * Z);
    const double t1 = z * z;
    const double t2 = U * U;

    double p3 = m_fma(71391., t1, 5297808.);
    p3 = m_fma(p3, t1, 45756144.);
    p3 = m_fma(p3, t1, 102842688.);
    p3 = m_fma(p3, t1, 72179904.);
    p3 = m_fma(p3, t1, 15287808.);
    p3 = m_fma(p3, t1, 744960.);
    p3 = m_fma(p3, t1, 3072.);
    double p2 = m_fma(-6592., t1, -271488.);
    p2 = m_fma(p2, t1, -1119552.);
    p2 = m_fma(p2, t1, -952576.);
    p2 = m_fma(p2, t1, -165120.);
    p2 = m_fma(p2, t1, -3072.);
    double p1 = m_fma(1248., t1, 22272.);
    p1 = m_fma(p1, t1, 27648.);
    p1 = m_fma(p1, t1, 3072.);
    const double p0 = m_fma(-768., t1, -3072.);
    double pe = m_fma(p3, t2, p2);
    pe = m_fma(pe, t2, p1);
    pe = m_fma(pe, t2, p0);
    const double exp_val = (t1 * t2 * pe) / 0.12288e5;

    const double Qt = n * (Z - m_acos01(n / x));

    double q3 = m_fma(16907985., t1, 954875250.);
    q3 = m_fma(q3, t1, 5897669400.);
    q3 = m_fma(q3, t1, 8653594320.);
    q3 = m_fma(q3, t1, 3405435264.);
    q3 = m_fma(q3, t1, 299351808.);
    q3 = m_fma(q3, t1, 2644992.);
    q3 = m_fma(q3, t1, -6144.);
    double q2 = m_fma(2163168., t1, 61254720.);
    q2 = m_fma(q2, t1, 151828480.);
    q2 = m_fma(q2, t1, 60518400.);
    q2 = m_fma(q2, t1, 2519040.);
    q2 = m_fma(q2, t1, 8192.);
    double q1 = m_fma(-672000., t1, -6547968.);
    q1 = m_fma(q1, t1, -2709504.);
    q1 = m_fma(q1, t1, 28672.);
    const double q0 = m_fma(1290240., t1, 860160.);
    double qs = m_fma(q3, t2, q2);
    qs = m_fma(qs, t2, q1);
    qs = m_fma(qs, t2, q0);
    const double Qsum = -(U * qs) / 0.10321920e8;

    const double factor = m_sqrt(2 / (RIM_PI * n * Z)) * m_cos_phase(Qsum + Qt - 0.78539816339744830962);
    return exp_factor(factor, exp_val);
}

double rimo_bessel_j(const double n, const double x)
{
    double logn;

    if (!(n >= 0 && x >= 0))
        return RIM_NAN;

    if (n < 30.) {
        int n_int = (int) n;
        if (n_int != n)
            return RIM_NAN;
        return rimo_bessel_jn_int(n_int, x);
    }

    if (x == n)
        return debye_eps_exp(n, x);

    logn = m_log10(n);

    if (x < n) {
        const double eta = m_log10_region((n - x) / n);
        const double eta_thresh_lo = -0.6666666 * logn + MINUS_ETA_A_INTERCEPT;
        const double eta_thresh_hi = -0.6666666 * logn + MINUS_ETA_B_INTERCEPT;

        if (eta < eta_thresh_lo)
            return debye_eps_exp(n, x);
        if (eta > eta_thresh_hi)
            return meissel_first(n, x);
        {
            const double debye = debye_eps_exp(n, x);
            const double meissel1 = meissel_first(n, x);
            const double pos = (eta - eta_thresh_lo) / (MINUS_ETA_B_INTERCEPT - MINUS_ETA_A_INTERCEPT);
            return debye * (1 - pos) + meissel1 * pos;
        }
    } else {
        const double eta = m_log10_region((x - n) / x);
        const double eta_thresh_lo = -0.6666666 * logn + PLUS_ETA_A_INTERCEPT;
        const double eta_thresh_hi = -0.6666666 * logn + PLUS_ETA_B_INTERCEPT;
        if (eta < eta_thresh_lo)
            return debye_eps_exp(n, x);
        if (eta > eta_thresh_hi)
            return meissel_second(n, x);
        {
            const double debye = debye_eps_exp(n, x);
            const double meissel2 = meissel_second(n, x);
            const double pos = (eta - eta_thresh_lo) / (PLUS_ETA_B_INTERCEPT - PLUS_ETA_A_INTERCEPT);
            return debye * (1 - pos) + meissel2 * pos;
        }
    }
}

double rimo_bessel_dj(const double n, const double x)
{
    if (n >= 1e15) {
        return RIM_NAN;
    } else {
        const double jn = rimo_bessel_j(n, x);
        const double jnp1 = rimo_bessel_j(n + 1, x);

        if (x == 0.) {
            if (n >= 2.)
                return 0.;
            if (n == 0.)
                return -jnp1;
            return n * jn / RIM_DBL_MIN - jnp1;
        }
        return n * jn / x - jnp1;
    }
}
