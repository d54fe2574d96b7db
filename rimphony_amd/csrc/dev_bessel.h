// dev_bessel.h -- Leung large-order Bessel evaluator as a HIP device function.
//
// Replaces the C library the reference links per sample through its Rust shim
// (leung-bessel/src/lib.rs:56-75 -> leung-bessel/src/bessel.c:318-405):
//   pkgw_bessel_j(n, x)   region select on eta = log10((n-x)/n) (bessel.c:341-357)
//     Meissel "first" expansion  (bessel.c:94-149)
//     Debye epsilon expansion    (bessel.c:159-213)
//     linear blend between them  (bessel.c:352-357)
//   pkgw_bessel_dj(n, x) = n J_n / x - J_{n+1}  (bessel.c:379-405)
//
// MI355X-first restructuring (values unchanged):
//   * Everything that depends on the order only -- log10 n, the two region
//     thresholds, 1/n, the V_n "part 2" sum, lgamma(n), the Stirling pieces of
//     the small-epsilon branch -- is hoisted into LeungOrder and computed once
//     per gamma-integral (the order is wave-uniform there), not once per
//     sample as the CPU code does.  That removes 2 log10 + 2 lgamma + ~30 flops
//     from each of the ~1e5..1e6 samples of a coefficient.
//   * J_n(z) is needed by both the "M J_n" and the "N J'_n" terms of the
//     integrand (symphony.rs:441-442); it is evaluated once.
//   * Polynomials are explicit Horner chains of v_fma_f64 on literals (SGPR
//     operands), no coefficient tables in memory.
//   * Integer orders below 30 (GSL's gsl_sf_bessel_Jn in the reference) use one
//     Miller backward-recurrence pass whose length depends on x only, so a
//     single pass yields J_n and J_{n+1} with a wave-uniform trip count.
//
// Rounding contract: identical operation order (including which steps are
// fused) to oracle/rimo_bessel.c, so results agree bit for bit.
#ifndef RIM_DEV_BESSEL_H
#define RIM_DEV_BESSEL_H

#include "detmath.h"
#include "leung_table.h"

#if defined(__HIPCC__)
#define RIM_DEV __host__ __device__ __forceinline__
#else
#define RIM_DEV inline
#endif

namespace rim {

// TINY_ZERO (the integrand only, never the Bessel seam): a value below 1e6 exp(-690) = 2.2e-294 is returned as
// f_factor * 0.  The integrand squares or multiplies the two Bessel values of a sample (symphony.rs:441-448): with
// |J| < 2.2e-294 the other order of the pair is below 1e-270 (for x < n and n >= 30, J_n >= 1e-294 needs
// x / 2n > 1e-10, and J_{n+1} / J_n lies in (x / 2n, 1)), every product underflows to exactly 0, and the sample is
// the same 0 either way.  What it saves: the large-exponent arm below -- a logarithm and a full exponential --
// which 56 % of the integrand passes used to execute for the few lanes with -760 < f_exp < -690.
template <int PREC = 0, bool TINY_ZERO = false>
RIM_DEV double exp_factor(double f_factor, double f_exp)
{
    typedef RimMath<PREC> M;
    if (f_factor == 0.) return 0.;
    const double a = rim_fabs(f_exp);
    if (a < 1e-3) {
        RIM_HIT(10);
        const double x = f_exp;
        double q = 8 + x;
        q = rim_fma_k(q, x, 56.);
        q = rim_fma_k(q, x, 336.);
        q = rim_fma_k(q, x, 1680.);
        q = rim_fma_k(q, x, 6720.);
        q = rim_fma_k(q, x, 20160.);
        q = rim_fma_k(q, x, 40320.);
        return f_factor * (1 + rim_div_by(q * x, 40320., 1. / 40320.));
    }
    // exp(-760) and exp(log|f| - 760) are exactly 0 for |f| < 1e6 (rim_exp returns 0 below -745.2), so
    // both arms of the large-exponent branch below give f_factor * 0: skip its log and exp.  Most
    // samples of a gamma-integral lie far out on the exponentially small side of J_n.
    if (f_exp < (TINY_ZERO ? -690. : -760.) && rim_fabs(f_factor) < 1e6) { RIM_HIT(11); return f_factor * 0.; }
    if (a > 690.) {
        RIM_HIT(12);
        // bessel.c:44-48 takes sign(f) exp(log|f| + e) when log|f| e < 0, else f exp(e).  The product is negative
        // exactly when |f| < 1 and e > 0, or |f| > 1 and e < 0 (log|f| has the sign of |f| - 1, |e| > 690 here, NaN
        // compares false both ways): the logarithm is only taken where its value is used.  For the Meissel prefactor,
        // |f| < 1 and e < 0 -- the second arm -- is the common case.
        const double af = rim_fabs(f_factor);
        if ((af < 1. && f_exp > 0.) || (af > 1. && f_exp < 0.)) {
            const double sign_f = (f_factor < 0) ? -1. : 1.;
            return sign_f * M::exp(rim_log(af) + f_exp);
        }
        return f_factor * M::exp(f_exp);
    }
    RIM_HIT(13);
    return f_factor * M::exp_bounded(f_exp);      // 1e-3 <= |f_exp| <= 690 here (or NaN, which stays NaN)
}

// 10^t to single precision (relative error ~1e-6; only ever used behind a 1e-3 guard band, so the
// host and device versions need not agree bit for bit).
RIM_DEV double guard_pow10(double t)
{
    const float a = (float) (t * 3.321928094887362);
#if defined(__HIP_DEVICE_COMPILE__)
    return (double) __builtin_amdgcn_exp2f(a);
#else
    return (double) __builtin_exp2f(a);
#endif
}

// Order-only quantities of pkgw_bessel_j / BesselJ_Meissel_First.
struct LeungOrder {
    double n;
    double thr_lo, thr_hi;      // -0.6666666 log10 n + {0.174857, 0.295966}   (x < n side)
    double thr_plus_lo;         // -0.6666666 log10 n + 0.151550               (x > n side)
    // The same three thresholds in the linear domain (10^thr, from a single-precision exp2) with a
    // relative guard band of 1e-3 pushed outwards: a ratio outside [r_*_dn, r_*_up] is on the same
    // side of the threshold as its log10 (which is what bessel.c compares) and needs no logarithm.
    double r_lo_dn, r_hi_up, rp_dn, rp_up;
    double ninv, np1inv;        // 1 / n, 1 / (n + 1)
    double vsum2;               // -(ninv (420 + (-14 + (-4 + 3 t2) t2) t2)) / 5040
    double lgam;                // lgamma(n)
    double small_eps_const;     // 0.5 log(0.5 n / pi) + loggamma_exp
    bool big_n;                 // n > 1e3
};

RIM_DEV LeungOrder leung_order(double n)
{
    LeungOrder o;
    o.n = n;
    const double logn = rim_log10(n);
    o.thr_lo = -0.6666666 * logn + 0.174857;
    o.thr_hi = -0.6666666 * logn + 0.295966;
    o.thr_plus_lo = -0.6666666 * logn + 0.151550;
    o.r_lo_dn = guard_pow10(o.thr_lo) * (1. - 1e-3);
    o.r_hi_up = guard_pow10(o.thr_hi) * (1. + 1e-3);
    o.rp_dn = guard_pow10(o.thr_plus_lo) * (1. - 1e-3);
    o.rp_up = guard_pow10(o.thr_plus_lo) * (1. + 1e-3);
    o.ninv = 1. / n;
    o.np1inv = 1. / (n + 1.);
    const double t2 = o.ninv * o.ninv;
    o.vsum2 = -(o.ninv * rim_fma(rim_fma(rim_fma(3., t2, -4.), t2, -14.), t2, 420.)) / 0.5040e4;
    o.lgam = rim_lgamma_pos(n);
    const double t3 = t2 * t2;
    const double loggamma_exp = (o.ninv * (-420 + 14 * t2 - 4 * t3 + 3 * t3 * t2)) / 0.5040e4;
    o.small_eps_const = 0.5 * rim_log(0.5 * n / RIM_PI) + loggamma_exp;
    o.big_n = n > 1e3;
    return o;
}

// Horner over the eight coefficient rows of the Meissel-first V_n sum
// (Chishtie et al. 2005, as tabulated in bessel.c:108-118).
template <int PREC = 0, bool TINY_ZERO = false>
RIM_DEV double meissel_first(const LeungOrder &o, double x)
{
    RIM_HIT(6);
    const double n = o.n;
    const double z = rim_div_by(x, n, o.ninv);
    const double eps = rim_div_by(n - x, n, o.ninv);
    // Z = sqrt(eps (1 + z)), U = 1 / (n Z^3) and factor = 1 / ((n + 1) sqrt(Z)) from one inverse fourth root
    // (eps in [2^-53, 1) since x < n here: 1e-8 < Z < 1.42)
    double Z, inv_z3, inv_sqrt_z;
    rim_meissel_roots(eps * (1 + z), &Z, &inv_z3, &inv_sqrt_z);
    const double U = o.ninv * inv_z3;
    const double t = z * z;

    // each coefficient row is folded into the Horner value in U as soon as it is formed (one row live at a time)
    double v, ak;
    v = rim_fma_k(59968440., t, 4450158720.);
    v = rim_fma_k(v, t, 38435160960.);
    v = rim_fma_k(v, t, 86387857920.);
    v = rim_fma_k(v, t, 60631119360.);
    v = rim_fma_k(v, t, 12841758720.);
    v = rim_fma_k(v, t, 625766400.);
    v = rim_fma_k(v, t, 2580480.);
    v = rim_fma_k(v, t, 0.);                          // a7
    ak = rim_fma_k(-16907985., t, -954875250.);
    ak = rim_fma_k(ak, t, -5897669400.);
    ak = rim_fma_k(ak, t, -8653594320.);
    ak = rim_fma_k(ak, t, -3405435264.);
    ak = rim_fma_k(ak, t, -299351808.);
    ak = rim_fma_k(ak, t, -2644992.);
    ak = rim_fma_k(ak, t, 6144.);                     // a6
    v = rim_fma(v, U, ak);
    ak = rim_fma_k(5537280., t, 228049920.);
    ak = rim_fma_k(ak, t, 940423680.);
    ak = rim_fma_k(ak, t, 800163840.);
    ak = rim_fma_k(ak, t, 138700800.);
    ak = rim_fma_k(ak, t, 2580480.);
    ak = rim_fma_k(ak, t, 0.);                        // a5
    v = rim_fma(v, U, ak);
    ak = rim_fma_k(-2163168., t, -61254720.);
    ak = rim_fma_k(ak, t, -151828480.);
    ak = rim_fma_k(ak, t, -60518400.);
    ak = rim_fma_k(ak, t, -2519040.);
    ak = rim_fma_k(ak, t, -8192.);                    // a4
    v = rim_fma(v, U, ak);
    ak = rim_fma_k(-1048320., t, -18708480.);
    ak = rim_fma_k(ak, t, -23224320.);
    ak = rim_fma_k(ak, t, -2580480.);
    ak = rim_fma_k(ak, t, 0.);                        // a3
    v = rim_fma(v, U, ak);
    ak = rim_fma_k(672000., t, 6547968.);
    ak = rim_fma_k(ak, t, 2709504.);
    ak = rim_fma_k(ak, t, -28672.);                   // a2
    v = rim_fma(v, U, ak);
    ak = rim_fma_k(-645120., t, -2580480.);
    ak = rim_fma_k(ak, t, 0.);                        // a1
    v = rim_fma(v, U, ak);
    ak = rim_fma_k(1290240., t, 860160.);             // a0
    v = rim_fma(v, U, ak);
    const double vsum1 = rim_div_by(U * v, 0.10321920e8, 1. / 0.10321920e8);

    const double factor = o.np1inv * inv_sqrt_z;

    double exp_val;
    if (eps < 1e-4 && o.big_n) {
        RIM_HIT(7);
        double q = rim_fma_k(0.139204065e9, eps, 0.160692840e9);
        q = rim_fma_k(q, eps, 0.190139040e9);
        q = rim_fma_k(q, eps, 0.233192960e9);
        q = rim_fma_k(q, eps, 0.303114240e9);
        q = rim_fma_k(q, eps, 0.442810368e9);
        q = rim_fma_k(q, eps, 0.984023040e9);
        const double exp2 = rim_div_by(-n * rim_sqrt(2. * eps) * eps * q, 0.1476034560e10, 1. / 0.1476034560e10);
        exp_val = o.small_eps_const + exp2 - vsum1 - o.vsum2;
    } else {
        double invZp1;
        if (Z < 1.e-3) {
            RIM_HIT(8);
            double q = 1 - Z;
            q = rim_fma_k(q, Z, -1.);
            q = rim_fma_k(q, Z, 1.);
            q = rim_fma_k(q, Z, -1.);
            q = rim_fma_k(q, Z, 1.);
            q = rim_fma_k(q, Z, -1.);
            q = rim_fma_k(q, Z, 1.);
            invZp1 = q;
        } else {
            invZp1 = rim_div_moderate(1., 1. + Z);
        }
        RIM_HIT(9);
        RIM_PROF_T(t_ml);
        // x * invZp1 is a positive normal number except for x = +0 or a subnormal x; there rim_log_normal gives
        // about -709 instead of -inf / -720, and either way n >= 30 makes exp_val < -2e4: exp_factor returns
        // f_factor * 0 in both cases, so the value of J is the same
        exp_val = n * (rim_log_normal(x * invZp1) - (1 - Z)) - vsum1 - o.vsum2 - o.lgam;
        RIM_PROF_ADD(15, t_ml);
    }
    RIM_PROF_T(t_me);
    const double mres = exp_factor<PREC, TINY_ZERO>(factor, exp_val);
    RIM_PROF_ADD(16, t_me);
    return mres;
}

// Debye epsilon expansion (bessel.c:159-213): degree-15 polynomial in ez = x - n.
RIM_DEV double debye_eps(double n, double x)
{
    if (x > 1.e55) return RIM_NAN;

    const double ez = x - n;
    RIM_PROF_T(t_dp);
    const double z = rim_cbrt_normal(x);      // bessel.c:180 pow(x, 1./3.); x > 0: the Debye band lies next to x = n >= 30
    RIM_PROF_ADD(17, t_dp);
    const double t3 = z * z;
    const double t4 = x * z;
    const double t10 = t4 * t4;
    const double t146 = t10 * t10;
    const double K38 = 810485676000000. * RIM_AT4;

    double q, p;
    // innermost first: d13, then d12 .. d0 folded into the running Horner value
    p = rim_fma(14875. * RIM_AT15, ez * ez, rim_fma_k(3123750. * RIM_AT13, t3, -833000. * RIM_AT15));
    p = rim_fma(p, ez, (40608750. * RIM_AT12) * x);                                       // d12
    p = rim_fma(p, ez, rim_fma_k(-113704500. * RIM_AT13, t3, 17481100. * RIM_AT15));        // d11
    q = rim_fma_k(5360355000. * RIM_AT10, t3, -1161410250. * RIM_AT12);
    p = rim_fma(p, ez, q * x);                                                            // d10
    q = 53603550000. * RIM_AT9;
    q = rim_fma_k(q, t4, 1474097625. * RIM_AT13);
    p = rim_fma(p, ez, rim_fma_k(q, t3, -173573400. * RIM_AT15));                           // d9
    q = rim_fma_k(-88445857500. * RIM_AT10, t3, 11448186750. * RIM_AT12);
    p = rim_fma(p, ez, q * x);                                                            // d8
    q = rim_fma_k(3859455600000. * RIM_AT7, t3, -643242600000. * RIM_AT9);
    q = rim_fma_k(q, t4, -8397889500. * RIM_AT13);
    p = rim_fma(p, ez, rim_fma_k(q, t3, 849093050. * RIM_AT15));                            // d7
    q = 27016189200000. * RIM_AT6;
    q = rim_fma_k(q, t4, 459918459000. * RIM_AT10);
    q = rim_fma_k(q, t3, -47153256150. * RIM_AT12);
    p = rim_fma(p, ez, q * x);                                                            // d6
    q = rim_fma_k(-21612951360000. * RIM_AT7, t3, 2283511230000. * RIM_AT9);
    q = rim_fma_k(q, t4, 20997160275. * RIM_AT13);
    p = rim_fma(p, ez, rim_fma_k(q, t3, -1938419560. * RIM_AT15));                          // d5
    q = rim_fma_k(K38, t3, -94556662200000. * RIM_AT6);
    q = rim_fma_k(q, t4, -860873013000. * RIM_AT10);
    q = rim_fma_k(q, t3, 78248884350. * RIM_AT12);
    p = rim_fma(p, ez, q * x);                                                            // d4
    q = 3241942704000000. * RIM_AT3;
    q = rim_fma_k(q, t4, 29331862560000. * RIM_AT7);
    q = rim_fma_k(q, t3, -2594411820000. * RIM_AT9);
    q = rim_fma_k(q, t4, -19964735910. * RIM_AT13);
    p = rim_fma(p, ez, rim_fma_k(q, t3, 1748257220. * RIM_AT15));                           // d3
    q = rim_fma_k(-K38, t3, 67540473000000. * RIM_AT6);
    q = rim_fma_k(q, t4, 484040056500. * RIM_AT10);
    q = rim_fma_k(q, t3, -41423013450. * RIM_AT12);
    p = rim_fma(p, ez, q * x);                                                            // d2
    q = 19451656224000000. * RIM_AT1;
    q = rim_fma_k(q, t3, -1296777081600000. * RIM_AT3);
    q = rim_fma_k(q, t4, -8027667648000. * RIM_AT7);
    q = rim_fma_k(q, t3, 8027667648000. * RIM_AT9);
    q = rim_fma_k(q, z, -36011689560. * RIM_AT10);
    q = rim_fma_k(q, t3, 3012121710. * RIM_AT12);
    q = rim_fma_k(q, z, 4707059994. * RIM_AT13);
    p = rim_fma(p, ez, rim_fma_k(q, t3, -401283384. * RIM_AT15));                           // d1
    q = 19451656224000000. * RIM_AT0;
    q = rim_fma_k(q, t4, 69470200800000. * RIM_AT4);
    q = rim_fma_k(q, t3, -5403237840000. * RIM_AT6);
    p = rim_fma(p, ez, q * t10 * z);                                                      // d0

    return p / (RIM_PI * t146 * 0.58354968672000000e17);
}

// The same for two orders at one x (J_n and J_{n+1} of a sample): everything except ez = x - n -- the cube root
// and the thirteen coefficient polynomials in x^(1/3) -- is shared, each coefficient being consumed by both
// Horner recurrences as soon as it is formed (two accumulators live instead of thirteen hoisted coefficients).
// Per order the operations and their order are those of debye_eps.
template <int PREC = 0>
RIM_DEV void debye_eps_pair(double n0, double n1, double x, double *r0, double *r1)
{
    if (x > 1.e55) { *r0 = RIM_NAN; *r1 = RIM_NAN; return; }
    RIM_HIT(5);

    const double ez0 = x - n0, ez1 = x - n1;
    RIM_PROF_T(t_dp);
    const double z = RimMath<PREC>::cbrt_normal(x);   // x > 0: the Debye band lies next to x = n >= 30
    RIM_PROF_ADD(17, t_dp);
    const double t3 = z * z;
    const double t4 = x * z;
    const double t10 = t4 * t4;
    const double t146 = t10 * t10;
    const double K38 = 810485676000000. * RIM_AT4;

    double q, c, p0, p1;
    // innermost first: d13, then d12 .. d0 folded into the running Horner value
    c = rim_fma_k(3123750. * RIM_AT13, t3, -833000. * RIM_AT15);
    p0 = rim_fma(14875. * RIM_AT15, ez0 * ez0, c);
    p1 = rim_fma(14875. * RIM_AT15, ez1 * ez1, c);
    c = (40608750. * RIM_AT12) * x;                                       // d12
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);
    c = rim_fma_k(-113704500. * RIM_AT13, t3, 17481100. * RIM_AT15);        // d11
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);
    q = rim_fma_k(5360355000. * RIM_AT10, t3, -1161410250. * RIM_AT12);
    c = q * x;                                                            // d10
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);
    q = 53603550000. * RIM_AT9;
    q = rim_fma_k(q, t4, 1474097625. * RIM_AT13);
    c = rim_fma_k(q, t3, -173573400. * RIM_AT15);                           // d9
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);
    q = rim_fma_k(-88445857500. * RIM_AT10, t3, 11448186750. * RIM_AT12);
    c = q * x;                                                            // d8
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);
    q = rim_fma_k(3859455600000. * RIM_AT7, t3, -643242600000. * RIM_AT9);
    q = rim_fma_k(q, t4, -8397889500. * RIM_AT13);
    c = rim_fma_k(q, t3, 849093050. * RIM_AT15);                            // d7
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);
    q = 27016189200000. * RIM_AT6;
    q = rim_fma_k(q, t4, 459918459000. * RIM_AT10);
    q = rim_fma_k(q, t3, -47153256150. * RIM_AT12);
    c = q * x;                                                            // d6
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);
    q = rim_fma_k(-21612951360000. * RIM_AT7, t3, 2283511230000. * RIM_AT9);
    q = rim_fma_k(q, t4, 20997160275. * RIM_AT13);
    c = rim_fma_k(q, t3, -1938419560. * RIM_AT15);                          // d5
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);
    q = rim_fma_k(K38, t3, -94556662200000. * RIM_AT6);
    q = rim_fma_k(q, t4, -860873013000. * RIM_AT10);
    q = rim_fma_k(q, t3, 78248884350. * RIM_AT12);
    c = q * x;                                                            // d4
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);
    q = 3241942704000000. * RIM_AT3;
    q = rim_fma_k(q, t4, 29331862560000. * RIM_AT7);
    q = rim_fma_k(q, t3, -2594411820000. * RIM_AT9);
    q = rim_fma_k(q, t4, -19964735910. * RIM_AT13);
    c = rim_fma_k(q, t3, 1748257220. * RIM_AT15);                           // d3
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);
    q = rim_fma_k(-K38, t3, 67540473000000. * RIM_AT6);
    q = rim_fma_k(q, t4, 484040056500. * RIM_AT10);
    q = rim_fma_k(q, t3, -41423013450. * RIM_AT12);
    c = q * x;                                                            // d2
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);
    q = 19451656224000000. * RIM_AT1;
    q = rim_fma_k(q, t3, -1296777081600000. * RIM_AT3);
    q = rim_fma_k(q, t4, -8027667648000. * RIM_AT7);
    q = rim_fma_k(q, t3, 8027667648000. * RIM_AT9);
    q = rim_fma_k(q, z, -36011689560. * RIM_AT10);
    q = rim_fma_k(q, t3, 3012121710. * RIM_AT12);
    q = rim_fma_k(q, z, 4707059994. * RIM_AT13);
    c = rim_fma_k(q, t3, -401283384. * RIM_AT15);                           // d1
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);
    q = 19451656224000000. * RIM_AT0;
    q = rim_fma_k(q, t4, 69470200800000. * RIM_AT4);
    q = rim_fma_k(q, t3, -5403237840000. * RIM_AT6);
    c = q * t10 * z;                                                      // d0
    p0 = rim_fma(p0, ez0, c);
    p1 = rim_fma(p1, ez1, c);

    const double den = RIM_PI * t146 * 0.58354968672000000e17;
    // Both quotients have the same divisor.  30 <= x < 1e15 in the integrator (the only caller), so den = 1.8e17 x^(16/3)
    // is a normal number below 1e98, and the quotients are Bessel values of the transition region (~ n^(-1/3)) or
    // exactly 0: moderate operands, for which the bare division sequence returns the bits of `/` (detmath.h).  Its
    // first five instructions -- the refined reciprocal -- depend on den only and are shared by the compiler.
    *r0 = rim_div_moderate(p0, den);
    *r1 = rim_div_moderate(p1, den);
}

// pkgw_bessel_j for n >= 30 given the hoisted order data, in three steps:
//   leung_select    the region logic of bessel.c:336-375 -> which expansions this x needs
//   debye_eps / meissel_first
//   leung_combine   the value bessel.c would return (incl. the linear blend)
// so that the integrand can run the Debye expansion of J_n and J_{n+1} together (debye_eps_pair).
struct LeungSel {
    bool need_debye, need_meissel, blend, unsupported, nan;
    double pos;
};

RIM_DEV LeungSel leung_select(const LeungOrder &o, double x)
{
    const double n = o.n;
    LeungSel s;
    s.need_debye = false; s.need_meissel = false; s.blend = false; s.unsupported = false; s.nan = false;
    s.pos = 0.;
    if (!(x >= 0)) { s.nan = true; return s; }
    if (x == n) {
        s.need_debye = true;
    } else if (x < n) {
        const double r = rim_div_by(n - x, n, o.ninv);
        if (r < o.r_lo_dn) s.need_debye = true;
        else if (r > o.r_hi_up) s.need_meissel = true;
        else {
            RIM_HIT(4);
            const double eta = rim_log10_region(r);
            if (eta < o.thr_lo) s.need_debye = true;
            else if (eta > o.thr_hi) s.need_meissel = true;
            else {
                RIM_HIT(18);
                s.need_debye = true; s.need_meissel = true; s.blend = true;
                s.pos = (eta - o.thr_lo) / (0.295966 - 0.174857);
            }
        }
    } else {
        RIM_HIT(17);
        const double r = (x - n) / x;
        if (r < o.rp_dn) s.need_debye = true;
        else if (r > o.rp_up) s.unsupported = true;      // Meissel "second" region: off the hot path
        else {
            const double eta = rim_log10_region(r);
            if (eta < o.thr_plus_lo) s.need_debye = true;
            else s.unsupported = true;
        }
    }
    return s;
}

// The same decision as a small integer, for the integrand (no struct of flags to keep per order):
//   bit 0 Debye expansion needed, bit 1 Meissel "first" needed, bit 2 linear blend of the two (pos = weight of
//   Meissel), bit 3 no value (NaN argument, or the Meissel "second" region, which the integrator cannot reach).
enum { LSEL_DEBYE = 1, LSEL_MEISSEL = 2, LSEL_BLEND = 4, LSEL_NONE = 8 };

RIM_DEV int leung_select_code(const LeungOrder &o, double x, double &pos)
{
    const double n = o.n;
    pos = 0.;
    if (!(x >= 0)) return LSEL_NONE;
    if (x < n) {
        const double r = rim_div_by(n - x, n, o.ninv);
        if (r < o.r_lo_dn) return LSEL_DEBYE;
        if (r > o.r_hi_up) return LSEL_MEISSEL;
        RIM_HIT(4);
        const double eta = rim_log10_region(r);
        if (eta < o.thr_lo) return LSEL_DEBYE;
        if (eta > o.thr_hi) return LSEL_MEISSEL;
        RIM_HIT(18);
        // (eta - thr_lo) / (0.295966 - 0.174857): a literal divisor, the exact 3-operation division
        pos = rim_div_by(eta - o.thr_lo, 0.295966 - 0.174857, 1. / (0.295966 - 0.174857));
        return LSEL_DEBYE | LSEL_MEISSEL | LSEL_BLEND;
    }
    if (x == n) return LSEL_DEBYE;
    RIM_HIT(17);
    const double r = (x - n) / x;
    if (r < o.rp_dn) return LSEL_DEBYE;
    if (r > o.rp_up) return LSEL_NONE;
    return rim_log10_region(r) < o.thr_plus_lo ? LSEL_DEBYE : LSEL_NONE;
}

RIM_DEV double leung_combine_code(int code, double pos, double debye, double meissel1)
{
    if (code & LSEL_NONE) return RIM_NAN;
    if (code & LSEL_BLEND) return debye * (1 - pos) + meissel1 * pos;
    return (code & LSEL_DEBYE) ? debye : meissel1;
}

RIM_DEV double leung_combine(const LeungSel &s, double debye, double meissel1)
{
    if (s.nan || s.unsupported) return RIM_NAN;
    if (s.blend) return debye * (1 - s.pos) + meissel1 * s.pos;
    return s.need_debye ? debye : meissel1;
}

// Scalar form: a wave whose lanes straddle a region boundary executes both bodies under exec masks,
// which is also what the blend zone needs.
RIM_DEV double leung_j(const LeungOrder &o, double x)
{
    const LeungSel s = leung_select(o, x);
    double debye = 0., meissel1 = 0.;
    if (s.need_debye) debye = debye_eps(o.n, x);
    if (s.need_meissel) meissel1 = meissel_first(o, x);
    return leung_combine(s, debye, meissel1);
}

// ---- integer orders below 30 ----------------------------------------------

RIM_DEV int miller_start(double x)
{
    const double m = (x > 32.) ? x : 32.;
    const int N = (int) (m + 24. + 4.5 * rim_sqrt(m));
    return N + (N & 1);
}

// J_n(x) and J_{n+1}(x) for integer 0 <= n <= 30 from one backward pass.
RIM_DEV void jn_int_pair(int n, double x, double *jn, double *jnp1)
{
    if (!(x >= 0)) { *jn = RIM_NAN; *jnp1 = RIM_NAN; return; }
    if (x == 0.) { *jn = (n == 0) ? 1. : 0.; *jnp1 = 0.; return; }
    if (x > 5.0e4) { *jn = RIM_NAN; *jnp1 = RIM_NAN; return; }

    const double hx = 0.5 * x;
    const bool tiny0 = x * x < 1.0e-16 * (n + 1);
    const bool tiny1 = x * x < 1.0e-16 * (n + 2);
    double s0 = 1., s1 = 1.;
    if (tiny0 || tiny1) {
        for (int k = 1; k <= n; k++) s0 = s0 * (hx / k);
        for (int k = 1; k <= n + 1; k++) s1 = s1 * (hx / k);
        if (tiny0 && tiny1) { *jn = s0; *jnp1 = s1; return; }
    }

    const int N = miller_start(x);
    const double tox = 2. / x;
    double jp1 = 0., j = 1e-300, sum = 0., w0 = 0., w1 = 0.;
    for (int k = N; k >= 1; k--) {
        const double jm1 = rim_fma(k * tox, j, -jp1);
        jp1 = j;
        j = jm1;
        if (rim_fabs(j) > 1e250) {
            j *= 1e-250; jp1 *= 1e-250; sum *= 1e-250; w0 *= 1e-250; w1 *= 1e-250;
        }
        if (k - 1 == n) w0 = j;
        if (k - 1 == n + 1) w1 = j;
        if (((k - 1) & 1) == 0 && k - 1 > 0) sum += 2. * j;
    }
    sum += j;
    *jn = tiny0 ? s0 : w0 / sum;
    *jnp1 = tiny1 ? s1 : w1 / sum;
}

// J_n(x) for integer 0 <= n < 30 and x > 5e4, beyond the reach of the backward recurrence above: Hankel's
// asymptotic expansion (x >> n^2 here, the terms fall off like (n^2 / 2x)^k).  GSL's gsl_sf_bessel_Jn covers
// these arguments in the reference (bessel.c:327-333); the integrator never needs them (x < n + 1 <= 30).
RIM_DEV double jn_hankel(int n, double x)
{
    const double mu = 4. * (double) n * (double) n;
    const double inv8x = 1. / (8. * x);
    double term = 1., p = 1., q = 0.;
    for (int k = 1; k <= 40; k++) {
        const double odd = (double) (2 * k - 1);
        const double next = term * ((mu - odd * odd) * inv8x / (double) k);
        if (rim_fabs(next) >= rim_fabs(term) && k > 1) break;      // the asymptotic series has started to diverge
        term = next;
        // terms alternate between Q (k odd) and P (k even), with signs + - - + + - - + ...
        if (k & 1) q = q + (((k >> 1) & 1) ? -term : term);
        else p = p + (((k >> 1) & 1) ? -term : term);
        if (rim_fabs(term) < 1e-18) break;
    }
    // chi = x - (2n + 1) pi / 4: rotate (cos x, sin x) by an exact multiple of pi/4
    double sx, cx;
    rim_sincos(x, &sx, &cx);
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
    return rim_sqrt(2. / (RIM_PI * x)) * (p * cchi - q * schi);
}

// Meissel's "second" expansion, x > n (bessel.c:57-88, Chishtie et al. 2005).  The integrator never gets here
// (z/n < 1 on the whole path); it completes the Bessel seam for arbitrary arguments.  The reference evaluates
// the phase n (Z - acos(n/x)) and its cosine in `long double`; here they are fp64 (rim_acos01, rim_sincos), so
// the phase carries an absolute error of about n * 1e-15 -- the one place where this library and the reference
// differ by more than rounding (oracle/rimo_bessel.c keeps acosl/cosl in its libm build).
RIM_DEV double meissel_second(double n, double x)
{
    const double z = x / n;
    const double eps = (x - n) / n;
    const double Z = rim_sqrt(eps * (1 + z));
    const double U = 1. / (n * Z * Z * Z);
    const double t1 = z * z;
    const double t2 = U * U;

    double p3 = rim_fma_k(71391., t1, 5297808.);
    p3 = rim_fma_k(p3, t1, 45756144.);
    p3 = rim_fma_k(p3, t1, 102842688.);
    p3 = rim_fma_k(p3, t1, 72179904.);
    p3 = rim_fma_k(p3, t1, 15287808.);
    p3 = rim_fma_k(p3, t1, 744960.);
    p3 = rim_fma_k(p3, t1, 3072.);
    double p2 = rim_fma_k(-6592., t1, -271488.);
    p2 = rim_fma_k(p2, t1, -1119552.);
    p2 = rim_fma_k(p2, t1, -952576.);
    p2 = rim_fma_k(p2, t1, -165120.);
    p2 = rim_fma_k(p2, t1, -3072.);
    double p1 = rim_fma_k(1248., t1, 22272.);
    p1 = rim_fma_k(p1, t1, 27648.);
    p1 = rim_fma_k(p1, t1, 3072.);
    const double p0 = rim_fma_k(-768., t1, -3072.);
    double pe = rim_fma(p3, t2, p2);
    pe = rim_fma(pe, t2, p1);
    pe = rim_fma(pe, t2, p0);
    const double exp_val = (t1 * t2 * pe) / 0.12288e5;

    const double Qt = n * (Z - rim_acos01(n / x));

    double q3 = rim_fma_k(16907985., t1, 954875250.);
    q3 = rim_fma_k(q3, t1, 5897669400.);
    q3 = rim_fma_k(q3, t1, 8653594320.);
    q3 = rim_fma_k(q3, t1, 3405435264.);
    q3 = rim_fma_k(q3, t1, 299351808.);
    q3 = rim_fma_k(q3, t1, 2644992.);
    q3 = rim_fma_k(q3, t1, -6144.);
    double q2 = rim_fma_k(2163168., t1, 61254720.);
    q2 = rim_fma_k(q2, t1, 151828480.);
    q2 = rim_fma_k(q2, t1, 60518400.);
    q2 = rim_fma_k(q2, t1, 2519040.);
    q2 = rim_fma_k(q2, t1, 8192.);
    double q1 = rim_fma_k(-672000., t1, -6547968.);
    q1 = rim_fma_k(q1, t1, -2709504.);
    q1 = rim_fma_k(q1, t1, 28672.);
    const double q0 = rim_fma_k(1290240., t1, 860160.);
    double qs = rim_fma(q3, t2, q2);
    qs = rim_fma(qs, t2, q1);
    qs = rim_fma(qs, t2, q0);
    const double Qsum = -(U * qs) / 0.10321920e8;

    double sn, cs;
    rim_sincos(Qsum + Qt - 0.78539816339744830962, &sn, &cs);
    const double factor = rim_sqrt(2 / (RIM_PI * n * Z)) * cs;
    return exp_factor(factor, exp_val);
}

// Full pkgw_bessel_j / pkgw_bessel_dj for arbitrary (n, x): the scalar seam
// (leung-bessel/src/lib.rs:36-42), used by the batch Bessel entry point.
RIM_DEV double bessel_j(double n, double x)
{
    if (!(n >= 0 && x >= 0)) return RIM_NAN;
    x = x + 0.;     // -0 -> +0: the restricted logarithm of meissel_first reads the sign bit as exponent (the integrator
                    // never passes -0: sym_bessel_pair tests z == 0 first; this is for callers of the seam)
    if (n < 30.) {
        const int n_int = (int) n;
        if (n_int != n) return RIM_NAN;
        if (x > 5.0e4) return jn_hankel(n_int, x);
        double a, b;
        jn_int_pair(n_int, x, &a, &b);
        return a;
    }
    if (x == n) return debye_eps(n, x);
    const LeungOrder o = leung_order(n);
    if (x > n) {
        // bessel.c:358-375: Debye, Meissel-2, or their linear blend (PLUS_ETA_B_INTERCEPT = 0.438914)
        const double eta = rim_log10_region((x - n) / x);
        const double thr_hi = -0.6666666 * rim_log10(n) + 0.438914;
        if (eta < o.thr_plus_lo) return debye_eps(n, x);
        if (eta > thr_hi) return meissel_second(n, x);
        const double debye = debye_eps(n, x);
        const double meissel2 = meissel_second(n, x);
        const double pos = (eta - o.thr_plus_lo) / (0.438914 - 0.151550);
        return debye * (1 - pos) + meissel2 * pos;
    }
    return leung_j(o, x);
}

RIM_DEV double bessel_dj(double n, double x)
{
    if (n >= 1e15) return RIM_NAN;
    const double jn = bessel_j(n, x);
    const double jnp1 = bessel_j(n + 1, x);
    if (x == 0.) {
        if (n >= 2.) return 0.;
        if (n == 0.) return -jnp1;
        return n * jn / RIM_DBL_MIN - jnp1;
    }
    return n * jn / x - jnp1;
}

}  // namespace rim
#endif
