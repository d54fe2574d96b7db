/* rimo_math.h -- elementary-function selection for the oracle (test infrastructure).
 * Default: the deterministic functions shared with the HIP kernels.
 * -DRIMO_LIBM: glibc libm, as the reference's Rust/C code would call.
 * -DRIMO_ATTR: the attribution build (liboracle_attr.so, tools/faraday_tail_attribution.py only): the deterministic
 *   flavour with every lock-step reformulation of DESIGN.md section 3 item 5 switchable back to its literal form AT RUN
 *   TIME through rimo_set_attr_mask(), one bit per reformulation -- mask 0 is liboracle.so bit for bit, the full mask is
 *   liboracle_libm.so bit for bit (the tool checks both), and every mask in between says what ONE form costs in
 *   distance from the literal arithmetic. */
#ifndef RIMO_MATH_H
#define RIMO_MATH_H

#include <math.h>
#include <float.h>
#include "../rimphony_amd/csrc/detmath.h"

/* one bit per reformulation (the literal form is what liboracle_libm.so evaluates) */
enum {
    RIMO_ATTR_THIRD    = 1 << 0,   /* (x/2)^nu / Gamma(1+nu) of the four fixed-order I_nu: pow + Gamma, not rim_third_powers + constants */
    RIMO_ATTR_RGAMMA   = 1 << 1,   /* 1/Gamma(nu+1) of the J_nu prefactors: a division by Gamma, not rim_rgamma_near */
    RIMO_ATTR_SERIES   = 1 << 2,   /* fixed-order I_nu series: the term recurrence with a division and a test per term, not Horner on tabulated coefficients */
    RIMO_ATTR_NR       = 1 << 3,   /* h_nr / f_nr: divisions by powers of sigma^2 - x^2, not products of powers of u */
    RIMO_ATTR_POW15    = 1 << 4,   /* x^1.5, x^2.5, x^(4/3) of the elements and range limits through pow(), not x sqrt(x) / x cbrt(x) */
    RIMO_ATTR_ELEM     = 1 << 5,   /* exp, log, pow, lgamma from glibc, not detmath.h */
    RIMO_ATTR_POWEXP   = 1 << 6,   /* gamma^y e^(..) as pow() * exp(), not one exponential of the double-double sum */
    RIMO_ATTR_GKSEQ    = 1 << 7,   /* GK31 sums in qk.c's sequential order, not the wave's tree */
    RIMO_ATTR_RESCALE  = 1 << 8,   /* rescale_error: pow(.., 1.5), not x sqrt(x) */
    RIMO_ATTR_SINCOS   = 1 << 9,   /* sin / cos from glibc, not rim_sincos */
    RIMO_ATTR_UNFUSED  = 1 << 10,  /* Horner steps a * b + c in two roundings (Leung Bessel: Symphony only) */
    RIMO_ATTR_BESSEL   = 1 << 11,  /* Leung Bessel roots / cube root / region log10 in their literal forms (Symphony only) */
    RIMO_ATTR_DINV     = 1 << 12,  /* gamma and the gamma-term of d f / d sigma: divisions by sigma0 sin(theta), not products with its reciprocal */
    RIMO_ATTR_GFORM    = 1 << 13,  /* g = sqrt(8)/3 (sigma - x)^1.5 / sqrt(x) literally, not sqrt(8)/3 (sigma - x) sqrt((sigma - x) / x) */
    RIMO_ATTR_JYFAST   = 1 << 14,  /* J/Y branch: four separate pow / Gamma / term-recurrence evaluations, not the joint one (jy_fast) */
    RIMO_ATTR_ALL      = (1 << 15) - 1
};

#if defined(RIMO_ATTR)
extern unsigned rimo_attr_mask;
#define RIMO_LIT(bit) ((rimo_attr_mask & (bit)) != 0)
#elif defined(RIMO_LIBM)
#define RIMO_LIT(bit) 1
#else
#define RIMO_LIT(bit) 0
#endif

#if defined(RIMO_ATTR)
static inline double m_exp(double x) { return RIMO_LIT(RIMO_ATTR_ELEM) ? exp(x) : rim_exp(x); }
static inline double m_log(double x) { return RIMO_LIT(RIMO_ATTR_ELEM) ? log(x) : rim_log(x); }
static inline double m_log10(double x) { return RIMO_LIT(RIMO_ATTR_ELEM) ? log10(x) : rim_log10(x); }
static inline double m_log10_region(double x) { return RIMO_LIT(RIMO_ATTR_BESSEL) ? log10(x) : rim_log10_region(x); }
static inline double m_acos01(double t) { return RIMO_LIT(RIMO_ATTR_BESSEL) ? (double) acosl((long double) t) : rim_acos01(t); }
static inline double m_cos_phase(double x)
{
    if (RIMO_LIT(RIMO_ATTR_BESSEL)) return (double) cosl((long double) x);
    double s, c; rim_sincos(x, &s, &c); return c;
}
static inline double m_pow(double x, double y) { return RIMO_LIT(RIMO_ATTR_ELEM) ? pow(x, y) : rim_pow(x, y); }
static inline double m_cbrt(double x) { return RIMO_LIT(RIMO_ATTR_BESSEL) ? m_pow(x, 1. / 3.) : rim_cbrt_normal(x); }
static inline double m_pow15(double x) { return RIMO_LIT(RIMO_ATTR_POW15) ? m_pow(x, 1.5) : rim_pow15(x); }
static inline double m_pow15_rescale(double x) { return RIMO_LIT(RIMO_ATTR_RESCALE) ? m_pow(x, 1.5) : rim_pow15(x); }
static inline double m_pow43(double x) { return RIMO_LIT(RIMO_ATTR_POW15) ? m_pow(x, 4. / 3.) : rim_pow43(x); }
static inline double m_pow25(double x) { return RIMO_LIT(RIMO_ATTR_POW15) ? m_pow(x, 2.5) : rim_pow25(x); }
static inline double m_lgamma(double x) { return RIMO_LIT(RIMO_ATTR_ELEM) ? lgamma(x) : rim_lgamma_pos(x); }
static inline void m_sincos(double x, double *s, double *c)
{
    if (RIMO_LIT(RIMO_ATTR_SINCOS)) { *s = sin(x); *c = cos(x); } else rim_sincos(x, s, c);
}
static inline double m_fma(double a, double b, double c) { return RIMO_LIT(RIMO_ATTR_UNFUSED) ? a * b + c : rim_fma(a, b, c); }
#elif defined(RIMO_LIBM)
#define m_exp(x) exp(x)
#define m_log(x) log(x)
#define m_log10(x) log10(x)
#define m_log10_region(x) log10(x)
/* bessel.c:73,84 evaluate these two in long double */
#define m_acos01(t) ((double) acosl((long double) (t)))
#define m_cos_phase(x) ((double) cosl((long double) (x)))
#define m_pow(x, y) pow(x, y)
#define m_cbrt(x) pow(x, 1. / 3.)
#define m_pow15(x) pow(x, 1.5)
#define m_pow15_rescale(x) pow(x, 1.5)
#define m_pow43(x) pow(x, 4. / 3.)
#define m_pow25(x) pow(x, 2.5)
#define m_lgamma(x) lgamma(x)
static inline void m_sincos(double x, double *s, double *c) { *s = sin(x); *c = cos(x); }
#define m_fma(a, b, c) ((a) * (b) + (c)) /* the reference never fuses */
#else
#define m_exp(x) rim_exp(x)
#define m_log(x) rim_log(x)
#define m_log10(x) rim_log10(x)
#define m_log10_region(x) rim_log10_region(x)
#define m_acos01(t) rim_acos01(t)
static inline double m_cos_phase(double x) { double s, c; rim_sincos(x, &s, &c); return c; }
#define m_pow(x, y) rim_pow(x, y)
#define m_cbrt(x) rim_cbrt_normal(x) /* the deterministic build's cube root (detmath.h); its one call site has x ~ n >= 30 */
#define m_pow15(x) rim_pow15(x)
#define m_pow15_rescale(x) rim_pow15(x)
#define m_pow43(x) rim_pow43(x)
#define m_pow25(x) rim_pow25(x)
#define m_lgamma(x) rim_lgamma_pos(x)
static inline void m_sincos(double x, double *s, double *c) { rim_sincos(x, s, c); }
#define m_fma(a, b, c) rim_fma(a, b, c)
#endif

#define m_sqrt(x) __builtin_sqrt(x)
#define m_fabs(x) __builtin_fabs(x)

#endif
