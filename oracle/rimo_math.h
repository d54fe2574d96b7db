/* rimo_math.h -- elementary-function selection for the oracle (test infrastructure).
 * Default: the deterministic functions shared with the HIP kernels.
 * -DRIMO_LIBM: glibc libm, as the reference's Rust/C code would call. */
#ifndef RIMO_MATH_H
#define RIMO_MATH_H

#include <math.h>
#include <float.h>
#include "../rimphony_amd/csrc/detmath.h"

#ifdef RIMO_LIBM
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
#define m_pow43(x) rim_pow43(x)
#define m_pow25(x) rim_pow25(x)
#define m_lgamma(x) rim_lgamma_pos(x)
static inline void m_sincos(double x, double *s, double *c) { rim_sincos(x, s, c); }
#define m_fma(a, b, c) rim_fma(a, b, c)
#endif

#define m_sqrt(x) __builtin_sqrt(x)
#define m_fabs(x) __builtin_fabs(x)

#endif
