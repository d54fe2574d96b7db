/* detmath.h -- deterministic fp64 elementary functions (C99 / C++ / HIP).
 *
 * Why this exists.  The reference hot path (src/symphony.rs:196-295) takes
 * control-flow decisions on noise-level quantities (a finite difference of an
 * eps_rel=1e-3 quadrature, symphony.rs:238-246).  A last-bit difference in
 * exp/log/pow between glibc (CPU) and ocml (GPU) can flip such a decision and
 * move the result by ~1e-5.  To make "GPU == CPU oracle" a bit-for-bit
 * statement instead of a statistical one, every transcendental used on the
 * path is written here using only IEEE-754 correctly-rounded primitives
 * (+ - * / sqrt fma, rint, ldexp), which gcc/x86-64 and hipcc/gfx950 evaluate
 * identically when both are built with -ffp-contract=off.  The same header is
 * compiled into the HIP kernels (rimphony_amd/csrc) and into the CPU oracle
 * (oracle/), so the two can only differ in control flow and data movement --
 * exactly what the parity tests are meant to check.  The functions themselves
 * are validated against mpmath/libm in tests/test_detmath.py (<= 1.5 ulp).
 *
 * All routines are plain Taylor/Stirling series with exactly representable
 * rational coefficients; no tables, no memory traffic.
 */
#ifndef RIM_DETMATH_H
#define RIM_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RIM_FN __host__ __device__ static inline
#else
#define RIM_FN static inline
#endif

/* Region timers of the diagnostic build (-DRIM_PROF, tools/region_profile.py); no-ops otherwise.
 * -DRIM_PROF -DRIM_PROF_COUNTS turns the timers off and counts instead how often a wave enters the places marked
 * RIM_HIT(idx) (counted by the first ACTIVE lane, so a branch that any lane takes counts once per wave):
 * execution frequencies to weight the static instruction counts of tools/isa_mix.py with. */
#if defined(RIM_ISA_MARKS) && defined(__HIP_DEVICE_COMPILE__)
/* -DRIM_ISA_MARKS (hipcc -S only): the region timers become comments in the listing, for tools/isa_regions.py */
#define RIM_PROF_T(t) asm volatile("; REGION_BEGIN " #t)
#define RIM_PROF_ADD(idx, t) asm volatile("; REGION_END " #t)
#define RIM_PROF_COUNT(idx, v)
#define RIM_HIT(idx)
#define RIM_LANES(idx, cond)
#elif defined(RIM_PROF) && defined(__HIP_DEVICE_COMPILE__)
extern __shared__ unsigned long long rim_prof_lds[];      /* 32 words of dynamic LDS per workgroup */
#if defined(RIM_PROF_COUNTS)
#define RIM_PROF_T(t)
#define RIM_PROF_ADD(idx, t)
#define RIM_PROF_COUNT(idx, v)
#define RIM_HIT(idx) do { const unsigned long long rim_ex_ = __builtin_amdgcn_read_exec(); \
    if (__builtin_amdgcn_mbcnt_hi((unsigned) (rim_ex_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) rim_ex_, 0u)) == 0u) \
        __hip_atomic_fetch_add(&rim_prof_lds[idx], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
/* lanes (of the active ones) on which cond holds, summed per wave */
#define RIM_LANES(idx, cond) do { const unsigned long long rim_ex_ = __builtin_amdgcn_read_exec(); \
    const unsigned long long rim_b_ = __builtin_amdgcn_ballot_w64(cond); \
    if (__builtin_amdgcn_mbcnt_hi((unsigned) (rim_ex_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) rim_ex_, 0u)) == 0u) \
        __hip_atomic_fetch_add(&rim_prof_lds[idx], (unsigned long long) __builtin_popcountll(rim_b_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
#else
#define RIM_LANES(idx, cond)
#define RIM_PROF_T(t) const unsigned long long t = __builtin_readcyclecounter()
#define RIM_PROF_ADD(idx, t) do { if ((threadIdx.x & 63) == 0) \
    __hip_atomic_fetch_add(&rim_prof_lds[idx], __builtin_readcyclecounter() - t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
#define RIM_PROF_COUNT(idx, v) do { if ((threadIdx.x & 63) == 0) \
    __hip_atomic_fetch_add(&rim_prof_lds[idx], (unsigned long long) (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
#define RIM_HIT(idx)
#endif
#else
#define RIM_PROF_T(t)
#define RIM_PROF_ADD(idx, t)
#define RIM_PROF_COUNT(idx, v)
#define RIM_HIT(idx)
#define RIM_LANES(idx, cond)
#endif

#define RIM_NAN (__builtin_nan(""))
#define RIM_INF (__builtin_inf())
#define RIM_PI 3.14159265358979323846
#define RIM_DBL_EPSILON 2.2204460492503131e-16
#define RIM_DBL_MIN 2.2250738585072014e-308

RIM_FN uint64_t rim_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
RIM_FN double rim_frombits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
RIM_FN double rim_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

/* fma(a, b, K) for a compile-time constant K (Horner steps).  Same value as rim_fma.
 * On gfx950 hipcc lowers fma-with-literal to a 2-address v_fmac_f64 preceded by two
 * v_mov_b32 that copy the 64-bit literal into the destination (VOP3 cannot encode
 * 64-bit literals): 3 VALU issues per Horner step.  Spelling the 3-address form keeps
 * K in an SGPR pair (materialised by the scalar unit, which runs beside the VALU):
 * 1 VALU issue per step. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double rim_fma_k(double a, double b, double k)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
#else
#define rim_fma_k(a, b, k) rim_fma(a, b, k)
#endif
/* fma(a, K, 1) for a compile-time constant K: K in an SGPR pair, the 1 an inline constant (1 VALU issue) */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double rim_fma_k1(double a, double k)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, 1.0" : "=v"(r) : "v"(a), "s"(k));
    return r;
}
#else
#define rim_fma_k1(a, k) rim_fma(a, k, 1.0)
#endif
/* a / b given binv = RN(1 / b): the correctly rounded quotient in 3 operations instead of the
 * ~11-instruction IEEE division sequence (Markstein: q = RN(a binv), r = a - q b exactly by FMA,
 * RN(q + r binv) = RN(a / b)).  For finite a, normal b and a quotient in the normal range -- the
 * divisors it is used with are harmonic numbers >= 30 and literals.  tests/test_detmath.py checks
 * it against the division operator. */
RIM_FN double rim_div_by(double a, double b, double binv)
{
    const double q = a * binv;
    const double r = rim_fma(-q, b, a);
    return rim_fma(r, binv, q);
}

/* num / den for operands of moderate magnitude.  hipcc expands an fp64 division into v_div_scale (x2), v_rcp,
 * two Newton steps, a quotient, its residual, v_div_fmas and v_div_fixup: 11 VALU issues.  The scale / fixup
 * instructions only act when an operand is zero, infinite, NaN or subnormal, or when the exponents are within
 * a few binades of the range limits; otherwise they pass their inputs through and the quotient is the result
 * of the 8 arithmetic instructions, which this function issues bare.  It is therefore bit-identical to `/`
 * whenever num, den and num / den are normal numbers far from the range limits (and NaN in, NaN out) -- it is
 * used only where that is certain by construction (each call site says why), never on Bessel values or
 * quadrature sums, which do underflow. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double rim_div_moderate(double num, double den)
{
    double y = __builtin_amdgcn_rcp(den);
    double e = __builtin_fma(-den, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-den, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double q = num * y;
    const double r = __builtin_fma(-den, q, num);
    return __builtin_fma(r, y, q);
}
#else
RIM_FN double rim_div_moderate(double num, double den) { return num / den; }
#endif

/* sqrt.  On the device hipcc expands __builtin_sqrt into v_rsq_f64 + a Goldschmidt/Newton sequence wrapped in
 * input scaling (applied when x < 2^-767) and special-value selects: 18 VALU issues per call, ~8 calls per
 * integrand sample.  The bare sequence below is that same expansion without the scaling (13 issues): it returns
 * the correctly rounded root -- bit for bit what the CPU's sqrt returns -- for +-0, +inf, NaN, negative x (NaN)
 * and, measured over 6e5 random arguments, every x >= 2^-1009; only for positive x next to and inside the
 * subnormal range (x < 1e-303) does the missing scaling show (up to 35 ulp).  No radicand of the path gets
 * there without being exactly 0 (they are 1 - t^2, eps (1 + z), differences of squares and ratios of
 * quadrature estimates); tests/test_detmath.py checks both statements
 * on the device.  A guarded version (exponent test + library fallback) was measured 7 % SLOWER than the library
 * expansion because of the extra control flow at every call site. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double rim_sqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return (x == 0. || x == __builtin_inf()) ? x : g;
}
#else
RIM_FN double rim_sqrt(double x) { return __builtin_sqrt(x); }
#endif
RIM_FN double rim_fabs(double x) { return __builtin_fabs(x); }
RIM_FN double rim_floor(double x) { return __builtin_floor(x); }
RIM_FN int rim_isnan(double x) { return x != x; }
RIM_FN int rim_isfinite(double x) { return (rim_bits(x) & 0x7ff0000000000000ull) != 0x7ff0000000000000ull; }
RIM_FN double rim_max(double a, double b) { return a > b ? a : b; } /* GSL_MAX_DBL semantics */
RIM_FN double rim_min(double a, double b) { return a < b ? a : b; }

/* x * 2^k, exact unless the result is subnormal (then one rounding). */
RIM_FN double rim_ldexp(double x, int k) { return __builtin_ldexp(x, k); }

/* ---- exp ------------------------------------------------------------- */

/* exp(xh + xl) for |xl| << |xh|; one ulp class accuracy (< 0.6 ulp). */
/* exp(xh + xl) for |xh| <= 700 (no NaN, overflow or underflow to look for): the arithmetic of rim_exp_dd. */
RIM_FN double rim_exp_dd_core(double xh, double xl)
{
    /* k = nearest integer to x / ln 2 */
    const double kd = __builtin_rint(xh * 1.44269504088896338700e+00);
    const int k = (int) kd;
    /* ln2 = LN2_HI + LN2_LO; LN2_HI has 21 trailing zero bits so kd*LN2_HI is exact */
    double r = rim_fma(-kd, 6.93147180369123816490e-01, xh);
    r = rim_fma(-kd, 1.90821492927058770002e-10, r);
    r = r + xl;

    /* exp(r) = 1 + r + r^2 * (1/2 + r/6 + ... + r^11/13!),  |r| <= 0.3466 */
    double q = 1.0 / 6227020800.0;              /* 1/13! */
    q = rim_fma_k(q, r, 1.0 / 479001600.0);    /* 1/12! */
    q = rim_fma_k(q, r, 1.0 / 39916800.0);
    q = rim_fma_k(q, r, 1.0 / 3628800.0);
    q = rim_fma_k(q, r, 1.0 / 362880.0);
    q = rim_fma_k(q, r, 1.0 / 40320.0);
    q = rim_fma_k(q, r, 1.0 / 5040.0);
    q = rim_fma_k(q, r, 1.0 / 720.0);
    q = rim_fma_k(q, r, 1.0 / 120.0);
    q = rim_fma_k(q, r, 1.0 / 24.0);
    q = rim_fma_k(q, r, 1.0 / 6.0);
    q = rim_fma_k(q, r, 0.5);
    const double t = rim_fma(q * r, r, r);     /* r + r^2 q */
    const double p = 1.0 + t;
    return rim_ldexp(p, k);
}


RIM_FN double rim_exp_dd(double xh, double xl)
{
    if (rim_isnan(xh)) return xh;
    if (xh > 709.782712893384) return RIM_INF;
    if (xh < -745.2) return 0.0;

    return rim_exp_dd_core(xh, xl);
}

RIM_FN double rim_exp(double x) { return rim_exp_dd(x, 0.0); }
/* exp(x) for |x| <= 700: same bits as rim_exp there */
RIM_FN double rim_exp_bounded(double x) { return rim_exp_dd_core(x, 0.0); }

/* ---- log ------------------------------------------------------------- */

/* log(x) as an unevaluated sum hi + lo with ~2^-60 relative accuracy and
 * hi = round-to-nearest(hi + lo).  Domain handling: x < 0 -> NaN, 0 -> -inf. */
RIM_FN double rim_log_dd(double x, double *lo)
{
    *lo = 0.0;
    if (rim_isnan(x)) return x;
    if (x < 0.0) return RIM_NAN;
    if (x == 0.0) return -RIM_INF;
    if (!rim_isfinite(x)) return x;

    int k = 0;
    uint64_t u = rim_bits(x);
    if ((u >> 52) == 0) {            /* subnormal: scale by 2^54 */
        x = x * 18014398509481984.0;
        u = rim_bits(x);
        k = -54;
    }
    k += (int) (u >> 52) - 1023;
    u = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = rim_frombits(u);      /* [1, 2) */
    if (m > 1.4142135623730951) { m = 0.5 * m; k += 1; }   /* [0.7071, 1.4142] */

    const double f = m - 1.0;        /* exact */
    /* s = f / (2 + f) in double-double */
    const double th = 2.0 + f;
    const double tl = (2.0 - th) + f;            /* exact (Fast2Sum, 2 >= |f|) */
    const double rcp = rim_div_moderate(1.0, th);     /* th in [1.70, 2.42] */
    const double sh = f * rcp;
    double res = rim_fma(-sh, th, f);
    res = rim_fma(-sh, tl, res);
    const double sl = res * rcp;

    /* atanh(s)/s - 1 = z/3 + z^2/5 + ... + z^11/23,  z = s^2 <= 0.0295 */
    const double z = sh * sh;
    double q = 1.0 / 23.0;
    q = rim_fma_k(q, z, 1.0 / 21.0);
    q = rim_fma_k(q, z, 1.0 / 19.0);
    q = rim_fma_k(q, z, 1.0 / 17.0);
    q = rim_fma_k(q, z, 1.0 / 15.0);
    q = rim_fma_k(q, z, 1.0 / 13.0);
    q = rim_fma_k(q, z, 1.0 / 11.0);
    q = rim_fma_k(q, z, 1.0 / 9.0);
    q = rim_fma_k(q, z, 1.0 / 7.0);
    q = rim_fma_k(q, z, 1.0 / 5.0);
    q = rim_fma_k(q, z, 1.0 / 3.0);
    q = q * z;

    /* log m = 2 sh + (2 sl + 2 sh q) */
    const double a = 2.0 * sh;
    const double c = rim_fma(a, q, 2.0 * sl);

    /* + k ln2, ln2 = LN2_HI + LN2_LO with k*LN2_HI exact */
    const double kd = (double) k;
    const double kh = kd * 6.93147180369123816490e-01;
    const double kl = kd * 1.90821492927058770002e-10;
    /* TwoSum(kh, a) */
    const double s1 = kh + a;
    const double bb = s1 - kh;
    const double e1 = (kh - (s1 - bb)) + (a - bb);
    const double low = e1 + (c + kl);
    const double hi = s1 + low;
    *lo = (s1 - hi) + low;
    return hi;
}

/* The same for a positive, finite, NORMAL x (no special values, no subnormal rescaling): identical bits there. */
RIM_FN double rim_log_dd_normal(double x, double *lo)
{
    int k = 0;
    uint64_t u = rim_bits(x);
    k += (int) (u >> 52) - 1023;
    u = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = rim_frombits(u);      /* [1, 2) */
    if (m > 1.4142135623730951) { m = 0.5 * m; k += 1; }   /* [0.7071, 1.4142] */

    const double f = m - 1.0;        /* exact */
    /* s = f / (2 + f) in double-double */
    const double th = 2.0 + f;
    const double tl = (2.0 - th) + f;            /* exact (Fast2Sum, 2 >= |f|) */
    const double rcp = rim_div_moderate(1.0, th);     /* th in [1.70, 2.42] */
    const double sh = f * rcp;
    double res = rim_fma(-sh, th, f);
    res = rim_fma(-sh, tl, res);
    const double sl = res * rcp;

    /* atanh(s)/s - 1 = z/3 + z^2/5 + ... + z^11/23,  z = s^2 <= 0.0295 */
    const double z = sh * sh;
    double q = 1.0 / 23.0;
    q = rim_fma_k(q, z, 1.0 / 21.0);
    q = rim_fma_k(q, z, 1.0 / 19.0);
    q = rim_fma_k(q, z, 1.0 / 17.0);
    q = rim_fma_k(q, z, 1.0 / 15.0);
    q = rim_fma_k(q, z, 1.0 / 13.0);
    q = rim_fma_k(q, z, 1.0 / 11.0);
    q = rim_fma_k(q, z, 1.0 / 9.0);
    q = rim_fma_k(q, z, 1.0 / 7.0);
    q = rim_fma_k(q, z, 1.0 / 5.0);
    q = rim_fma_k(q, z, 1.0 / 3.0);
    q = q * z;

    /* log m = 2 sh + (2 sl + 2 sh q) */
    const double a = 2.0 * sh;
    const double c = rim_fma(a, q, 2.0 * sl);

    /* + k ln2, ln2 = LN2_HI + LN2_LO with k*LN2_HI exact */
    const double kd = (double) k;
    const double kh = kd * 6.93147180369123816490e-01;
    const double kl = kd * 1.90821492927058770002e-10;
    /* TwoSum(kh, a) */
    const double s1 = kh + a;
    const double bb = s1 - kh;
    const double e1 = (kh - (s1 - bb)) + (a - bb);
    const double low = e1 + (c + kl);
    const double hi = s1 + low;
    *lo = (s1 - hi) + low;
    return hi;
}


RIM_FN double rim_log(double x) { double lo; return rim_log_dd(x, &lo); }
/* log of a positive normal finite x (same bits as rim_log there; for +0 and subnormals it returns about -709.1
 * instead of -inf / the true value -- see the one call site for why that is harmless there) */
RIM_FN double rim_log_normal(double x) { double lo; return rim_log_dd_normal(x, &lo); }

RIM_FN double rim_log10(double x)
{
    double lo;
    const double hi = rim_log_dd(x, &lo);
    if (!rim_isfinite(hi)) return hi;
    /* 1/ln(10) = C_HI + C_LO */
    const double C_HI = 4.34294481903251816668e-01;
    const double C_LO = 1.09831965021676507340e-17;
    const double p = hi * C_HI;
    const double e = rim_fma(hi, C_HI, -p);
    return p + (e + rim_fma(hi, C_LO, lo * C_HI));
}

/* log10 of a positive, finite, normal x in plain double arithmetic (<= 4 ulp; tests/test_detmath.py).
 * Used for the Bessel region variable eta = log10((n - x) / n) (bessel.c:341-357), which is only compared
 * with thresholds and turned into a blend weight: the double-double tail of rim_log10 buys nothing there
 * and costs a third of the function.  Oracle (deterministic build) and kernels share it. */
RIM_FN double rim_log10_region(double x)
{
    uint64_t u = rim_bits(x);
    int k = (int) (u >> 52) - 1023;
    u = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = rim_frombits(u);      /* [1, 2) */
    if (m > 1.4142135623730951) { m = 0.5 * m; k += 1; }   /* [0.7071, 1.4142] */
    const double f = m - 1.0;        /* exact */
    const double s = rim_div_moderate(f, 2.0 + f);    /* |f| <= 0.42 (f = 0 gives 0 either way), 2 + f in [1.70, 2.42] */
    const double z = s * s;
    /* atanh(s)/s - 1 = z/3 + z^2/5 + ... + z^11/23 */
    double q = 1.0 / 23.0;
    q = rim_fma_k(q, z, 1.0 / 21.0);
    q = rim_fma_k(q, z, 1.0 / 19.0);
    q = rim_fma_k(q, z, 1.0 / 17.0);
    q = rim_fma_k(q, z, 1.0 / 15.0);
    q = rim_fma_k(q, z, 1.0 / 13.0);
    q = rim_fma_k(q, z, 1.0 / 11.0);
    q = rim_fma_k(q, z, 1.0 / 9.0);
    q = rim_fma_k(q, z, 1.0 / 7.0);
    q = rim_fma_k(q, z, 1.0 / 5.0);
    q = rim_fma_k(q, z, 1.0 / 3.0);
    const double a = 2.0 * s;
    const double ln_m = rim_fma(a, q * z, a);                       /* ln m = 2 s (1 + q z) */
    /* k log10(2) + ln(m) / ln(10) */
    return rim_fma((double) k, 0.30102999566398120, ln_m * 0.43429448190325182);
}

/* ---- pow ------------------------------------------------------------- */

RIM_FN double rim_pow(double x, double y)
{
    if (y == 0.0) return 1.0;
    if (x == 1.0) return 1.0;
    if (rim_isnan(x) || rim_isnan(y)) return RIM_NAN;

    double sign = 1.0;
    if (x < 0.0 || (x == 0.0 && (rim_bits(x) >> 63))) {
        /* negative base: only integer exponents are defined */
        const double yi = rim_floor(y);
        const int y_is_int = (yi == y) && rim_isfinite(y);
        if (x != 0.0 && !y_is_int) {
            if (!rim_isfinite(y)) {
                /* (-a)^(+-inf) */
                const double ax = -x;
                if (ax == 1.0) return 1.0;
                return ((ax > 1.0) == (y > 0.0)) ? RIM_INF : 0.0;
            }
            return RIM_NAN;
        }
        if (y_is_int && rim_fabs(y) < 9007199254740992.0) {
            const double half = 0.5 * y;
            if (rim_floor(half) != half) sign = -1.0;
        }
        x = -x;
    }

    if (x == 0.0) return (y > 0.0) ? sign * 0.0 : sign * RIM_INF;
    if (!rim_isfinite(x)) return (y > 0.0) ? sign * RIM_INF : sign * 0.0;
    if (!rim_isfinite(y)) {
        if (x == 1.0) return 1.0;
        return ((x > 1.0) == (y > 0.0)) ? RIM_INF : 0.0;
    }

    double ll;
    const double lh = rim_log_dd(x, &ll);
    const double ph = y * lh;
    const double pl = rim_fma(y, lh, -ph) + y * ll;
    if (ph > 800.0) return sign * RIM_INF;
    if (ph < -800.0) return sign * 0.0;
    return sign * rim_exp_dd(ph, pl);
}

/* x^1.5 for x >= 0 as x * sqrt(x) (two roundings; used by the GK error rescaling, qk.c) */
RIM_FN double rim_pow15(double x) { return x * rim_sqrt(x); }
/* x^2.5 as x^2 * sqrt(x) (three roundings) */
RIM_FN double rim_pow25(double x) { return (x * x) * rim_sqrt(x); }

/* x^y from a double-double log of x that the caller already holds (lh + ll = rim_log_dd(x)): the last two steps
 * of rim_pow_pos, for callers that raise one x to several powers. */
RIM_FN double rim_pow_from_log(double lh, double ll, double y)
{
    const double ph = y * lh;
    const double pl = rim_fma(y, lh, -ph) + y * ll;
    if (ph > 800.0) return RIM_INF;
    if (ph < -800.0) return 0.0;
    return rim_exp_dd(ph, pl);
}

/* rim_pow(x, y) for a positive, finite x and a finite y: the same three steps (double-double log, product,
 * exp of the sum) without the dozen special-case tests in front of them, hence the same bits on that domain
 * (x = 1 and y = 0 still give exactly 1: log 1 = 0 and exp 0 = 1 are exact here).  tests/test_detmath.py. */
RIM_FN double rim_pow_pos(double x, double y)
{
    double ll;
    const double lh = rim_log_dd(x, &ll);
    const double ph = y * lh;
    const double pl = rim_fma(y, lh, -ph) + y * ll;
    if (ph > 800.0) return RIM_INF;
    if (ph < -800.0) return 0.0;
    return rim_exp_dd(ph, pl);
}

/* ... and for a positive NORMAL finite x (no zero, subnormal, infinity or NaN to look for). */
RIM_FN double rim_pow_normal(double x, double y)
{
    double ll;
    const double lh = rim_log_dd_normal(x, &ll);
    const double ph = y * lh;
    const double pl = rim_fma(y, lh, -ph) + y * ll;
    /* rim_pow's two range tests and rim_exp_dd's own two collapse into these (same outcome for every ph) */
    if (ph > 709.782712893384) return RIM_INF;
    if (ph < -745.2) return 0.0;
    return rim_exp_dd_core(ph, pl);
}

/* y^(-1/4) for a positive normal finite y, ~1 ulp (tests/test_detmath.py), from integer and correctly rounded fp64
 * operations only (bit-identical on the CPU and the GPU): the exponent trick (3.1 %), two third-order steps
 * w <- w (1 + a + 5/2 a^2) with a = (1 - y w^4) / 4, and a first-order one.  The Meissel expansion needs
 * Z = sqrt(y), 1 / Z^3 and 1 / sqrt(Z) of one y (bessel.c:100-122): all three are products of powers of w, where the
 * literal evaluation takes two square roots and two divisions. */
RIM_FN double rim_rqrt4_normal(double y)
{
    const uint32_t hi = (uint32_t) (rim_bits(y) >> 32);
    double w = rim_frombits((uint64_t) (0x4feb0c00u - (hi >> 2)) << 32);
    const double y4 = 0.25 * y;
    for (int it = 0; it < 2; it++) {
        const double w2 = w * w;
        const double a = rim_fma_k(-y4, w2 * w2, 0.25);
        w = rim_fma(w * a, rim_fma_k1(a, 2.5), w);
    }
    const double w2 = w * w;
    const double a = rim_fma_k(-y4, w2 * w2, 0.25);
    return rim_fma(w, a, w);
}

/* The three roots of the Meissel expansion from one y = eps (1 + z) in [2^-53, 2) (bessel.c:100-122: Z = sqrt(y),
 * U = 1 / (n Z^3), factor = 1 / ((n + 1) sqrt(Z))): w = y^(-1/4) gives 1 / sqrt(Z) = w, 1 / Z^3 = w^6 and
 * Z = y w^2, the latter polished by one Newton step to within an ulp because the exponent n (log(x / (1 + Z)) - (1 - Z))
 * multiplies its error by n Z^2.  22 fp64 operations where two square roots and two divisions take 42. */
RIM_FN void rim_meissel_roots(double y, double *Z, double *inv_z3, double *inv_sqrt_z)
{
    const double w = rim_rqrt4_normal(y);
    const double w2 = w * w;
    const double z0 = y * w2;
    *Z = rim_fma(rim_fma(-z0, z0, y) * w2, 0.5, z0);
    *inv_z3 = (w2 * w2) * w2;
    *inv_sqrt_z = w;
}

/* x^y exp(e) for a positive normal finite x as ONE exponential, exp(y ln x + e) with the sum carried in double-double:
 * the gamma^-p exp(-gamma / gamma_cutoff) of the distribution functions (power_law.rs:51-56, pitchy_pl.rs:74-80,
 * pitchy_kappa.rs:66-70 write powf(..) * exp(..)).  Within an ulp of the product of the two correctly rounded factors
 * and a third cheaper than rim_pow_normal(x, y) * rim_exp(e); where the separate factors would over- or underflow
 * on their own and the product not, this returns the product.  Oracle (deterministic build) and kernels share it. */
/* ... its two halves: the double-double logarithm of x, and everything after it.  rim_powexp_normal(x, y, e) ==
 * rim_powexp_from_log(lh, ll, y, e) with (lh, ll) = rim_log_dd_normal(x): the same operations, the same bits -- for callers
 * that need x^y1 e^e and x^y2 e^e of one x (the emission and absorption terms of one sample, symphony_group.h). */
RIM_FN double rim_powexp_from_log(double lh, double ll, double y, double e)
{
    const double ph = y * lh;
    const double pl = rim_fma(y, lh, -ph) + y * ll;
    const double s = ph + e;                      /* two-sum of ph and e */
    const double bv = s - ph;
    const double sl = ((ph - (s - bv)) + (e - bv)) + pl;
    if (s > 709.782712893384) return RIM_INF;
    if (s < -745.2) return 0.0;
    return rim_exp_dd_core(s, sl);
}

RIM_FN double rim_powexp_normal(double x, double y, double e)
{
    double ll;
    const double lh = rim_log_dd_normal(x, &ll);
    return rim_powexp_from_log(lh, ll, y, e);
}

/* x^(1/3) for a positive normal finite x, < 1 ulp (tests/test_detmath.py), in 22 operations where the general power
 * takes 89.  Every step is an integer operation or a correctly rounded fp64 one, so the CPU and the GPU agree bit for
 * bit: y ~ x^(-1/3) from the exponent trick (3.5 %), two third-order steps y <- y (1 + a + 2 a^2), a = (1 - x y^3) / 3
 * (2e-4, then 5e-11), z0 = x y^2, and one Newton step on z0 with the residual x - z0^3 taken by FMA.
 * bessel.c:180 calls pow(x, 1./3.), itself a rounding error away from the cube root: this is the deterministic
 * build's stand-in (oracle and kernels share it). */
RIM_FN double rim_cbrt_normal(double x)
{
    const uint32_t hi = (uint32_t) (rim_bits(x) >> 32);
    const uint32_t third = (uint32_t) (((uint64_t) hi * 0xAAAAAAABull) >> 33);         /* hi / 3 */
    double y = rim_frombits((uint64_t) (0x553ef000u - third) << 32);
    const double x3 = x * (1. / 3.);
    for (int it = 0; it < 2; it++) {
        const double c = y * y;
        const double a = rim_fma_k(-(x3 * y), c, 1. / 3.);
        y = rim_fma(y * a, rim_fma(a, 2., 1.), y);
    }
    const double c = y * y;
    const double z0 = x * c;
    const double r = rim_fma(-(z0 * z0), z0, x);
    return rim_fma_k(r * c, 1. / 3., z0);
}

/* h^(2/3), h^(-2/3), h^(1/3), h^(-1/3) for h >= 0 (the prefactors (x/2)^nu of the four modified Bessel functions of a
 * quasi-resonant Faraday sample, heyvaerts.rs:312-326,409-418) from ONE cube root and one division, where four
 * powers take a double-double logarithm and four exponentials.  +0 gives {0, inf, 0, inf} as pow does; a negative or
 * NaN argument NaN.  A few ulp (the squares); oracle (deterministic build) and kernels share it. */
RIM_FN void rim_third_powers(double h, double out[4])
{
    double a;
    if (h >= 2.2250738585072014e-308) a = rim_cbrt_normal(h);
    else if (h > 0.) a = rim_cbrt_normal(h * 1.532495540865889e+54) * 8.673617379884035e-19;      /* 2^180, 2^-60 */
    else a = (h == 0.) ? 0. : RIM_NAN;
    const double ainv = 1. / a;
    out[0] = a * a;
    out[1] = ainv * ainv;
    out[2] = a;
    out[3] = ainv;
}
/* 1 / Gamma(1 + nu) for nu = 2/3, -2/3, 1/3, -1/3, correctly rounded */
#define RIM_RGAMMA_5_3 1.1077321674324725
#define RIM_RGAMMA_1_3 0.3732821739073952
#define RIM_RGAMMA_4_3 1.1198465217221858
#define RIM_RGAMMA_2_3 0.7384881116216483

/* x^(4/3) (the physical limit of the quasi-resonant pomega range, heyvaerts.rs:270-272 powf(4./3.)) as x cbrt(x) for a
 * positive normal finite x, through the general power otherwise (0, subnormal, inf, NaN, negative) */
RIM_FN double rim_pow43(double x)
{
    if (x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308) return x * rim_cbrt_normal(x);
    return rim_pow(x, 4. / 3.);
}

/* 1 / Gamma(z) for -8.5 < z < 9.5, ~2 ulp (tests/test_detmath.py): z = m + z0 with m the nearest integer, the Taylor
 * series 1 / Gamma(z0) = z0 S(z0) (|z0| <= 1/2, 20 terms), and the recurrence: m >= 1 gives S / ((z0 + 1) .. (z0 + m - 1)),
 * m <= 0 gives z0 S (z0 - 1) .. (z0 + m) -- exactly 0 at the poles.  The J_nu prefactors of the Faraday elements
 * ((x/2)^nu / Gamma(nu + 1), nu = +-sigma, +-(sigma - 1) with sigma < ~4) divide by a Gamma function per series; the
 * shift-up-and-Stirling evaluation it replaces there took ~170 operations, this ~50. */
#define RIM_RGAMMA_NEAR(z) ((z) > -8.5 && (z) < 9.5)
/* S(z0) of the Taylor series 1 / Gamma(z0) = z0 S(z0), |z0| <= 1/2 (S_0 .. S_19) */
RIM_FN double rim_rgamma_s(double z0)
{
    double s = 7.782263439905071e-12;
    s = rim_fma_k(s, z0, 1.0434267116911005e-10);
    s = rim_fma_k(s, z0, -1.18127457048702e-09);
    s = rim_fma_k(s, z0, 5.002007644469223e-09);
    s = rim_fma_k(s, z0, 6.116095104481416e-09);
    s = rim_fma_k(s, z0, -2.056338416977607e-07);
    s = rim_fma_k(s, z0, 1.133027231981696e-06);
    s = rim_fma_k(s, z0, -1.2504934821426706e-06);
    s = rim_fma_k(s, z0, -2.013485478078824e-05);
    s = rim_fma_k(s, z0, 0.0001280502823881162);
    s = rim_fma_k(s, z0, -0.00021524167411495098);
    s = rim_fma_k(s, z0, -0.0011651675918590652);
    s = rim_fma_k(s, z0, 0.0072189432466631);
    s = rim_fma_k(s, z0, -0.009621971527876973);
    s = rim_fma_k(s, z0, -0.04219773455554433);
    s = rim_fma_k(s, z0, 0.16653861138229148);
    s = rim_fma_k(s, z0, -0.04200263503409524);
    s = rim_fma_k(s, z0, -0.6558780715202539);
    s = rim_fma_k(s, z0, 0.5772156649015329);
    s = rim_fma_k(s, z0, 1.0);
    return s;
}
/* 1 / Gamma(m + z0) from S = S(z0) by the recurrence, m an integer-valued double: m >= 1 gives S / ((z0 + 1) .. (z0 + m - 1)),
 * m <= 0 gives z0 S (z0 - 1) .. (z0 + m) -- exactly 0 at the poles */
RIM_FN double rim_rgamma_shift(double s, double z0, double m)
{
    if (m >= 1.) {
        double p = 1.;
        for (double k = 1.; k < m; k += 1.) p = p * (z0 + k);
        return s / p;
    }
    double p = z0;
    for (double k = -1.; k >= m; k -= 1.) p = p * (z0 + k);
    return s * p;
}
RIM_FN double rim_rgamma_near(double z)
{
    const double m = __builtin_rint(z);
    const double z0 = z - m;                       /* exact */
    return rim_rgamma_shift(rim_rgamma_s(z0), z0, m);
}
/* The four reciprocal Gamma values in front of the J series of a quasi-resonant Faraday sample on its J/Y branch
 * (heyvaerts.rs:335-336, 359-363, 437-441: besselj / bessely of orders sigma and sigma - 1, and by the reflection formula
 * of Y their negatives): out = { 1/Gamma(sigma + 1), 1/Gamma(sigma), 1/Gamma(2 - sigma), 1/Gamma(1 - sigma) } for
 * -7 < sigma < 8.  Two Taylor sums -- the arguments sigma + 1, sigma share their fractional part z0, and 2 - sigma,
 * 1 - sigma the part -z0 -- two recurrences, and Gamma(z + 1) = z Gamma(z) for the other member of each pair (where the
 * four separate evaluations took four sums and four recurrences).  A few ulp (tests/test_detmath.py). */
RIM_FN void rim_rgamma_quad(double sigma, double out[4])
{
    const double m = __builtin_rint(sigma);
    const double z0 = sigma - m;                   /* exact */
    out[0] = rim_rgamma_shift(rim_rgamma_s(z0), z0, m + 1.);
    out[1] = out[0] * sigma;
    out[2] = rim_rgamma_shift(rim_rgamma_s(-z0), -z0, 2. - m);
    out[3] = out[2] * (1. - sigma);
}

/* ---- atan, acos (only the Meissel "second" expansion of the Bessel seam needs them) -------------- */

/* atan(u) for u >= 0: reflection for u > 1, three half-angle steps atan u = 2 atan(u / (1 + sqrt(1 + u^2)))
 * down to u < 0.1, then the alternating series to u^21.  A few ulp. */
RIM_FN double rim_atan_pos(double u)
{
    if (rim_isnan(u)) return u;
    int reflect = 0;
    if (u > 1.0) { u = 1.0 / u; reflect = 1; }       /* also maps +inf to 0 */
    u = u / (1.0 + rim_sqrt(rim_fma(u, u, 1.0)));
    u = u / (1.0 + rim_sqrt(rim_fma(u, u, 1.0)));
    u = u / (1.0 + rim_sqrt(rim_fma(u, u, 1.0)));
    const double z = u * u;
    double q = -1.0 / 23.0;
    q = rim_fma_k(q, z, 1.0 / 21.0);
    q = rim_fma_k(q, z, -1.0 / 19.0);
    q = rim_fma_k(q, z, 1.0 / 17.0);
    q = rim_fma_k(q, z, -1.0 / 15.0);
    q = rim_fma_k(q, z, 1.0 / 13.0);
    q = rim_fma_k(q, z, -1.0 / 11.0);
    q = rim_fma_k(q, z, 1.0 / 9.0);
    q = rim_fma_k(q, z, -1.0 / 7.0);
    q = rim_fma_k(q, z, 1.0 / 5.0);
    q = rim_fma_k(q, z, -1.0 / 3.0);
    const double a = 8.0 * rim_fma(u * z, q, u);
    return reflect ? 1.5707963267948966 - a + 6.123233995736766e-17 : a;
}

/* acos(t) for 0 <= t <= 1 as 2 atan(sqrt((1 - t) / (1 + t))): no cancellation as t -> 1 */
RIM_FN double rim_acos01(double t)
{
    return 2.0 * rim_atan_pos(rim_sqrt((1.0 - t) / (1.0 + t)));
}

/* ---- lgamma (positive arguments only) --------------------------------- */

/* Stirling series for x >= 16 (error of the truncated series < 1e-19 rel). */
RIM_FN double rim_lgamma_stirling(double x)
{
    double ll;
    const double lh = rim_log_dd(x, &ll);
    /* (x - 1/2) * log x in double-double */
    const double xm = x - 0.5;                    /* exact for x >= 1 below 2^52 */
    const double ph = xm * lh;
    const double pl = rim_fma(xm, lh, -ph) + xm * ll;
    /* asymptotic tail: sum B_2k / (2k (2k-1) x^(2k-1)), k = 1..7 */
    const double w = 1.0 / x;
    const double w2 = w * w;
    double t = 1.0 / 156.0;
    t = rim_fma_k(t, w2, -691.0 / 360360.0);
    t = rim_fma_k(t, w2, 1.0 / 1188.0);
    t = rim_fma_k(t, w2, -1.0 / 1680.0);
    t = rim_fma_k(t, w2, 1.0 / 1260.0);
    t = rim_fma_k(t, w2, -1.0 / 360.0);
    t = rim_fma_k(t, w2, 1.0 / 12.0);
    t = t * w;
    /* ph - x + (pl + t + 0.5 log(2 pi)) */
    const double HALF_LOG_2PI = 9.18938533204672741780e-01;
    const double d = ph - x;
    const double bv = d - ph;
    const double e = (ph - (d - bv)) + (-x - bv);     /* TwoSum(ph, -x) */
    return d + (e + (pl + (t + HALF_LOG_2PI)));
}

RIM_FN double rim_lgamma_pos(double x)
{
    if (rim_isnan(x) || x <= 0.0) return RIM_NAN;
    if (!rim_isfinite(x)) return x;
    if (x >= 16.0) return rim_lgamma_stirling(x);
    /* shift up: Gamma(x) = Gamma(x + m) / (x (x+1) ... (x+m-1)) */
    double prod = 1.0;
    while (x < 16.0) { prod = prod * x; x = x + 1.0; }
    return rim_lgamma_stirling(x) - rim_log(prod);
}

/* ---- sin / cos --------------------------------------------------------- */

/* Accurate (< 1 ulp) for |x| <= 1e5; beyond that the Cody-Waite reduction
 * gradually loses bits (the reference only ever uses observer angles in
 * [0, pi]). */
RIM_FN void rim_sincos(double x, double *sn, double *cs)
{
    if (!rim_isfinite(x)) { *sn = RIM_NAN; *cs = RIM_NAN; return; }
    const double kd = __builtin_rint(x * 6.36619772367581382433e-01); /* 2/pi */
    const long long kq = (long long) kd;
    /* pi/2 = P1 + P2 + P2T; P1, P2 carry 33 significant bits so kd*P1, kd*P2 are exact */
    const double r1 = rim_fma(-kd, 1.57079632673412561417e+00, x);
    const double wa = kd * 6.07710050630396597660e-11;
    double rh = r1 - wa;
    double bv = rh - r1;
    double rl = (r1 - (rh - bv)) + (-wa - bv);      /* TwoSum(r1, -wa) */
    const double wb = kd * 2.02226624879595063154e-21;
    {
        const double t = rh;
        rh = t - wb;
        bv = rh - t;
        rl = rl + ((t - (rh - bv)) + (-wb - bv));
    }
    const double z = rh * rh;
    /* sin(r) = r + r^3 S(z), cos(r) = 1 - z/2 + z^2 C(z) */
    double s = 1.0 / 121645100408832000.0;           /* 1/19! */
    s = rim_fma(s, z, -1.0 / 355687428096000.0);     /* 1/17! */
    s = rim_fma(s, z, 1.0 / 1307674368000.0);        /* 1/15! */
    s = rim_fma_k(s, z, -1.0 / 6227020800.0);
    s = rim_fma_k(s, z, 1.0 / 39916800.0);
    s = rim_fma_k(s, z, -1.0 / 362880.0);
    s = rim_fma_k(s, z, 1.0 / 5040.0);
    s = rim_fma_k(s, z, -1.0 / 120.0);
    s = rim_fma(s, z, 1.0 / 6.0);                    /* note sign folded below */
    double c = -1.0 / 6402373705728000.0;            /* -1/18! */
    c = rim_fma(c, z, 1.0 / 20922789888000.0);       /* 1/16! */
    c = rim_fma_k(c, z, -1.0 / 87178291200.0);
    c = rim_fma_k(c, z, 1.0 / 479001600.0);
    c = rim_fma_k(c, z, -1.0 / 3628800.0);
    c = rim_fma_k(c, z, 1.0 / 40320.0);
    c = rim_fma_k(c, z, -1.0 / 720.0);
    c = rim_fma_k(c, z, 1.0 / 24.0);
    /* sin r = rh - rh z s' with s' = 1/6 - z/120 + ... ; above built with
     * alternating signs so that s = 1/6 - z/120 + z^2/5040 - ... */
    const double sin_r = rim_fma(-(rh * z), s, rh);
    const double hz = 0.5 * z;
    const double wv = 1.0 - hz;
    const double cos_r = wv + (((1.0 - wv) - hz) + (z * z) * c);
    /* first-order correction for the low part of the reduced argument */
    const double sv = rim_fma(rl, cos_r, sin_r);
    const double cv = rim_fma(-rl, sin_r, cos_r);
    switch ((int) (kq & 3)) {
    case 0: *sn = sv; *cs = cv; break;
    case 1: *sn = cv; *cs = -sv; break;
    case 2: *sn = -sv; *cs = -cv; break;
    default: *sn = -cv; *cs = sv; break;
    }
}

/* ---- fp32-core variants (RIMPHONY_PRECISION_F32_INTEGRAND; BASELINE.json configs[4]) --------------------
 * The "fp32 integrand" variant keeps every difference and every product with a large factor in fp64 (the
 * kinematics n - z, eps = (n - x) / n, gamma sin xi, the Meissel exponent n (log .. - (1 - Z))) and evaluates the
 * BODIES of the elementary functions in single precision on the hardware transcendental unit: the exponential of a
 * reduced argument, the logarithm of a mantissa.  Relative error ~1e-7 per call instead of ~1e-16; a quarter to a
 * half of the instructions.  No bit-parity contract: the host forms use libm's exp2f / log2f. */
#if defined(__HIP_DEVICE_COMPILE__)
#define RIM_EXP2F(x) __builtin_amdgcn_exp2f(x)
#define RIM_LOG2F(x) __builtin_amdgcn_logf(x)
#else
#define RIM_EXP2F(x) __builtin_exp2f(x)
#define RIM_LOG2F(x) __builtin_log2f(x)
#endif

/* exp(x): range reduction in fp64 (k = rint(x / ln 2), r = x - k ln 2 exactly as rim_exp does), 2^(r / ln 2) in
 * fp32, scaling by 2^k in fp64: relative error ~1e-7 for every x, however large |x| */
RIM_FN double rim_exp_f32core(double x)
{
    if (rim_isnan(x)) return x;
    if (x > 709.782712893384) return RIM_INF;
    if (x < -745.2) return 0.0;
    const double kd = __builtin_rint(x * 1.44269504088896338700e+00);
    double r = rim_fma(-kd, 6.93147180369123816490e-01, x);
    r = rim_fma(-kd, 1.90821492927058770002e-10, r);
    const float e = RIM_EXP2F((float) (r * 1.44269504088896338700e+00));
    return rim_ldexp((double) e, (int) kd);
}

/* x^y for a positive normal finite x: log2 x = exponent (exact) + log2(mantissa) (fp32), y log2 x split into integer
 * and fraction in fp64, 2^fraction in fp32.  Relative error ~ (1 + |y| |log2 x|) 1e-7. */
RIM_FN double rim_pow_f32core(double x, double y)
{
    uint64_t u = rim_bits(x);
    const int e = (int) (u >> 52) - 1023;
    u = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    const double l2 = (double) e + (double) RIM_LOG2F((float) rim_frombits(u));
    const double p = y * l2;
    if (rim_isnan(p)) return p;
    if (p > 1024.0) return RIM_INF;
    if (p < -1075.0) return 0.0;
    const double kd = __builtin_rint(p);
    const float f = RIM_EXP2F((float) (p - kd));
    return rim_ldexp((double) f, (int) kd);
}

/* x^(1/3) for a positive normal finite x: the fp32-core power refined by one Newton step in fp64
 * (y - (y^3 - x) / (3 y^2) with the quotient's reciprocal in fp32): relative error ~1e-13 */
RIM_FN double rim_cbrt_f32core(double x)
{
    const double y = rim_pow_f32core(x, 1. / 3.);
    const double y2 = y * y;
    const float rc = 1.0f / (float) (3. * y2);
    return rim_fma(-rim_fma(y2, y, -x), (double) rc, y);
}

/* The elementary functions of the integrand by precision variant (0: fp64, the reference's arithmetic; 1: fp32 core) */
#if defined(__cplusplus)
template <int PREC> struct RimMath {
    RIM_FN double exp(double x) { return rim_exp(x); }
    RIM_FN double exp_bounded(double x) { return rim_exp_bounded(x); }
    /* sin_xi^k of the pitch-angle distributions: the base may be 0 (or NaN) and the exponent 0 -- those keep the general
     * function's special cases; everything else is a positive normal base with a finite exponent, where
     * rim_pow_normal returns the same bits for half the instructions (tests/test_detmath.py) */
    RIM_FN double pow(double x, double y)
    {
        if (!(x >= 2.2250738585072014e-308) || !rim_isfinite(x) || y == 0.0 || !rim_isfinite(y)) return rim_pow(x, y);
        return rim_pow_normal(x, y);
    }
    RIM_FN double pow_normal(double x, double y) { return rim_pow_normal(x, y); }
    RIM_FN double cbrt_normal(double x) { return rim_cbrt_normal(x); }              /* bessel.c:180 pow(x, 1./3.) */
    RIM_FN double powexp_normal(double x, double y, double e) { return rim_powexp_normal(x, y, e); }
};
#if defined(RIM_F32_SMOOTH)
/* Measurement build only (tools/build_variant.sh ... -DRIM_F32_SMOOTH; tools/f32_variant.py): the "smooth" fp32 variant VERDICT
 * round 3 asked to have MEASURED -- the fp32-core value as a seed, one fp64 correction step that removes its 1e-7 noise:
 * exp: y = y0 (1 + (x - log y0)); pow: y = y0 (1 + (y log x - log y0)).  The logarithms are the fp64 ones: a correction
 * step is only as good as its residual. */
RIM_FN double rim_exp_smooth(double x)
{
    const double y0 = rim_exp_f32core(x);
    if (!(y0 >= 2.2250738585072014e-308) || !rim_isfinite(y0)) return y0;
    return y0 * (1. + (x - rim_log_normal(y0)));
}
RIM_FN double rim_pow_smooth(double x, double y)
{
    const double y0 = rim_pow_f32core(x, y);
    if (!(y0 >= 2.2250738585072014e-308) || !rim_isfinite(y0)) return y0;
    return y0 * (1. + (y * rim_log_normal(x) - rim_log_normal(y0)));
}
template <> struct RimMath<1> {
    RIM_FN double exp(double x) { return rim_exp_smooth(x); }
    RIM_FN double exp_bounded(double x) { return rim_exp_smooth(x); }
    RIM_FN double pow(double x, double y)
    {
        if (!(x > 2.3e-308) || !rim_isfinite(x) || y == 0.0 || !rim_isfinite(y)) return rim_pow(x, y);
        return rim_pow_smooth(x, y);
    }
    RIM_FN double pow_normal(double x, double y) { return rim_pow_smooth(x, y); }
    RIM_FN double cbrt_normal(double x) { return rim_cbrt_f32core(x); }
    RIM_FN double powexp_normal(double x, double y, double e) { return rim_pow_smooth(x, y) * rim_exp_smooth(e); }
};
#else
template <> struct RimMath<1> {
    RIM_FN double exp(double x) { return rim_exp_f32core(x); }
    RIM_FN double exp_bounded(double x) { return rim_exp_f32core(x); }
    /* sin_xi^k and friends: x may be 0 or the exponent 0 -- keep the general function's special cases */
    RIM_FN double pow(double x, double y)
    {
        if (!(x > 2.3e-308) || !rim_isfinite(x) || y == 0.0 || !rim_isfinite(y)) return rim_pow(x, y);
        return rim_pow_f32core(x, y);
    }
    RIM_FN double pow_normal(double x, double y) { return rim_pow_f32core(x, y); }
    RIM_FN double cbrt_normal(double x) { return rim_cbrt_f32core(x); }
    RIM_FN double powexp_normal(double x, double y, double e) { return rim_pow_f32core(x, y) * rim_exp_f32core(e); }
};
#endif
#endif

#endif /* RIM_DETMATH_H */
