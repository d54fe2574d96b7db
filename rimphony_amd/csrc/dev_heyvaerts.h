// dev_heyvaerts.h -- per-sample device functions of the Heyvaerts Faraday path.
//
//   fill_coord_vars                      heyvaerts.rs:194-201
//   h_qr / h_nr / f_qr / f_nr elements   heyvaerts.rs:302-468
//   dfdsigma                             heyvaerts.rs:472-493
//   special_fun::FloatSpecial::{besseli, besselj, bessely}   call sites heyvaerts.rs:331-364, 437-441
//
// The Bessel functions come from an un-vendored crate in the reference; they are
// evaluated here from their ascending series (see oracle/rimo_heyvaerts.c for why
// that covers every argument the path can produce).  The Gamma-function
// prefactors of the four fixed fractional orders are computed once per task
// (HeyConsts) instead of once per sample.  Operation order matches
// oracle/rimo_heyvaerts.c exactly (bit-for-bit parity is tested).
#ifndef RIM_DEV_HEYVAERTS_H
#define RIM_DEV_HEYVAERTS_H

#include "dev_symphony.h"

namespace rim {

#define RIM_FOUR_OVER_SQRT_27 0.769800358919501
#define RIM_INVERSE_C (1. / 2.99792458e10)
#define RIM_INVERSE_SQRT_3 0.5773502691896257
#define RIM_SQRT_8_OVER_3 0.9428090415820635
#define RIM_THREE_TWO_THIRDS 2.080083823051904
#define RIM_G_APPROXIMATION_CUTOFF 10.

// Rust f64::min / f64::max (a NaN operand is ignored)
RIM_DEV double rust_min(double a, double b) { if (a != a) return b; if (b != b) return a; return a < b ? a : b; }
RIM_DEV double rust_max(double a, double b) { if (a != a) return b; if (b != b) return a; return a > b ? a : b; }

RIM_DEV double gamma_real(double z)
{
    double prod = 1., w = z;
    int guard = 0;
    while (w < 16. && guard < 64) { RIM_HIT(24); prod = prod * w; w = w + 1.; guard++; }
    return rim_exp(rim_lgamma_stirling(w)) / prod;
}

// 1 / Gamma(z): the short-range series-and-recurrence evaluation where it applies (the orders of the J/Y branch lie
// within a few units of 0), else through gamma_real
RIM_DEV double rgamma_real(double z)
{
    if (RIM_RGAMMA_NEAR(z)) return rim_rgamma_near(z);
    return 1. / gamma_real(z);
}

RIM_DEV double ascending_series(double nu, double q, double sign)
{
    double term = 1., sum = 1.;
    for (int k = 1; k <= 500; k++) {
        // q = (x/2)^2 and k (k + nu) are normal numbers of moderate size (a negative order is never an exact
        // integer here: bessel_ynu perturbs those) -> the bare division is exact (detmath.h)
        RIM_HIT(30);
        term = term * rim_div_moderate(sign * q, k * (k + nu));
        sum = sum + term;
        if (rim_fabs(term) < 1e-17 * rim_fabs(sum)) break;
    }
    return sum;
}

// The ascending series of the four modified Bessel functions of a quasi-resonant sample (orders 2/3, -2/3, 1/3, -1/3),
//   S_j(q) = sum_k a[k][j] q^k,   a[k][j] = 1 / (k! (nu_j + 1)_k),   q = (g/2)^2 < 25   (iseries_table.h, generated),
// by Horner's rule -- one FMA per term and order, the four orders side by side -- from a degree that depends on q ALONE:
// degree 6 for every lane, three more terms for the lanes whose q is beyond each threshold (below it the terms left out
// are under 2^-60 of the sum).  A lane that enters at degree n starts from 0, so its first step leaves a[n][j]: the value
// is a function of q and of nothing else -- not of the other lanes of the wave, which only decide which blocks are
// skipped.  The coefficients are wave-uniform constants (scalar loads / the scalar operand of v_fma_f64).  Until round 3
// the terms were built by the recurrence term *= q / (k (k + nu)), two products and a sum per term plus a convergence test
// per pair of terms; Horner is 0.5-0.8 ulp rms from the true sum where the recurrence is 1.3-1.9 (tests/test_detmath.py).
#include "iseries_table.h"
#if defined(__HIP_DEVICE_COMPILE__)
static __constant__ const double RIM_ISERIES[4 * (RIM_ISERIES_NMAX + 1)] = RIM_ISERIES_TABLE;
#else
static const double RIM_ISERIES[4 * (RIM_ISERIES_NMAX + 1)] = RIM_ISERIES_TABLE;
#endif

RIM_DEV double hey_series_order(int j) { return j == 0 ? 2. / 3. : j == 1 ? -2. / 3. : j == 2 ? 1. / 3. : -1. / 3.; }

// Device: the rows of a block are fetched by the scalar unit right where they are used (s_load_dwordx16 / x8 into SGPRs,
// which v_fma_f64 reads as its scalar operand) -- spelled as volatile inline assembly because LLVM otherwise hoists the
// 100 loop-invariant loads to the top of the kernel and parks them in VGPR lanes (v_writelane / v_readlane: two vector
// instructions per constant, measured on the first build of this function).
#if defined(__HIP_DEVICE_COMPILE__)
typedef double rim_d4 __attribute__((ext_vector_type(4)));
typedef double rim_d8 __attribute__((ext_vector_type(8)));
#define RIM_ISERIES_FMA4(r) do { s0 = rim_fma_k(s0, q, (r).x); s1 = rim_fma_k(s1, q, (r).y); s2 = rim_fma_k(s2, q, (r).z); s3 = rim_fma_k(s3, q, (r).w); } while (0)
// rows k, k - 1, k - 2 (12 consecutive doubles from row k - 2)
#define RIM_ISERIES_ROWS3(k) do { \
        rim_d8 lo_; rim_d4 hi_; \
        asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx8 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" \
                     : "=&s"(lo_), "=&s"(hi_) : "s"(tab), "n"(32 * ((k) - 2)), "n"(32 * ((k) - 2) + 64)); \
        RIM_ISERIES_FMA4(hi_); \
        { const rim_d4 r_ = { lo_.s4, lo_.s5, lo_.s6, lo_.s7 }; RIM_ISERIES_FMA4(r_); } \
        { const rim_d4 r_ = { lo_.s0, lo_.s1, lo_.s2, lo_.s3 }; RIM_ISERIES_FMA4(r_); } \
    } while (0)
// rows 3, 2, 1, 0
#define RIM_ISERIES_ROWS4_BASE() do { \
        rim_d8 lo_, hi_; \
        asm volatile("s_load_dwordx16 %0, %2, 0\n\ts_load_dwordx16 %1, %2, 64\n\ts_waitcnt lgkmcnt(0)" \
                     : "=&s"(lo_), "=&s"(hi_) : "s"(tab)); \
        { const rim_d4 r_ = { hi_.s4, hi_.s5, hi_.s6, hi_.s7 }; RIM_ISERIES_FMA4(r_); } \
        { const rim_d4 r_ = { hi_.s0, hi_.s1, hi_.s2, hi_.s3 }; RIM_ISERIES_FMA4(r_); } \
        { const rim_d4 r_ = { lo_.s4, lo_.s5, lo_.s6, lo_.s7 }; RIM_ISERIES_FMA4(r_); } \
        { const rim_d4 r_ = { lo_.s0, lo_.s1, lo_.s2, lo_.s3 }; RIM_ISERIES_FMA4(r_); } \
    } while (0)
#else
#define RIM_ISERIES_STEP(k) do { s0 = rim_fma(s0, q, tab[4 * (k)]); s1 = rim_fma(s1, q, tab[4 * (k) + 1]); \
                                 s2 = rim_fma(s2, q, tab[4 * (k) + 2]); s3 = rim_fma(s3, q, tab[4 * (k) + 3]); } while (0)
#define RIM_ISERIES_ROWS3(k) do { RIM_ISERIES_STEP(k); RIM_ISERIES_STEP((k) - 1); RIM_ISERIES_STEP((k) - 2); } while (0)
#define RIM_ISERIES_ROWS4_BASE() do { RIM_ISERIES_STEP(3); RIM_ISERIES_STEP(2); RIM_ISERIES_STEP(1); RIM_ISERIES_STEP(0); } while (0)
#endif
#define RIM_ISERIES_BLOCK(qmin, k) if (q >= (qmin)) { RIM_HIT(31); RIM_ISERIES_ROWS3(k); }

// tab: the address of RIM_ISERIES as hey_consts() hands it out (an opaque scalar register pair on the device)
RIM_DEV void rim_iseries4(const double *tab, double q, double out[4])
{
    double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
    RIM_ISERIES_BLOCK(RIM_ISERIES_Q21, 24)
    RIM_ISERIES_BLOCK(RIM_ISERIES_Q18, 21)
    RIM_ISERIES_BLOCK(RIM_ISERIES_Q15, 18)
    RIM_ISERIES_BLOCK(RIM_ISERIES_Q12, 15)
    RIM_ISERIES_BLOCK(RIM_ISERIES_Q9, 12)
    RIM_ISERIES_BLOCK(RIM_ISERIES_Q6, 9)
    RIM_ISERIES_ROWS3(6);
    RIM_ISERIES_ROWS4_BASE();
    out[0] = s0; out[1] = s1; out[2] = s2; out[3] = s3;
}

RIM_DEV double bessel_jnu(double nu, double x)
{
    const double h = 0.5 * x;
    return rim_pow_pos(h, nu) * rgamma_real(nu + 1.) * ascending_series(nu, h * h, -1.);   // h >= 0 (x = sqrt(..))
}

RIM_DEV double bessel_ynu(double nu, double x)
{
    double sn, cs;
    if (nu == rim_floor(nu)) nu = nu + 1.4901161193847656e-08 * (rim_fabs(nu) > 1. ? rim_fabs(nu) : 1.);
    rim_sincos(RIM_PI * nu, &sn, &cs);
    return (cs * bessel_jnu(nu, x) - bessel_jnu(-nu, x)) / sn;
}

// J_sigma, Y_sigma, J_{sigma-1} and (if want_ym1) Y_{sigma-1} at x, as x.besselj(nu) / x.bessely(nu) return them
// (heyvaerts.rs:335-336, 359-363, 437-441).  Y_nu = (cos(pi nu) J_nu - J_{-nu}) / sin(pi nu) needs J_nu again, so a
// literal transcription sums the same series twice; here every distinct series is summed once, in one loop
// body: orders sigma, sigma - 1, their negatives, and -- only for the integer orders that bessel_ynu
// perturbs -- the perturbed orders.  Each value is the one bessel_jnu / bessel_ynu return.
//
// Round 4: for a NON-INTEGER order sigma in (0, 8) -- every sample of the path but the measure-zero integer ones, which
// keep the evaluation above (bessel_jy_generic) -- the four series J_sigma, J_(sigma-1), J_(-sigma), J_(1-sigma) are
// evaluated together (bessel_jy_fast), sharing what the four separate evaluations repeated:
//   * ONE power h^sigma (h = x/2; one exponential of the double-double logarithm) and two reciprocals give
//     h^(sigma-1) = h^sigma / h, h^(-sigma) = 1 / h^sigma, h^(1-sigma) = h / h^sigma (four exponentials before);
//   * rim_rgamma_quad: the four 1 / Gamma from two Taylor sums and Gamma(z + 1) = z Gamma(z) (four sums before);
//   * ONE sincos: sin(pi (sigma - 1)) = -sin(pi sigma), cos likewise, so Y_(sigma-1) = (cos(pi sigma) J_(sigma-1) +
//     J_(1-sigma)) / sin(pi sigma);
//   * the four ascending series in ONE loop that shares the powers (-q)^k, each as a numerator / denominator pair
//     N_k = N_(k-1) d_k + (-q)^k, D_k = D_(k-1) d_k, d_k = k (k + nu), with one division at the end (a division per term
//     before), until the term is below 1e-17 of the partial sum for all four.
// The deterministic oracle evaluates the same expressions (oracle/rimo_heyvaerts.c jy_fast); the literal one keeps the
// four separate pow / Gamma / term-recurrence evaluations.  special-fun's source is absent, so neither is "the
// reference's arithmetic": both are pinned to scipy (tests/test_oracle_heyvaerts.py).
RIM_DEV void bessel_jy_fast(double sigma, double x, bool want_ym1, double *js, double *ys, double *jm1, double *ym1)
{
    RIM_HIT(28);
    const double h = 0.5 * x, q = h * h;
    double ll;
    const double lh = rim_log_dd(h, &ll);
    const double p = rim_pow_from_log(lh, ll, sigma);          // h^sigma
    const double hinv = 1. / h, pinv = 1. / p;
    double rg[4];
    rim_rgamma_quad(sigma, rg);                                 // 1 / Gamma of sigma + 1, sigma, 2 - sigma, 1 - sigma
    double sn, cs;
    rim_sincos(RIM_PI * sigma, &sn, &cs);
    // orders: 0 sigma, 1 sigma - 1, 2 -sigma, 3 1 - sigma
    const double nu0 = sigma, nu1 = sigma - 1., nu2 = -sigma, nu3 = 1. - sigma;
    double n0 = 1., n1 = 1., n2 = 1., n3 = 1., d0 = 1., d1 = 1., d2 = 1., d3 = 1., pw = 1.;
    const double mq = -q;
    for (int k = 1; k <= 80; k++) {
        RIM_HIT(30);
        const double kd = (double) k;
        pw = pw * mq;
        const double e0 = kd * (kd + nu0), e1 = kd * (kd + nu1), e2 = kd * (kd + nu2), e3 = kd * (kd + nu3);
        n0 = rim_fma(n0, e0, pw); d0 = d0 * e0;
        n1 = rim_fma(n1, e1, pw); d1 = d1 * e1;
        n2 = rim_fma(n2, e2, pw); d2 = d2 * e2;
        n3 = rim_fma(n3, e3, pw); d3 = d3 * e3;
        const double t = rim_fabs(pw);
        if (t < 1e-17 * rim_fabs(n0) && t < 1e-17 * rim_fabs(n1) && t < 1e-17 * rim_fabs(n2) && t < 1e-17 * rim_fabs(n3)) break;
    }
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

RIM_DEV void bessel_jy_generic(double sigma, double x, bool want_ym1, double *js, double *ys, double *jm1, double *ym1)
{
    double nu_y[2], sn[2], cs[2];
    for (int w = 0; w < 2; w++) {
        double nu = sigma - (double) w;
        if (nu == rim_floor(nu)) nu = nu + 1.4901161193847656e-08 * (rim_fabs(nu) > 1. ? rim_fabs(nu) : 1.);
        nu_y[w] = nu;
        rim_sincos(RIM_PI * nu, &sn[w], &cs[w]);
    }
    RIM_HIT(28);
    // jobs: 0 J(sigma), 1 J(sigma-1), 2 J(-nu_y0), 3 J(-nu_y1), 4 J(nu_y0) if perturbed, 5 J(nu_y1) if perturbed
    // (the six prefactors (x/2)^nu start from one double-double log of x/2)
    const double h = 0.5 * x, hsq = h * h;
    double ll;
    const double lh = rim_log_dd(h, &ll);
    double r[6];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (int k = 0; k < 6; k++) {
        const int w = k & 1;
        double nu;
        bool need;
        if (k < 2) { nu = sigma - (double) w; need = true; }
        else if (k < 4) { nu = -nu_y[w]; need = w == 0 || want_ym1; }
        else { nu = nu_y[w]; need = (w == 0 || want_ym1) && nu_y[w] != sigma - (double) w; }
        double v = 0.;
        if (need) { RIM_HIT(29); v = rim_pow_from_log(lh, ll, nu) * rgamma_real(nu + 1.) * ascending_series(nu, hsq, -1.); }   // = bessel_jnu(nu, x)
        r[k] = v;
    }
    *js = r[0];
    *jm1 = r[1];
    const double j0 = (nu_y[0] != sigma) ? r[4] : r[0];
    *ys = (cs[0] * j0 - r[2]) / sn[0];
    const double j1 = (nu_y[1] != sigma - 1.) ? r[5] : r[1];
    *ym1 = want_ym1 ? (cs[1] * j1 - r[3]) / sn[1] : 0.;
}

RIM_DEV void bessel_jy_set(double sigma, double x, bool want_ym1, double *js, double *ys, double *jm1, double *ym1)
{
    // (NaN takes the generic path and comes out as NaN there)
    if (sigma > 0. && sigma < 8. && sigma != rim_floor(sigma)) bessel_jy_fast(sigma, x, want_ym1, js, ys, jm1, ym1);
    else bessel_jy_generic(sigma, x, want_ym1, js, ys, jm1, ym1);
}

// Per-wave constants of the Faraday elements: the address of the series coefficient table.  On the device it is passed
// through an empty volatile asm once, so that the compiler keeps it in a scalar register pair across the quadrature loops
// instead of re-deriving the constant address (s_getpc + a GOT load and its wait) in front of every block of terms.
struct HeyConsts { const double *iser; };
RIM_DEV HeyConsts hey_consts()
{
    HeyConsts c;
    c.iser = RIM_ISERIES;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(c.iser));
#endif
    return c;
}

// The four I_nu(x) of a quasi-resonant sample (nu = 2/3, -2/3, 1/3, -1/3), I_nu = (x/2)^nu / Gamma(1 + nu) * series:
// the four prefactors from one cube root (rim_third_powers), the four 1 / Gamma(1 + nu) as constants.
RIM_DEV void bessel_i_g4(const HeyConsts &hc, double x, double out[4])
{
    RIM_HIT(27);
    const double h = 0.5 * x;
    const double hsq = h * h;
    double pref[4];
    rim_third_powers(h, pref);
    pref[0] *= RIM_RGAMMA_5_3; pref[1] *= RIM_RGAMMA_1_3; pref[2] *= RIM_RGAMMA_4_3; pref[3] *= RIM_RGAMMA_2_3;
    double ser[4];
    rim_iseries4(hc.iser, hsq, ser);
    for (int j = 0; j < 4; j++)
        out[j] = pref[j] * ser[j];
}

// Observer data of one Faraday coefficient (wave-uniform)
struct HeyPoint {
    double s, cos_th, sin_th, sigma0, sigma0_sq;
    double dinv;        // 1 / (sigma0 sin(theta)): gamma and the gamma-term of d f / d sigma MULTIPLY by it (round 4;
                        // heyvaerts.rs:197, 478 divide by sigma0 sin(theta) per sample -- the literal oracle still does)
    double endless_gamma;   // a Lorentz factor beyond which d f / d gamma is exactly 0 for good (hey_point_endless); +inf: none known
    int stokes;
};

// sigma0 and what follows from it, from s and the observer angle's sine
RIM_DEV void hey_point_derive(HeyPoint &pt)
{
    pt.sigma0 = pt.s * pt.sin_th;
    pt.sigma0_sq = pt.sigma0 * pt.sigma0;
    pt.dinv = 1. / (pt.sigma0 * pt.sin_th);
    pt.endless_gamma = RIM_INF;
}

// The quasi-resonant marching loop of the reference (heyvaerts.rs:156-185) only tests `keep_going` once qr_val is non-zero:
// while every chunk contributes exactly 0 it marches on for ever.  For the thermal distribution that is what happens to
// cold, high-frequency points (0.13 % of the rows of configs[2]'s table): d f / d gamma = norm exp(-gamma / T) (-1 / T) has
// underflowed to exactly 0 on the whole first chunk and gamma only grows from there.  This restatement's step cap turned
// "never returns" into NaN after 4096 such chunks (deviation 3) -- 4096 x 31 x 31 samples of exact
// zeros per coefficient, the longest chains of the thermal table.  Where the loop is PROVABLY endless the cap is now
// applied at once: endless_gamma = 760 T (rim_exp returns exactly 0 below -745.2; the margin covers every rounding of the
// sample's gamma), and hey_qr_is_endless() holds when every sample of every later chunk has gamma >= endless_gamma:
// gamma = (sigma - pomega cos(theta)) dinv >= (sigma - |pomega|) dinv >= (sigma - 3^(1/3) sigma^(2/3)) dinv, because
// |pomega| <= sqrt(3^(2/3) sigma^(4/3) - sigma0^2) (heyvaerts.rs:270-272), and that bound grows with sigma for sigma >= 1.
// Same value (NaN), same status (CHUNK_CAP); the deterministic AND the literal oracle apply the same test.
template <int KIND>
RIM_DEV void hey_point_endless(HeyPoint &pt, const DistParams &d)
{
    pt.endless_gamma = RIM_INF;
    if (KIND == DIST_THERMAL_JUETTNER) {
        const double g0 = -760. / d.neg_inverse_t;          // 760 T
        if (g0 > 0. && g0 < 1e300) pt.endless_gamma = g0;
    }
}
RIM_DEV bool hey_qr_is_endless(const HeyPoint &pt, double sigma_low)
{
    if (!(sigma_low >= 1.) || !(pt.endless_gamma < 1e300)) return false;
    const double c = rim_cbrt_normal(sigma_low);
    return (sigma_low - 1.4422495703074083 * (c * c)) * pt.dinv >= pt.endless_gamma;
}

struct HeyCoord { double sigma, pomega, x, gamma, mu; };

RIM_DEV HeyCoord fill_coord_vars(const HeyPoint &pt, double sigma, double pomega)
{
    HeyCoord c;
    c.sigma = sigma;
    c.pomega = pomega;
    c.x = rim_sqrt(sigma * sigma - pomega * pomega - pt.sigma0_sq);
    c.gamma = (sigma - pomega * pt.cos_th) * pt.dinv;
    c.mu = (sigma * pt.cos_th - pomega) / (pt.sigma0 * pt.sin_th * rim_sqrt(c.gamma * c.gamma - 1.));
    return c;
}

template <int KIND>
RIM_DEV double dfdsigma(const HeyPoint &pt, const DistParams &d, const HeyCoord &c)
{
    double dfdg, dfdcxi;
    calc_f_derivatives<KIND>(d, c.gamma, c.mu, dfdg, dfdcxi);
    const double g_term = dfdg * pt.dinv;
    double mu_term;
    if (dfdcxi == 0.) {
        mu_term = 0.;
    } else {
        const double q = c.sigma - c.pomega * pt.cos_th;
        const double r = c.pomega - c.sigma * pt.cos_th;
        const double t = pt.sigma0 * pt.sin_th;
        const double u = q * q - t * t;
        const double dcxi_dsigma = (q * u * pt.cos_th + u * r + r * (t * t)) / (rim_pow15(u) * q);
        mu_term = dcxi_dsigma * dfdcxi;
    }
    return g_term + mu_term;
}

// The inner integrand of the Faraday double integral in two pieces, so that rho_Q ("h") and rho_V ("f") of one
// parameter point can share what depends on the sample only -- the coordinates, d f / d sigma (the distribution's
// derivatives with their power and exponential), and the Bessel functions I_{+-1/3}, I_{+-2/3}(g) or J, Y of order sigma,
// sigma - 1 (heyvaerts.rs:302-468 evaluates them once per element) -- and differ in the element's closing formula.
struct HeyShared {
    HeyCoord c;
    double dfds;
    // quasi-resonant elements (heyvaerts.rs:302-373, 400-447)
    bool small_g;                   // g < G_APPROXIMATION_CUTOFF: the I_nu(g) forms
    double smxox;                   // (sigma - x) / x
    double g, iv0, iv1, iv2, iv3;   // I_{2/3}, I_{-2/3}, I_{1/3}, I_{-1/3} of g
    double js, ys, jm1, ym1;        // J_sigma, Y_sigma, J_{sigma-1}, Y_{sigma-1} of x (ym1 only for "h")
    // non-resonant elements (heyvaerts.rs:379-394, 453-468)
    double a1, a2, xa1p, x_sq, u, u2;
};

// The non-resonant elements divide by the powers 1/2, 1, 3/2, 2 and 5/2 of sigma^2 - x^2 eleven times between them.
// Here every one is a product of powers of u = 1 / sqrt(sigma^2 - x^2): one square root and one division per sample
// (the deterministic oracle does the same; the literal one divides).
RIM_DEV void nr_common(const HeyCoord &c, double &a1, double &a2, double &xa1p, double &x_sq, double &u, double &u2)
{
    const double s_sq = c.sigma * c.sigma;
    x_sq = c.x * c.x;
    u = 1. / rim_sqrt(s_sq - x_sq);
    u2 = u * u;
    const double ratio = s_sq * u2;
    a1 = 1. / 8. - 5. / 24. * ratio;
    a2 = 3. / 128. - 77. / 576. * ratio + 385. / 3456. * (ratio * ratio);
    xa1p = -5. / 12. * (s_sq * x_sq) * (u2 * u2);
}

// want_h: the "h" element will be asked for (it needs Y_{sigma-1} on the J/Y branch)
template <int KIND>
RIM_DEV HeyShared hey_shared(const HeyPoint &pt, const DistParams &d, const HeyConsts &hc, bool qr, double fixed, double v, bool want_h)
{
    HeyShared sh;
    sh.small_g = true;
    sh.smxox = 0.; sh.g = 0.; sh.iv0 = 0.; sh.iv1 = 0.; sh.iv2 = 0.; sh.iv3 = 0.;
    sh.js = 0.; sh.ys = 0.; sh.jm1 = 0.; sh.ym1 = 0.;
    sh.a1 = 0.; sh.a2 = 0.; sh.xa1p = 0.; sh.x_sq = 0.; sh.u = 0.; sh.u2 = 0.;
    if (qr) {
        RIM_HIT(26);
        sh.c = fill_coord_vars(pt, fixed, v);
        const HeyCoord &c = sh.c;
        // g = sqrt(8)/3 (sigma - x)^1.5 / sqrt(x) (heyvaerts.rs:309, 406) as sqrt(8)/3 (sigma - x) sqrt((sigma - x) / x): one
        // square root and the quotient the "h" element needs anyway, where the literal form takes two roots and a division
        sh.smxox = (c.sigma - c.x) / c.x;
        sh.g = RIM_SQRT_8_OVER_3 * (c.sigma - c.x) * rim_sqrt(sh.smxox);
        RIM_PROF_COUNT(22, 1);
#if defined(RIM_PROF) && defined(__HIP_DEVICE_COMPILE__)
        { const unsigned long long big = __ballot(!(sh.g < RIM_G_APPROXIMATION_CUTOFF)); RIM_PROF_COUNT(23, big != 0 ? 1 : 0);
          RIM_PROF_COUNT(24, __builtin_popcountll(big)); }
#endif
        sh.small_g = sh.g < RIM_G_APPROXIMATION_CUTOFF;
        if (sh.small_g) {
            double iv[4];
            bessel_i_g4(hc, sh.g, iv);
            sh.iv0 = iv[0]; sh.iv1 = iv[1]; sh.iv2 = iv[2]; sh.iv3 = iv[3];
        } else {
            bessel_jy_set(c.sigma, c.x, want_h, &sh.js, &sh.ys, &sh.jm1, &sh.ym1);
        }
    } else {
        RIM_HIT(25);
        sh.c = fill_coord_vars(pt, v, fixed);
        nr_common(sh.c, sh.a1, sh.a2, sh.xa1p, sh.x_sq, sh.u, sh.u2);
    }
    sh.dfds = dfdsigma<KIND>(pt, d, sh.c);
    return sh;
}

// the element of Stokes parameter `stokes` (Q: "h", heyvaerts.rs:302-394; V: "f", :400-468) from the shared part
RIM_DEV double hey_member(const HeyPoint &pt, int stokes, bool qr, const HeyShared &sh)
{
    const HeyCoord &c = sh.c;
    if (qr) {
        if (stokes == STOKES_Q) {
            const double po_sq = c.pomega * c.pomega;
            const double smxox = sh.smxox;
            double y1, y2;
            if (sh.small_g) {
                const double plus = sh.iv0, minus = sh.iv1, plus1 = sh.iv2, minus1 = sh.iv3;
                y1 = RIM_FOUR_OVER_SQRT_27 * (smxox * smxox) * (minus - plus) * (minus + plus);
                y2 = 0.5 * RIM_FOUR_OVER_SQRT_27 * smxox * (minus1 - plus1) * (minus1 + plus1);
            } else {
                const double jvp = sh.jm1 - c.sigma * sh.js / c.x;
                const double yvp = sh.ym1 - c.sigma * sh.ys / c.x;
                y1 = jvp * yvp;
                y2 = -sh.js * sh.ys;
            }
            const double t1 = RIM_PI * RIM_PI * (c.x * c.x) * y1;
            const double t2 = RIM_PI * RIM_PI * (c.pomega * c.pomega) * y2;
            const double t3 = -RIM_PI * (2. * po_sq + pt.sigma0_sq) / rim_sqrt(po_sq + pt.sigma0_sq);
            return RIM_INVERSE_C * (t1 + t2 + t3) * sh.dfds;
        }
        double y;
        if (sh.small_g) {
            y = RIM_INVERSE_SQRT_3
                * sh.g
                * (sh.iv1 - sh.iv0)
                * (sh.iv3 + sh.iv2);
        } else {
            const double jvp = sh.jm1 - c.sigma * sh.js / c.x;
            y = -c.x * jvp * sh.ys;
        }
        return -(2. * RIM_PI) * RIM_INVERSE_C * c.pomega * (RIM_PI * y - 1.) * sh.dfds;
    }
    const double a1 = sh.a1, a2 = sh.a2, xa1p = sh.xa1p, x_sq = sh.x_sq, u = sh.u, u2 = sh.u2;
    if (stokes == STOKES_Q) {
        const double u3 = u2 * u;
        const double t1 = (6. * a2 - a1 * a1 + xa1p) * u + a1 * x_sq * u3
            - (x_sq * x_sq) * (u3 * u2) * 0.125;
        const double t2 = (6. * a2 - a1 * a1) * u3;
        const double u1 = 2. * t1 - pt.sigma0_sq * t2;
        return RIM_PI * RIM_INVERSE_C * u1 * sh.dfds;
    }
    const double z =
        0.5 * x_sq * (u2 * u)
        + (6. * a2 + xa1p - a1 * a1) * u2
        + 1.5 * a1 * x_sq * (u2 * u2);
    return -2. * RIM_PI * RIM_INVERSE_C * z * c.pomega * sh.dfds;
}

// The inner integrand of either regime: `qr` selects quasi-resonant (integration variable
// pomega at fixed sigma) or non-resonant (integration variable sigma at fixed pomega).
template <int KIND>
RIM_DEV double hey_element(const HeyPoint &pt, const DistParams &d, const HeyConsts &hc, bool qr, double fixed, double v)
{
    RIM_PROF_T(t_el);
    const HeyShared sh = hey_shared<KIND>(pt, d, hc, qr, fixed, v, pt.stokes == STOKES_Q);
    const double r = hey_member(pt, pt.stokes, qr, sh);
    RIM_PROF_ADD(qr ? 19 : 18, t_el);
    return r;
}

}  // namespace rim
#endif
