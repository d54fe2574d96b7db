// wave_qag.h -- wavefront-cooperative Gauss-Kronrod 31 and adaptive QAG for gfx950.
//
// Replaces gsl_integration_qag(key = GK31) as the reference drives it through
// src/gsl.rs:156-207 (QUADPACK dqage/dqk31; algorithm restated in
// oracle/rimo_quad.c).  Design:
//
//   * One 64-lane wavefront owns one integral.  The 31 Kronrod nodes of an
//     interval sit on lanes 0..30 of a 32-lane half-wave (lane 31 is padding
//     with zero weight); each lane keeps its abscissa and its two weights in
//     registers (loaded once from the LDS copy of the table).
//   * A bisection evaluates BOTH children in one pass: left child on lanes
//     0..31, right child on lanes 32..63 (62 of 64 lanes busy).
//   * The four rule sums (Kronrod, Gauss, |f|, |f - mean|) are formed from QUADPACK's own terms: the two nodes
//     +-xgk[k] of the rule sit on ADJACENT lanes (2k, 2k + 1; lane 30 is the centre, lane 31 padding), so the first
//     butterfly step (ds_swizzle xor 1) is qk.c's pair sum f1 + f2, the product with the pair's weight follows, and four
//     more steps (xor 2, 4, 8, 16) add the 16 terms w (f1 + f2).  Floating-point addition commutes, so every lane of
//     a half ends with the same bits: the value of a fixed balanced binary tree over the 16 terms.  The CPU oracle
//     forms the same terms and adds them in the same tree, which is what makes results comparable bit for bit.
//     (Until round 3 the kernels summed 32 per-NODE products w f.  Far out in the harmonic tail the samples are
//     subnormal -- a few quanta of 2^-1074 -- and every product rounds to whole quanta: per-node products of small
//     samples vanish where the reference's pair terms survive, GSL's round-off detectors saw systematically different
//     error estimates, and 1 % of the thermal coefficients were NaN here and numbers in the reference's arithmetic.
//     With the reference's terms the sums of subnormal samples are exact in any order: DESIGN.md section 2.)
//   * The subinterval list (alist/blist/rlist/elist of the GSL workspace) lives
//     in LDS, wave-private.  GSL's sorted `order` list is replaced by a
//     wave-wide argmax over (error, insertion stamp): GSL inserts a new entry
//     in front of entries of equal error, so "largest error, latest insertion
//     wins ties" selects the same interval (exact while the list is shorter
//     than limit/2 + 2, beyond which GSL stops keeping it fully sorted; the LDS
//     capacity is far below that).  After the first bisection GSL always takes
//     slot 0 (qpsrt's `last < 2` shortcut); so do we.
//
// All bookkeeping values are wave-uniform; they are computed redundantly by
// every lane (there is no scalar fp64 unit) so no broadcast is needed.
#ifndef RIM_WAVE_QAG_H
#define RIM_WAVE_QAG_H

#if defined(RIM_WAVE_EMU)
#include "wave_emu.h"      // tests/support: 64 host threads + barrier collectives (CPU debugging only)
#else
#include <hip/hip_runtime.h>
#endif
#include "detmath.h"
#include "dev_bessel.h"
#include "gk31_table.h"

namespace rim {

#if !defined(RIM_WAVE_EMU)
// ---- wave primitives (gfx950) ---------------------------------------------------
__device__ __forceinline__ int wv_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ void wv_sync() { __syncthreads(); }     // workgroup == one wave
__device__ __forceinline__ int wv_readlane(int v, int srclane) { return __builtin_amdgcn_readlane(v, srclane); }
__device__ __forceinline__ int wv_readfirstlane(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double wv_shfl_xor(double v, int m) { return __shfl_xor(v, m); }
__device__ __forceinline__ int wv_shfl_xor(int v, int m) { return __shfl_xor(v, m); }
// The value of lane (lane ^ M), M < 32, by ds_swizzle in bit-mask mode (and 0x1f, or 0, xor M within each group of
// 32 lanes): the same exchange as wv_shfl_xor(v, M) without the per-step address arithmetic of ds_bpermute
// (lane ^ M, the bounds select, the shift: 4 VALU issues per distinct M and a live VGPR each).
template <int M>
__device__ __forceinline__ double wv_swz_xor(double v)
{
    static_assert(M > 0 && M < 32, "ds_swizzle bit-mask mode works within 32 lanes");
    const unsigned long long u = rim_bits(v);
    const unsigned lo = (unsigned) __builtin_amdgcn_ds_swizzle((int) (unsigned) u, (M << 10) | 0x1f);
    const unsigned hi = (unsigned) __builtin_amdgcn_ds_swizzle((int) (unsigned) (u >> 32), (M << 10) | 0x1f);
    return rim_frombits(((unsigned long long) hi << 32) | lo);
}
__device__ __forceinline__ unsigned long long wv_ballot(bool p) { return __ballot(p); }
#else
template <int M> inline double wv_swz_xor(double v) { return wv_shfl_xor(v, M); }     // emulator: same exchange
#endif

// A wave-uniform decision taken from lane 0's operands (the bookkeeping of the lean QAG below computes in "lane 0's
// view": every lane executes the arithmetic, lane 0 holds the operands that matter).  One v_cmp into an SGPR pair and
// scalar bit tests; the result is provably uniform, so the branch on it is a scalar branch.
__device__ __forceinline__ bool lane0(bool c) { return (wv_ballot(c) & 1ull) != 0; }
// lane 0's (a && b): two compares into scalar masks and scalar logic (a fused condition would be materialised as a
// 0/1 vector register and compared again)
__device__ __forceinline__ bool lane0_and(bool a, bool b) { return (wv_ballot(a) & wv_ballot(b) & 1ull) != 0; }

// max(a, b) of two non-NaN doubles in one instruction (__builtin_fmax adds a canonicalising v_max per operand)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RIM_WAVE_EMU)
__device__ __forceinline__ double rim_vmax(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
__device__ __forceinline__ double rim_vmax(double a, double b) { return __builtin_fmax(a, b); }
#endif

enum {
    QAG_SUCCESS = 0, QAG_EFAILED = 5, QAG_EMAXITER = 11, QAG_EBADTOL = 13, QAG_EROUND = 18, QAG_ESING = 21,
    QAG_ESTORE = 1001   // LDS subinterval store exhausted (not a GSL code)
};

__constant__ double c_gk_x[32] = RIM_GK31_X;
__constant__ double c_gk_wk[32] = RIM_GK31_WK;
__constant__ double c_gk_wg[32] = RIM_GK31_WG;

struct GKLane {
    const double *tab;  // LDS image of the rule IN LANE ORDER: [0..31] abscissae, [32..63] Kronrod weights, [64..95] Gauss weights
    int lane, half, j;
    bool node;          // j < 31
};
__device__ __forceinline__ double gk_t(const GKLane &g) { return g.tab[g.j]; }
__device__ __forceinline__ double gk_wk(const GKLane &g) { return g.tab[32 + g.j]; }
__device__ __forceinline__ double gk_wg(const GKLane &g) { return g.tab[64 + g.j]; }

// Node of the ascending table (gk31_table.h) that lane j of a half-wave evaluates: lanes 2k and 2k + 1 hold the
// symmetric pair -xgk[k], +xgk[k] (table entries k and 30 - k), lane 30 the centre.  Lane 31 is padding: it never
// evaluates the integrand (GKLane::node is false, its sample counts as 0), but it carries the CENTRE's weights, so
// that both lanes of the last pair hold the same term w_c (f_c + 0) and every lane of a half-wave ends with the same
// rule sums (with a zero weight there the odd lanes would miss the centre term).
__device__ __forceinline__ int gk_node_of_lane(int j)
{
    return j >= 30 ? 15 : ((j & 1) ? 30 - (j >> 1) : (j >> 1));
}

// LDS image of the rule + per-lane registers.  tab must hold 96 doubles.
__device__ __forceinline__ GKLane gk_lane_init(double *tab)
{
    GKLane g;
    g.lane = wv_lane();
    g.half = g.lane >> 5;
    g.j = g.lane & 31;
    g.node = g.j < 31;
    if (g.lane < 32) {
        const int node = gk_node_of_lane(g.lane);
        tab[g.lane] = c_gk_x[node];
        tab[32 + g.lane] = c_gk_wk[node];
        tab[64 + g.lane] = c_gk_wg[node];
    }
    wv_sync();
    g.tab = tab;
    return g;
}

__device__ __forceinline__ double readlane_d(double v, int srclane)
{
    const unsigned long long u = rim_bits(v);
    const unsigned lo = (unsigned) wv_readlane((int) (unsigned) u, srclane);
    const unsigned hi = (unsigned) wv_readlane((int) (unsigned) (u >> 32), srclane);
    return rim_frombits(((unsigned long long) hi << 32) | lo);
}

// Mark a wave-uniform value as such: the result lives in SGPRs, which keeps the
// long-lived per-point / per-order constants out of the vector register file.
__device__ __forceinline__ double uni(double v)
{
    const unsigned long long u = rim_bits(v);
    const unsigned lo = (unsigned) wv_readfirstlane((int) (unsigned) u);
    const unsigned hi = (unsigned) wv_readfirstlane((int) (unsigned) (u >> 32));
    return rim_frombits(((unsigned long long) hi << 32) | lo);
}
__device__ __forceinline__ int uni(int v) { return wv_readfirstlane(v); }
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned) wv_readfirstlane((int) v); }
__device__ __forceinline__ bool uni(bool v) { return wv_readfirstlane((int) v) != 0; }

// The rule sums: v + (value of the other node of the pair) is qk.c's f1 + f2 ...
__device__ __forceinline__ double pair_sum(double v) { return v + wv_swz_xor<1>(v); }
// ... and the 16 weighted pair terms (both lanes of a pair hold the same term) are added over the pairs of a half-wave.
__device__ __forceinline__ double sum16(double v)
{
    v = v + wv_swz_xor<2>(v);
    v = v + wv_swz_xor<4>(v);
    v = v + wv_swz_xor<8>(v);
    v = v + wv_swz_xor<16>(v);
    return v;
}

struct GKRes { double result, abserr, resabs, resasc; };

// qk.c: rescale_error
__device__ __forceinline__ double rescale_error(double err, double result_abs, double result_asc)
{
    err = rim_fabs(err);
    if (result_asc != 0 && err != 0) {
        const double scale = rim_pow15(200 * err / result_asc);
        if (scale < 1) err = result_asc * scale;
        else err = result_asc;
    }
    if (result_abs > RIM_DBL_MIN / (50 * RIM_DBL_EPSILON)) {
        const double min_err = 50 * RIM_DBL_EPSILON * result_abs;
        if (min_err > err) err = min_err;
    }
    return err;
}

// Apply the rule to the samples fv (one per lane; 0 on padding/inactive lanes).
// half_length is the lane's interval half-length (uniform within a half-wave).
__device__ __forceinline__ GKRes wave_gk31(double fv, double half_length, const GKLane &g)
{
    const double wk = gk_wk(g);
    const double fsum = pair_sum(fv);                                  // qk.c: fsum = fval1 + fval2 (centre: fc + 0)
    double rk = sum16(wk * fsum);
    const double rg = sum16(gk_wg(g) * fsum);
    double ra = sum16(wk * pair_sum(rim_fabs(fv)));
    const double mean = rk * 0.5;
    const double dev = g.node ? rim_fabs(fv - mean) : 0.;              // the padding lane holds 0, not |0 - mean|
    double rasc = sum16(wk * pair_sum(dev));
    const double ahl = rim_fabs(half_length);
    const double err = (rk - rg) * half_length;
    rk *= half_length;
    ra *= ahl;
    rasc *= ahl;
    GKRes r;
    r.result = rk;
    r.resabs = ra;
    r.resasc = rasc;
    r.abserr = rescale_error(err, ra, rasc);
    return r;
}

// ---- subinterval store ------------------------------------------------------

// The first `cap` subintervals live in LDS; anything beyond spills to a wave-private
// region of global memory (`g`, capacity `gcap`), so the GSL `limit` (up to 5000) is honoured
// without reserving 160 KB of LDS per wave.  On the bench tables 99.9 % of the integrals stay
// within 64 entries; the long lists belong to integrals that end in a GSL round-off error.
struct IStore {
    double *a, *b, *r, *e;   // LDS, [cap]
    int *stamp;
    int cap;
    double *g;               // global spill: [4 * gcap] doubles + [gcap] ints, or null
    int gcap;
};

__device__ __forceinline__ IStore istore_carve(double *base, int cap, double *gbase, int gcap)
{
    IStore s;
    s.a = base;
    s.b = base + cap;
    s.r = base + 2 * cap;
    s.e = base + 3 * cap;
    s.stamp = (int *) (base + 4 * cap);
    s.cap = cap;
    s.g = gbase;
    s.gcap = gcap;
    return s;
}
// doubles needed per store (LDS part, and per-wave global part)
#define RIM_ISTORE_DOUBLES(cap) (4 * (cap) + ((cap) + 1) / 2)

__device__ __forceinline__ int istore_capacity(const IStore &s) { return s.cap + s.gcap; }
__device__ __forceinline__ double ist_a(const IStore &s, int i) { return i < s.cap ? s.a[i] : s.g[i - s.cap]; }
__device__ __forceinline__ double ist_b(const IStore &s, int i) { return i < s.cap ? s.b[i] : s.g[s.gcap + (i - s.cap)]; }
__device__ __forceinline__ double ist_r(const IStore &s, int i) { return i < s.cap ? s.r[i] : s.g[2 * s.gcap + (i - s.cap)]; }
__device__ __forceinline__ double ist_e(const IStore &s, int i) { return i < s.cap ? s.e[i] : s.g[3 * s.gcap + (i - s.cap)]; }
__device__ __forceinline__ int ist_stamp(const IStore &s, int i)
{
    return i < s.cap ? s.stamp[i] : ((const int *) (s.g + 4 * s.gcap))[i - s.cap];
}
__device__ __forceinline__ void ist_set_a(const IStore &s, int i, double v) { if (i < s.cap) s.a[i] = v; else s.g[i - s.cap] = v; }
__device__ __forceinline__ void ist_set_b(const IStore &s, int i, double v) { if (i < s.cap) s.b[i] = v; else s.g[s.gcap + (i - s.cap)] = v; }
__device__ __forceinline__ void ist_set_r(const IStore &s, int i, double v) { if (i < s.cap) s.r[i] = v; else s.g[2 * s.gcap + (i - s.cap)] = v; }
__device__ __forceinline__ void ist_set_e(const IStore &s, int i, double v) { if (i < s.cap) s.e[i] = v; else s.g[3 * s.gcap + (i - s.cap)] = v; }
__device__ __forceinline__ void ist_set_stamp(const IStore &s, int i, int v)
{
    if (i < s.cap) s.stamp[i] = v; else ((int *) (s.g + 4 * s.gcap))[i - s.cap] = v;
}

// ---- QAG bookkeeping (qag.c), shared by the inner and the outer integrals ----

struct QagState {
    double epsabs, epsrel;
    int limit;
    double area, errsum, tolerance;
    int iteration, rt1, rt2, error_type, size, imax;
    double a1, b1, a2, b2, r_i, e_i;   // children of the interval being bisected
    int status;
    double result, abserr;
};

// After a reload from LDS the compiler can no longer see that the state is
// wave-uniform; passing every field through readfirstlane restores that
// knowledge (scalar branches, SGPR residency) -- and it is what keeps the
// cross-lane operations inside the QAG loops in provably uniform control flow.
__device__ __forceinline__ void qag_uniformize(QagState &q)
{
    q.epsabs = uni(q.epsabs); q.epsrel = uni(q.epsrel); q.limit = uni(q.limit);
    q.area = uni(q.area); q.errsum = uni(q.errsum); q.tolerance = uni(q.tolerance);
    q.iteration = uni(q.iteration); q.rt1 = uni(q.rt1); q.rt2 = uni(q.rt2);
    q.error_type = uni(q.error_type); q.size = uni(q.size); q.imax = uni(q.imax);
    q.a1 = uni(q.a1); q.b1 = uni(q.b1); q.a2 = uni(q.a2); q.b2 = uni(q.b2);
    q.r_i = uni(q.r_i); q.e_i = uni(q.e_i);
    q.status = uni(q.status); q.result = uni(q.result); q.abserr = uni(q.abserr);
}

// The same for the fields that are live between two passes of a running QAG; the tolerances and the
// limit are re-imposed from the caller's (constant) arguments, status/result/abserr are outputs.
__device__ __forceinline__ void qag_uniformize_live(QagState &q, double epsabs, double epsrel, int limit)
{
    q.epsabs = epsabs; q.epsrel = epsrel; q.limit = limit;
    q.area = uni(q.area); q.errsum = uni(q.errsum);
    q.tolerance = 0.;                          // recomputed from area before every use
    q.iteration = uni(q.iteration); q.rt1 = uni(q.rt1); q.rt2 = uni(q.rt2);
    q.error_type = 0;                          // a running QAG has error_type == 0 (it ends the loop otherwise)
    q.size = uni(q.size); q.imax = uni(q.imax);
    q.a1 = uni(q.a1); q.b1 = uni(q.b1); q.a2 = q.b1; q.b2 = uni(q.b2);      // the children share their midpoint
    q.r_i = uni(q.r_i); q.e_i = uni(q.e_i);
    q.status = QAG_SUCCESS; q.result = 0.; q.abserr = 0.;
}

__device__ __forceinline__ void qag_begin(QagState &q, double epsabs, double epsrel, int limit)
{
    q.epsabs = epsabs;
    q.epsrel = epsrel;
    q.limit = limit;
    q.iteration = 0;
    q.rt1 = q.rt2 = q.error_type = 0;
    q.size = 0;
    q.imax = 0;
    q.status = QAG_SUCCESS;
    q.result = 0.;
    q.abserr = 0.;
    q.area = q.errsum = q.tolerance = 0.;
}

// After the first rule application on [a, b].  Returns true when finished.
__device__ __forceinline__ bool qag_after_first(QagState &q, const IStore &st, int lane, double a, double b,
                                       double result0, double abserr0, double resabs0, double resasc0)
{
    if (q.epsabs <= 0 && (q.epsrel < 50 * RIM_DBL_EPSILON || q.epsrel < 0.5e-28)) {
        q.status = QAG_EBADTOL;
        q.result = 0.;
        return true;
    }
    if (lane == 0) {
        ist_set_a(st, 0, a); ist_set_b(st, 0, b); ist_set_r(st, 0, result0); ist_set_e(st, 0, abserr0);
        ist_set_stamp(st, 0, 0);
    }
    q.size = 1;
    q.tolerance = rim_max(q.epsabs, q.epsrel * rim_fabs(result0));
    const double round_off = 50 * RIM_DBL_EPSILON * resabs0;
    q.result = result0;
    q.abserr = abserr0;
    if (abserr0 <= round_off && abserr0 > q.tolerance) { q.status = QAG_EROUND; return true; }
    if ((abserr0 <= q.tolerance && abserr0 != resasc0) || abserr0 == 0.0) { q.status = QAG_SUCCESS; return true; }
    if (q.limit == 1) { q.status = QAG_EMAXITER; return true; }
    q.area = result0;
    q.errsum = abserr0;
    q.iteration = 1;
    q.imax = 0;
    return false;
}

// Choose the interval to bisect and set up its two children.
__device__ __forceinline__ void qag_pick(QagState &q, const IStore &st, int lane)
{
    wv_sync();   // make lane 0's store writes visible to the whole wave
    int imax = 0;
    if (q.size > 2) {
        RIM_HIT(2);
        // wave-wide argmax over (error, stamp)
        double be = -1.0;
        int bs = -1, bi = 0;
        for (int i = lane; i < q.size; i += 64) {
            const double e = ist_e(st, i);
            const int s = ist_stamp(st, i);
            if (e > be || (e == be && s > bs)) { be = e; bs = s; bi = i; }
        }
        // The largest error first (one max per butterfly step); the (error, stamp) tie-break of the
        // sorted list only matters when two lanes hold that same error, which is rare.
        double top = be;
        top = __builtin_fmax(top, wv_swz_xor<1>(top));
        top = __builtin_fmax(top, wv_swz_xor<2>(top));
        top = __builtin_fmax(top, wv_swz_xor<4>(top));
        top = __builtin_fmax(top, wv_swz_xor<8>(top));
        top = __builtin_fmax(top, wv_swz_xor<16>(top));
        top = __builtin_fmax(top, wv_shfl_xor(top, 32));
        const unsigned long long holders = wv_ballot(be == top);
        if (__builtin_popcountll(holders) == 1) {
            bi = wv_readlane(bi, __builtin_ffsll((long long) holders) - 1);
        } else {
            RIM_HIT(3);
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const double oe = wv_shfl_xor(be, m);
                const int os = wv_shfl_xor(bs, m);
                const int oi = wv_shfl_xor(bi, m);
                if (oe > be || (oe == be && os > bs)) { be = oe; bs = os; bi = oi; }
            }
        }
        imax = wv_readfirstlane(bi);
    }
    q.imax = imax;
    const double a_i = uni(ist_a(st, imax)), b_i = uni(ist_b(st, imax));
    q.r_i = uni(ist_r(st, imax));
    q.e_i = uni(ist_e(st, imax));
    q.a1 = a_i;
    q.b1 = 0.5 * (a_i + b_i);
    q.a2 = q.b1;
    q.b2 = b_i;
}

__device__ __forceinline__ bool subinterval_too_small(double a1, double a2, double b2)
{
    const double tmp = (1 + 100 * RIM_DBL_EPSILON) * (rim_fabs(a2) + 1000 * RIM_DBL_MIN);
    return rim_fabs(a1) <= tmp && rim_fabs(b2) <= tmp;
}

// After both children have been evaluated.  Returns true when the QAG is finished
// (q.status / q.result set).
__device__ __forceinline__ bool qag_after_bisect(QagState &q, const IStore &st, int lane,
                                        double area1, double error1, double resasc1,
                                        double area2, double error2, double resasc2)
{
    const double area12 = area1 + area2;
    const double error12 = error1 + error2;

    q.errsum += (error12 - q.e_i);
    q.area += area12 - q.r_i;

    if (resasc1 != error1 && resasc2 != error2) {
        const double delta = q.r_i - area12;
        if (rim_fabs(delta) <= 1.0e-5 * rim_fabs(area12) && error12 >= 0.99 * q.e_i) q.rt1++;
        if (q.iteration >= 10 && error12 > q.e_i) q.rt2++;
    }

    q.tolerance = rim_max(q.epsabs, q.epsrel * rim_fabs(q.area));

    if (q.errsum > q.tolerance) {
        if (q.rt1 >= 6 || q.rt2 >= 20) q.error_type = 2;
        if (subinterval_too_small(q.a1, q.a2, q.b2)) q.error_type = 3;
    }

    // update(): the child with the larger error keeps the parent's slot
    bool overflow = false;
    if (q.size >= istore_capacity(st)) {
        overflow = true;
    } else {
        wv_sync();   // all lanes have finished reading the parent's slot
        if (lane == 0) {
            const int inew = q.size;
            const int im = q.imax;
            if (error2 > error1) {
                ist_set_a(st, im, q.a2); ist_set_r(st, im, area2); ist_set_e(st, im, error2);
                ist_set_a(st, inew, q.a1); ist_set_b(st, inew, q.b1); ist_set_r(st, inew, area1); ist_set_e(st, inew, error1);
            } else {
                ist_set_b(st, im, q.b1); ist_set_r(st, im, area1); ist_set_e(st, im, error1);
                ist_set_a(st, inew, q.a2); ist_set_b(st, inew, q.b2); ist_set_r(st, inew, area2); ist_set_e(st, inew, error2);
            }
            ist_set_stamp(st, im, 2 * q.iteration);
            ist_set_stamp(st, inew, 2 * q.iteration + 1);
        }
        q.size++;
    }
    q.iteration++;

    if (!overflow && q.iteration < q.limit && !q.error_type && q.errsum > q.tolerance)
        return false;   // keep bisecting

    wv_sync();
    double sum = 0;
    for (int k = 0; k < q.size; k++) sum += ist_r(st, k);
    q.result = uni(sum);
    q.abserr = q.errsum;

    if (q.errsum <= q.tolerance) q.status = QAG_SUCCESS;
    else if (overflow) q.status = QAG_ESTORE;
    else if (q.error_type == 2) q.status = QAG_EROUND;
    else if (q.error_type == 3) q.status = QAG_ESING;
    else if (q.iteration == q.limit) q.status = QAG_EMAXITER;
    else q.status = QAG_EFAILED;
    return true;
}

// Work counters, wave-private in LDS (lane 0 updates them).
struct WaveCounters {
    unsigned long long samples;   // integrand samples (active lanes)
    unsigned long long steps;     // wave-wide integrand passes (each costs one full-wave evaluation)
    unsigned long long inner_qags;
};

// LDS scratch of one wave's QAG: the bookkeeping state is parked here while the
// integrand runs, so that the integrand has the vector register file to itself.
// State of the lean (LDS-resident list) QAG that does not fit uniform integer registers: the running sums, and the
// first-rule result of integral 1 while integral 0 runs.
struct LeanPark {
    double area, errsum;
    double fb[4];             // result, abserr, resabs, resasc of integral 1's first rule application
};

struct QagParkBase {
    LeanPark lean;
    QagState q;               // wave_qag (the single-integral form) parks its state here
    WaveCounters ctr;
    unsigned long long *hb;   // optional heartbeat words in host-mapped memory (diagnostics), else null
};
// ... plus, for the Symphony integrand, the order records (n, n + 1) of the one or two gamma-integrals in flight
// (SymOrder); the Heyvaerts kernel spends that LDS on the head of its series divisor table instead
struct QagPark : QagParkBase {
    LeungOrder ord[4];
};

// Diagnostics: make a word visible to the host while the kernel is still running.
__device__ __forceinline__ void hb_store(unsigned long long *p, unsigned long long v)
{
#if defined(RIM_WAVE_EMU)
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

// Adaptive QAG of a per-lane integrand f(x): one wave, one integral.
// f is called once per pass with the lane's abscissa and an `active` flag and
// must return 0 for inactive lanes.
template <class F>
__device__ __forceinline__ void wave_qag(F &f, const GKLane &g, const IStore &st, double a, double b,
                                double epsabs, double epsrel, int limit, QagState &q, QagParkBase *park)
{
    qag_begin(q, epsabs, epsrel, limit);
    bool first = true;
    for (;;) {
        double la, lb;
        bool active;
        if (first) {
            la = a; lb = b;
            active = g.node && g.half == 0;
        } else {
            RIM_PROF_T(t_pick);
            qag_pick(q, st, g.lane);
            RIM_PROF_ADD(13, t_pick);
            la = g.half ? q.a2 : q.a1;
            lb = g.half ? q.b2 : q.b1;
            active = g.node;
        }
        double fv;
        {
            const double center = 0.5 * (la + lb);
            const double hl = 0.5 * (lb - la);
            const double x = center + hl * gk_t(g);
            if (g.lane == 0) {
                park->q = q;
                park->ctr.samples += first ? 31 : 62;
                park->ctr.steps += 1;
                if (park->hb) {
                    hb_store(park->hb + 3, park->ctr.steps);
                    hb_store(park->hb + 4, (unsigned long long) q.iteration);
                }
            }
            RIM_PROF_T(t_int);
            fv = f(x, active);
            RIM_PROF_ADD(1, t_int);
            RIM_PROF_T(t_unpark);
            wv_sync();
            q = park->q;
            qag_uniformize(q);
            RIM_PROF_ADD(14, t_unpark);
        }
        if (first) {
            const double hl = 0.5 * (b - a);
            RIM_PROF_T(t_gk);
            const GKRes r = wave_gk31(fv, hl, g);
            RIM_PROF_ADD(11, t_gk);
            first = false;
            if (qag_after_first(q, st, g.lane, a, b, readlane_d(r.result, 0), readlane_d(r.abserr, 0),
                                readlane_d(r.resabs, 0), readlane_d(r.resasc, 0)))
                return;
        } else {
            const double la2 = g.half ? q.a2 : q.a1;
            const double lb2 = g.half ? q.b2 : q.b1;
            const double hl = 0.5 * (lb2 - la2);
            RIM_PROF_T(t_gk);
            const GKRes r = wave_gk31(fv, hl, g);
            RIM_PROF_ADD(11, t_gk);
            RIM_PROF_T(t_ab);
            const bool fin = qag_after_bisect(q, st, g.lane,
                                 readlane_d(r.result, 0), readlane_d(r.abserr, 0), readlane_d(r.resasc, 0),
                                 readlane_d(r.result, 32), readlane_d(r.abserr, 32), readlane_d(r.resasc, 32));
            RIM_PROF_ADD(12, t_ab);
            if (fin) return;
        }
    }
}

// ---- lean QAG: the subinterval list in LDS, the bookkeeping in vector registers ----------------------------
//
// The general functions above keep the QAG state wave-uniform in scalar registers.  gfx950 has no scalar fp64 ALU, so
// every uniform fp64 operation is a full vector instruction anyway, and feeding it from scalar registers costs extra
// moves (a VALU instruction takes one scalar operand), readlanes to make the rule sums uniform, moves again to store
// them, and the whole state is copied to LDS and back around every integrand call: ~380 of the ~1600 vector
// instructions of a pass of the Symphony kernel (profiles/r1_final_pmc_symphony_65536pts.json), which is issue-bound.
// The lean form below does the same arithmetic in the same order (every value is the one the functions above
// compute; tests/test_gpu_parity.py::test_qag_selftest_bit_exact and the end-to-end parity tests hold it to that)
// with the operands where they already are:
//   * the rule sums of child 1 sit in the lanes of half-wave 0, those of child 2 in half-wave 1; the bookkeeping is
//     computed "in lane 0's view": child 1's values are lane 0's own registers, child 2's arrive by six readlanes
//     (scalar operands, one per instruction), every lane executes the same instructions and only lane 0's results are
//     used: decisions through lane0() (one compare into a scalar mask), stores by lane 0;
//   * lanes 0 and 32 each file their own child in the list -- one ds_write per field with two active lanes;
//   * the list lives in LDS as one interval per lane (lane i reads entry i), so the interval of largest error is a
//     5-step max butterfly per half-wave, a scalar comparison of the two half-wave maxima (bit patterns of non-negative
//     doubles order like integers) and a ballot; ties -- which GSL resolves through its sorted list -- take the general
//     (error, stamp) reduction, as do lists that outgrow the LDS part (st.cap entries): the lean loop hands such an
//     integral to the general functions, which continue it in the spill store;
//   * the running sums area / errsum are two doubles in LDS (written by lane 0 from the registers they were computed
//     in), the counters are scalar integers that stay in scalar registers across the integrand.

// Index of the interval with the largest error among entries 0..size-1 (2 < size <= min(64, st.cap): one entry per
// lane, all in LDS).
__device__ __forceinline__ int lean_pick(const IStore &st, int size, int lane)
{
    const double e_l = lane < size ? st.e[lane] : -1.0;
    double top = e_l;
    top = rim_vmax(top, wv_swz_xor<1>(top));
    top = rim_vmax(top, wv_swz_xor<2>(top));
    top = rim_vmax(top, wv_swz_xor<4>(top));
    top = rim_vmax(top, wv_swz_xor<8>(top));
    top = rim_vmax(top, wv_swz_xor<16>(top));
    // errors are >= +0 (or the -1 padding): their bit patterns order like signed integers
    const long long t0 = (long long) rim_bits(readlane_d(top, 0));
    const long long t1 = (long long) rim_bits(readlane_d(top, 32));
    unsigned long long holders = wv_ballot(e_l == top);
    holders &= (t1 > t0) ? 0xffffffff00000000ull : 0x00000000ffffffffull;
    if (t0 != t1 && __builtin_popcountll(holders) == 1) return __builtin_ffsll((long long) holders) - 1;
    // tie (or a NaN error): largest error, latest insertion -- the order of GSL's sorted list
    double be = e_l;
    int bs = lane < size ? st.stamp[lane] : -1, bi = lane;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const double oe = wv_shfl_xor(be, m);
        const int os = wv_shfl_xor(bs, m);
        const int oi = wv_shfl_xor(bi, m);
        if (oe > be || (oe == be && os > bs)) { be = oe; bs = os; bi = oi; }
    }
    return wv_readfirstlane(bi);
}

// The same for a list of any length (entries beyond st.cap live in the wave's global spill region): each lane scans
// entries lane, lane + 64, ... and the wave reduces over (error, stamp).
__device__ __forceinline__ int spill_pick(const IStore &st, int size, int lane)
{
    double be = -1.0;
    int bs = -1, bi = 0;
    for (int i = lane; i < size; i += 64) {
        const double e = ist_e(st, i);
        const int sp = ist_stamp(st, i);
        if (e > be || (e == be && sp > bs)) { be = e; bs = sp; bi = i; }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const double oe = wv_shfl_xor(be, m);
        const int os = wv_shfl_xor(bs, m);
        const int oi = wv_shfl_xor(bi, m);
        if (oe > be || (oe == be && os > bs)) { be = oe; bs = os; bi = oi; }
    }
    return wv_readfirstlane(bi);
}

// One entry of the list at a wave-uniform index: LDS (typed accesses) or, behind a scalar branch, the spill region.
struct IEntry { double a, b, r, e; };
__device__ __forceinline__ IEntry ist_entry(const IStore &st, int i)
{
    IEntry en;
    if (i < st.cap) { en.a = st.a[i]; en.b = st.b[i]; en.r = st.r[i]; en.e = st.e[i]; }
    else {
        const int j = i - st.cap;
        en.a = st.g[j]; en.b = st.g[st.gcap + j]; en.r = st.g[2 * st.gcap + j]; en.e = st.g[3 * st.gcap + j];
    }
    return en;
}

// Two adaptive QAGs that share their first rule application.
//
// The first GK31 application of an integral uses 31 lanes, so two integrals are started together:
// lanes 0..30 apply the rule to integral 0 on [a0, b0], lanes 32..62 to integral 1 on [a1, b1]; then
// integral 0's bisections run to the end on the whole wave, then integral 1 continues from its parked
// first-rule result.  Each integral sees exactly the arithmetic of wave_qag (a half-wave's rule sums
// never involve the other half).  f(x, active, second) evaluates the integrand of integral 0 or 1
// (`second` is per lane during the joint application and uniform afterwards); `flip` swaps the roles
// (used to run integral 1 alone).  Results: r0/qs0 (and r1/qs1 if have1) = result and GSL status.
template <class F>
__device__ __forceinline__ void wave_qag_pair(F &f, const GKLane &g, const IStore &st,
                                              double a0, double b0, double a1, double b1, bool have1, bool flip,
                                              double epsabs, double epsrel, int limit, QagParkBase *park,
                                              double &r0, int &qs0, double &r1, int &qs1,
                                              double *abserr_out = nullptr, int *size_out = nullptr)   // [2] each, optional
{
    const int lane = g.lane;
    LeanPark *const lp = &park->lean;
    enum { M_FIRST = 0, M_LEAN = 1 };
    int mode = M_FIRST;              // M_FIRST: the joint first rule application; then bisections of integral `cur`
    int cur = 0;
    // the counters of the lean QAG: wave-uniform integers, in scalar registers across the integrand
    int iteration = 0, size = 0, rt1 = 0, rt2 = 0, imax = 0;
    unsigned n_pass = 0, n_samp = 0;
    r0 = 0.; r1 = 0.; qs0 = QAG_SUCCESS; qs1 = QAG_SUCCESS;

    for (;;) {
        // ---------------- before the integrand: this pass's interval(s) ----------------
        double la, lb;
        bool active, second;
        if (mode == M_FIRST) {
            second = g.half != 0 && have1;
            la = second ? a1 : a0;
            lb = second ? b1 : b0;
            active = g.node && (g.half == 0 || have1);
            n_samp += have1 ? 62u : 31u;
        } else {
            RIM_PROF_T(t_pick);
            wv_sync();               // the list entries written after the previous pass are visible to every lane
            imax = 0;                // qpsrt: slot 0 while the list has <= 2 entries
            if (size > 2) imax = (size <= 64 && size <= st.cap) ? lean_pick(st, size, lane) : spill_pick(st, size, lane);
            RIM_PROF_ADD(13, t_pick);
            double a_i, b_i;
            if (imax < st.cap) { a_i = st.a[imax]; b_i = st.b[imax]; }
            else { a_i = st.g[imax - st.cap]; b_i = st.g[st.gcap + (imax - st.cap)]; }
            const double mid = 0.5 * (a_i + b_i);
            la = g.half ? mid : a_i;
            lb = g.half ? b_i : mid;
            second = cur != 0;
            active = g.node;
            n_samp += 62u;
        }
        n_pass += 1u;
        const double hl = 0.5 * (lb - la);
        const double x = 0.5 * (la + lb) + hl * gk_t(g);
        RIM_HIT(0); if (mode == M_FIRST) RIM_HIT(1);
        RIM_PROF_T(t_int);
        const double fv = f(x, active, second != flip);
        RIM_PROF_ADD(1, t_int);

        // ---------------- after the integrand: apply the rule, book the result ----------------
        RIM_PROF_T(t_gk);
        const GKRes r = wave_gk31(fv, hl, g);
        RIM_PROF_ADD(11, t_gk);
        bool finished;
        int status = QAG_SUCCESS;
        double result = 0., abserr = 0.;
        double f_res, f_err, f_abs, f_asc;       // first-rule sums handed to the code after the first application
        bool first_pending = false;
        if (mode == M_FIRST) {
            if (lane == 32) { lp->fb[0] = r.result; lp->fb[1] = r.abserr; lp->fb[2] = r.resabs; lp->fb[3] = r.resasc; }
            f_res = r.result; f_err = r.abserr; f_abs = r.resabs; f_asc = r.resasc;     // lane 0 holds integral 0's
            first_pending = true;
            finished = false;
        } else {
            RIM_PROF_T(t_ab);
            // child 1 = this lane's own sums (right in lane 0), child 2 = lane 32's
            const double area2 = readlane_d(r.result, 32), error2 = readlane_d(r.abserr, 32);
            const double area1 = r.result, error1 = r.abserr;
            const IEntry en = ist_entry(st, imax);
            const double a_i = en.a, b_i = en.b, r_i = en.r, e_i = en.e;
            const double area12 = area1 + area2;
            const double error12 = error1 + error2;
            const double errsum = lp->errsum + (error12 - e_i);
            const double area = lp->area + (area12 - r_i);
            {
                // resasc != error for both children: each half-wave compares its own pair
                const unsigned long long ne = wv_ballot(r.resasc != r.abserr);
                if ((ne & 1ull) && ((ne >> 32) & 1ull)) {
                    const double delta = r_i - area12;
                    if (lane0_and(rim_fabs(delta) <= 1.0e-5 * rim_fabs(area12), error12 >= 0.99 * e_i)) rt1++;
                    if (iteration >= 10 && lane0(error12 > e_i)) rt2++;
                }
            }
            const double tolerance = (epsabs == 0.) ? epsrel * rim_fabs(area)          // = rim_max(0, x) for x >= 0 or NaN
                                                    : rim_max(epsabs, epsrel * rim_fabs(area));
            const bool more = lane0(errsum > tolerance);
            int error_type = 0;
            if (more) {
                if (rt1 >= 6 || rt2 >= 20) error_type = 2;
                const double mid = 0.5 * (a_i + b_i);
                const double tmp = (1 + 100 * RIM_DBL_EPSILON) * (rim_fabs(mid) + 1000 * RIM_DBL_MIN);
                if (lane0_and(rim_fabs(a_i) <= tmp, rim_fabs(b_i) <= tmp)) error_type = 3;
            }
            // update(): the child with the larger error keeps the parent's slot; lanes 0 and 32 file their own child
            const bool c2gt = lane0(error2 > error1);
            const bool overflow = size >= istore_capacity(st);
            wv_sync();               // every lane has read the parent's entry
            if (!overflow) {
                if ((lane & 31) == 0) {
                    const bool keep = (g.half != 0) == c2gt;
                    const int slot = keep ? imax : size;
                    const int stamp = 2 * iteration + (keep ? 0 : 1);
                    if (slot < st.cap) {
                        st.a[slot] = la; st.b[slot] = lb; st.r[slot] = r.result; st.e[slot] = r.abserr;
                        st.stamp[slot] = stamp;
                    } else {         // beyond the LDS part: the wave's spill region in global memory
                        const int j = slot - st.cap;
                        st.g[j] = la; st.g[st.gcap + j] = lb; st.g[2 * st.gcap + j] = r.result; st.g[3 * st.gcap + j] = r.abserr;
                        ((int *) (st.g + 4 * st.gcap))[j] = stamp;
                    }
                }
                size++;
            }
            iteration++;
            if (lane == 0) { lp->area = area; lp->errsum = errsum; }
            finished = !(!overflow && iteration < limit && !error_type && more);
            if (finished) {
                wv_sync();
                double sum = 0;
                if (size <= st.cap) { for (int k = 0; k < size; k++) sum += st.r[k]; }
                else { for (int k = 0; k < size; k++) sum += ist_r(st, k); }
                result = sum;
                abserr = errsum;
                if (lane0(errsum <= tolerance)) status = QAG_SUCCESS;      // (not "!more": a NaN error sum is a failure)
                else if (overflow) status = QAG_ESTORE;
                else if (error_type == 2) status = QAG_EROUND;
                else if (error_type == 3) status = QAG_ESING;
                else if (iteration == limit) status = QAG_EMAXITER;
                else status = QAG_EFAILED;
            }
            RIM_PROF_ADD(12, t_ab);
        }

        for (;;) {
            if (first_pending) {
                // qag_after_first on the first-rule sums in f_* (decisions from lane 0's values)
                first_pending = false;
                const double fa = cur ? a1 : a0, fb = cur ? b1 : b0;
                if (epsabs <= 0 && (epsrel < 50 * RIM_DBL_EPSILON || epsrel < 0.5e-28)) {
                    status = QAG_EBADTOL; result = 0.; abserr = 0.; size = 0;
                    finished = true;
                } else {
                    wv_sync();
                    if (lane == 0) {
                        st.a[0] = fa; st.b[0] = fb; st.r[0] = f_res; st.e[0] = f_err; st.stamp[0] = 0;
                        lp->area = f_res; lp->errsum = f_err;
                    }
                    size = 1;
                    const double tolerance = rim_max(epsabs, epsrel * rim_fabs(f_res));
                    const double round_off = 50 * RIM_DBL_EPSILON * f_abs;
                    result = f_res;
                    abserr = f_err;
                    finished = true;
                    if (lane0_and(f_err <= round_off, f_err > tolerance)) status = QAG_EROUND;
                    else if (lane0_and(f_err <= tolerance, f_err != f_asc) || lane0(f_err == 0.0)) status = QAG_SUCCESS;
                    else if (limit == 1) status = QAG_EMAXITER;
                    else {
                        finished = false;
                        iteration = 1; rt1 = 0; rt2 = 0; imax = 0;
                        mode = M_LEAN;
                    }
                }
            }
            if (!finished) break;
            // integral `cur` is complete
            RIM_HIT(22);
            if (cur == 0) { r0 = readlane_d(result, 0); qs0 = status; } else { r1 = readlane_d(result, 0); qs1 = status; }
            if (abserr_out) abserr_out[cur] = readlane_d(abserr, 0);
            if (size_out) size_out[cur] = size;
            if (cur != 0 || !have1) {
                if (lane == 0) {
                    park->ctr.samples += n_samp;
                    park->ctr.steps += n_pass;
                    park->ctr.inner_qags += have1 ? 2 : 1;
                    if (park->hb) {
                        hb_store(park->hb + 3, park->ctr.steps);
                        hb_store(park->hb + 4, (unsigned long long) iteration);
                    }
                }
                return;
            }
            cur = 1;                 // integral 1 continues from its parked first-rule result
            wv_sync();
            f_res = lp->fb[0]; f_err = lp->fb[1]; f_abs = lp->fb[2]; f_asc = lp->fb[3];
            first_pending = true;
            finished = false;
        }
    }
}

}  // namespace rim
#endif
