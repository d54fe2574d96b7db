// symphony_wave.h -- one wavefront computes one Symphony coefficient.
//
//   CalculationState::compute   symphony.rs:66-187   (30 discrete harmonics + n tail, V in two lobes)
//   n_integration               symphony.rs:196-295  (deriv_central probe, chunk growth, QAG over n)
//   gamma_integral              symphony.rs:312-389  (limits, rel_width, QAG over gamma)
//   gsl::deriv_central          gsl.rs:233-257 -> gsl_deriv_central (restated in oracle/rimo_quad.c)
//
// Control structure.  The reference nests callbacks: QAG(n) -> [per node]
// QAG(gamma) -> [per node] integrand.  On the GPU the innermost level is the
// only place with lane parallelism (31 or 62 samples per pass), so the wave
// runs the gamma-integrals one after another and everything above them is
// wave-uniform bookkeeping.  To keep a SINGLE inlined copy of the expensive
// code (integrand + inner QAG) in the kernel, the three callers of
// gamma_integral -- the discrete harmonics, the finite-difference probe and the
// outer GK31 rule -- are written as phases of one state machine: each phase
// posts a batch of up to 62 (n, lobe) requests, one per lane, the common loop
// evaluates them, and the phase's continuation consumes the 62 results.
//
// Every arithmetic step is in the same order as oracle/rimo_symphony.c.
#ifndef RIM_SYMPHONY_WAVE_H
#define RIM_SYMPHONY_WAVE_H

#include "wave_qag.h"
#include "dev_symphony.h"

namespace rim {

#define RIM_TWO_PI (2. * RIM_PI)
#define RIM_MASS_ELECTRON 9.1093826e-28
#define RIM_SPEED_LIGHT 2.99792458e10
#define RIM_ELECTRON_CHARGE 4.80320680e-10
#define RIM_MAX_CHUNKS 4096
#define RIM_MAX_BATCHES (1 << 22)   // backstop: no coefficient needs anywhere near this many batches

// status bits reported per coefficient
enum {
    ST_OK = 0,
    ST_INNER_FAIL = 1,     // some gamma-integral returned an error (-> NaN sample upstream)
    ST_OUTER_FAIL = 2,     // an n-chunk QAG failed
    ST_CHUNK_CAP = 4,      // RIM_MAX_CHUNKS exceeded
    ST_STORE_FULL = 8,     // LDS subinterval store exhausted
    ST_NONFINITE = 16,     // result is NaN
    ST_NORM_FAIL = 32      // distribution normalisation failed
};

// Rust `x as i64`: truncate toward zero, saturating, NaN -> 0
__device__ __forceinline__ long long sat_i64(double x)
{
    if (!(x == x)) return 0;
    if (x >= 9223372036854775807.0) return 9223372036854775807LL;
    if (x <= -9223372036854775808.0) return (-9223372036854775807LL - 1);
    return (long long) x;
}

// gamma_integral (symphony.rs:312-389): limits for order n and lobe.
struct GammaLimits { double g0, g1; };

__device__ __forceinline__ GammaLimits gamma_limits(const SymPoint &pt, double n, int negative_lobe)
{
    const double s = pt.s;
    const double acos_th = rim_fabs(pt.cos_th);
    const double sin2 = pt.sin_th * pt.sin_th;
    const double nos = n / s;
    const double root = rim_sqrt(nos * nos - sin2);
    const double gamma_minus = (nos - acos_th * root) / sin2;
    const double gamma_plus = (nos + acos_th * root) / sin2;
    const double gamma_peak = 0.5 * (gamma_plus + gamma_minus);
    const double rel_width = (s < 1e6) ? 1. : rim_exp(-0.27 * rim_log(n) - 0.1);
    const double gamma_minus_high = gamma_peak - (gamma_peak - gamma_minus) * rel_width;
    const double gamma_plus_high = gamma_peak - (gamma_peak - gamma_plus) * rel_width;
    GammaLimits L;
    if (pt.stokes == STOKES_V) {
        if (!negative_lobe) { L.g0 = gamma_peak; L.g1 = gamma_plus_high; }
        else { L.g0 = gamma_minus_high; L.g1 = gamma_peak; }
    } else {
        L.g0 = gamma_minus_high;
        L.g1 = gamma_plus_high;
    }
    return L;
}

enum Phase { PH_DISCRETE = 0, PH_DERIV1, PH_DERIV2, PH_QAG_FIRST, PH_QAG_BISECT, PH_DONE };

// Wave-uniform state of one coefficient.  It is only touched between batches of
// gamma-integrals, so it is parked in LDS while a batch runs: the vector
// registers stay free for the integrand.
struct TaskState {
    double ans;                                   // compute(): running sum
    double ni_ans, contrib, delta_n, incr, n_start;   // n_integration locals
    double dr_h, dr_r0, dr_err;                   // deriv_central locals
    double qa, qb;                                // chunk limits
    long long disc_next, disc_hi;                 // next discrete harmonic to post, end of the discrete range
    int chunks, lobe, phase, ni_failed, status, batches;
    QagState oq;                                  // outer QAG over n
};

__device__ __forceinline__ void task_uniformize(TaskState &T)
{
    T.ans = uni(T.ans); T.ni_ans = uni(T.ni_ans); T.contrib = uni(T.contrib); T.delta_n = uni(T.delta_n);
    T.incr = uni(T.incr); T.n_start = uni(T.n_start);
    T.dr_h = uni(T.dr_h); T.dr_r0 = uni(T.dr_r0); T.dr_err = uni(T.dr_err);
    T.qa = uni(T.qa); T.qb = uni(T.qb);
    T.disc_next = (long long) rim_bits(uni(rim_frombits((unsigned long long) T.disc_next)));
    T.disc_hi = (long long) rim_bits(uni(rim_frombits((unsigned long long) T.disc_hi)));
    T.chunks = uni(T.chunks); T.lobe = uni(T.lobe); T.phase = uni(T.phase); T.ni_failed = uni(T.ni_failed);
    T.status = uni(T.status); T.batches = uni(T.batches);
    qag_uniformize(T.oq);
}

// ---- the coefficient as a resumable computation -------------------------------------------
//
//   sym_begin    initialise the task state
//   sym_post     phase -> a batch of up to 62 (n, lobe) requests, one per lane
//   (the caller evaluates the requests: sym_eval_request, the only place the integrand lives)
//   sym_consume  continuation of the posting phase on the 62 results; T.phase == PH_DONE at the end
//   sym_result   final NaN policy + dimensional prefactor
//
// Splitting it this way lets the kernel hand the requests of a batch to OTHER waves once the task
// queue has run dry (cooperative tail, rimphony_hip.hip) while keeping one copy of the integrand.

// the most intervals whose children one batch of a long outer quadrature evaluates (heyvaerts_wave.h, "rounds")
#define RIM_TURBO_MAX 4

struct SymBatch {
    double req_n;        // per lane
    int req_lobe;        // per lane
    bool req_active;     // per lane
    int n_req;           // uniform: requests of the discrete / deriv phases (lanes 0 .. n_req-1)
    int phase;           // uniform: the phase that posted the batch
};

#define RIM_N_MAX 30.

// n_integration prologue (symphony.rs:197-215)
__device__ __forceinline__ void sym_start_lobe(const SymPoint &pt, TaskState &T, int which)
{
    const double n_minus = pt.s * rim_fabs(pt.sin_th);
    T.lobe = which;
    T.ni_ans = 0.; T.contrib = 0.; T.delta_n = 1e5; T.incr = 10.; T.chunks = 0;
    T.ni_failed = 0;
    T.n_start = rim_floor(n_minus + 1. + RIM_N_MAX);
    if (pt.s < 10.) { T.delta_n = 1.; T.incr = 2.; }
}

// What follows once the current lobe's n_integration has ended (symphony.rs:127-146).
__device__ __forceinline__ int sym_end_of_lobe(const SymPoint &pt, TaskState &T)
{
    const double c = T.ni_failed ? RIM_NAN : T.ni_ans;
    T.ans += c;
    if (!rim_isfinite(T.ans)) { T.ans = RIM_NAN; return PH_DONE; }
    if (pt.stokes == STOKES_V && T.lobe == 0) {
        sym_start_lobe(pt, T, 1);
        return PH_DERIV1;   // the caller re-checks the while condition for the new lobe
    }
    return PH_DONE;
}

// while-condition of n_integration (symphony.rs:225) + chunk cap
__device__ __forceinline__ int sym_chunk_loop_head(const SymPoint &pt, TaskState &T)
{
    for (;;) {
        if (!(rim_fabs(T.contrib) >= rim_fabs(T.ni_ans / 1e5))) {
            if (sym_end_of_lobe(pt, T) == PH_DONE) return PH_DONE;
            continue;
        }
        if (++T.chunks > RIM_MAX_CHUNKS) {
            T.status |= ST_CHUNK_CAP;
            T.ni_failed = 1;
            if (sym_end_of_lobe(pt, T) == PH_DONE) return PH_DONE;
            continue;
        }
        T.dr_h = 1e-10 * T.n_start;
        return PH_DERIV1;
    }
}

__device__ __forceinline__ void sym_begin(const SymPoint &pt, TaskState &T)
{
    const double n_minus = pt.s * rim_fabs(pt.sin_th);
    const long long n_lo = sat_i64(n_minus + 1.);
    const long long n_hi = sat_i64(n_minus + 1. + RIM_N_MAX);

    T.ans = 0.;
    T.status = ST_OK;
    T.phase = PH_DISCRETE;
    T.disc_next = n_lo;
    T.disc_hi = n_hi;
    T.lobe = 0;
    T.ni_ans = 0.; T.contrib = 0.; T.delta_n = 1e5; T.incr = 10.; T.n_start = 0.;
    T.chunks = 0;
    T.batches = 0;
    T.ni_failed = 0;
    T.dr_h = 0.; T.dr_r0 = 0.; T.dr_err = 0.;
    T.qa = 0.; T.qb = 0.;
    qag_begin(T.oq, 0., 1e-3, 1000);

    if (n_hi <= n_lo) {
        // no discrete harmonics (only for absurd n_minus): straight to the tail
        sym_start_lobe(pt, T, 0);
        T.phase = sym_chunk_loop_head(pt, T);
    }
}

// Post the batch of the current phase.  Returns false when the backstop on the number of
// batches fired (the task is then finished with NaN).
__device__ __forceinline__ bool sym_post(const SymPoint &pt, const GKLane &g, const IStore &outer, TaskState &T, SymBatch &B)
{
    const int lane = g.lane;
    const bool is_v = pt.stokes == STOKES_V;
    B.req_n = 0.;
    B.req_lobe = 0;
    B.req_active = false;
    B.n_req = 0;
    B.phase = T.phase;

    if (++T.batches > RIM_MAX_BATCHES) {
        T.status |= ST_CHUNK_CAP;
        T.ans = RIM_NAN;
        T.phase = PH_DONE;
        return false;
    }

    if (T.phase == PH_DISCRETE) {
        // up to 31 harmonics per batch; for V the two lobes of a harmonic sit on adjacent lanes
        const int per = is_v ? 2 : 1;
        const long long remaining = T.disc_hi - T.disc_next;
        const int cnt = remaining > 31 ? 31 : (int) remaining;
        B.n_req = cnt * per;
        if (lane < B.n_req) {
            B.req_n = (double) (T.disc_next + lane / per);
            B.req_lobe = is_v ? (lane & 1) : 0;
            B.req_active = true;
        }
    } else if (T.phase == PH_DERIV1 || T.phase == PH_DERIV2) {
        // central_deriv: f(x-h), f(x+h), f(x-h/2), f(x+h/2)  (deriv.c order)
        B.n_req = 4;
        if (lane < 4) {
            const double h = T.dr_h;
            B.req_n = (lane == 0) ? T.n_start - h : (lane == 1) ? T.n_start + h
                    : (lane == 2) ? T.n_start - h / 2 : T.n_start + h / 2;
            B.req_lobe = T.lobe;
            B.req_active = true;
        }
    } else if (T.phase == PH_QAG_FIRST) {
        const double center = 0.5 * (T.qa + T.qb);
        const double hl = 0.5 * (T.qb - T.qa);
        B.req_n = center + hl * gk_t(g);
        B.req_lobe = T.lobe;
        B.req_active = g.node && g.half == 0;
    } else {   // PH_QAG_BISECT
        qag_pick(T.oq, outer, lane);
        const double la = g.half ? T.oq.a2 : T.oq.a1;
        const double lb = g.half ? T.oq.b2 : T.oq.b1;
        const double center = 0.5 * (la + lb);
        const double hl = 0.5 * (lb - la);
        B.req_n = center + hl * gk_t(g);
        B.req_lobe = T.lobe;
        B.req_active = g.node;
    }
    return true;
}

// One or two requests: gamma_integral(n) for the given lobe (symphony.rs:312-389).  THE integrand site.
// Two requests share their first rule application (wave_qag_pair); each half-wave also computes its own
// request's order data and limits.
template <int KIND, int PREC = 0>
__device__ __forceinline__ void sym_eval_pair(const SymPoint &pt, const DistParams &dist, const GKLane &g,
                                              const IStore &inner, QagPark *qpark,
                                              double n0, int lobe0, double n1, int lobe1, bool have1,
                                              double &val0, int &st0, double &val1, int &st1)
{
    const int lane = g.lane;
    RIM_HIT(21);
    RIM_PROF_T(t_setup);
    // ---- setup: every lane computes data of its half's request -- even lanes the order record of n, odd lanes that of
    // n + 1 (one leung_order per lane instead of two: the same values as sym_order, half the instructions) --, and
    // lanes 0, 1, 32, 33 file them in LDS ----
    const bool second = g.half != 0 && have1;
    const double n_l = second ? n1 : n0;
    const double n_mine = (lane & 1) ? n_l + 1. : n_l;
    LeungOrder ord_mine = LeungOrder();
    if (!(n_mine < 30.)) ord_mine = leung_order(n_mine);
    SymOrder so_l;
    so_l.small = n_l < 30.;
    so_l.np1_small = (n_l + 1.) < 30.;
    so_l.dj_nan = n_l >= 1e15;
    const GammaLimits L = gamma_limits(pt, n_l, second ? lobe1 : lobe0);
    wv_sync();                       // nobody is still reading the previous requests' records
    if ((lane & 30) == 0) qpark->ord[(lane >> 4) | (lane & 1)] = ord_mine;          // lanes 0, 1, 32, 33 -> records 0, 1, 2, 3
    wv_sync();
    const double a0 = readlane_d(L.g0, 0), b0 = readlane_d(L.g1, 0);
    const double a1 = readlane_d(L.g0, 32), b1 = readlane_d(L.g1, 32);
    const int fl_l = (so_l.small ? 1 : 0) | (so_l.np1_small ? 2 : 0) | (so_l.dj_nan ? 4 : 0);
    const int fl0 = wv_readlane(fl_l, 0), fl1 = wv_readlane(fl_l, 32);
    n0 = uni(n0); n1 = uni(n1);
    RIM_PROF_ADD(7, t_setup);

    auto f = [&](double x, bool active, bool second) __attribute__((always_inline)) -> double {
        SymOrder so;
        const int fl = second ? fl1 : fl0;
        so.n = second ? n1 : n0;
        so.small = (fl & 1) != 0; so.np1_small = (fl & 2) != 0; so.dj_nan = (fl & 4) != 0;
        so.o = qpark->ord + (second ? 2 : 0);
        return active ? gamma_integrand<KIND, PREC>(pt, dist, so, x) : 0.;
    };
    int qs0, qs1;
    wave_qag_pair(f, g, inner, a0, b0, a1, b1, have1, false, 0., 1e-3, 5000, qpark, val0, qs0, val1, qs1);
    if (qs0 != QAG_SUCCESS) {
        val0 = RIM_NAN;
        st0 |= ST_INNER_FAIL;
        if (qs0 == QAG_ESTORE) st0 |= ST_STORE_FULL;
    }
    if (have1 && qs1 != QAG_SUCCESS) {
        val1 = RIM_NAN;
        st1 |= ST_INNER_FAIL;
        if (qs1 == QAG_ESTORE) st1 |= ST_STORE_FULL;
    }
}

template <int KIND>
__device__ __forceinline__ double sym_eval_request(const SymPoint &pt, const DistParams &dist, const GKLane &g,
                                                   const IStore &inner, QagPark *qpark, double n, int lobe, int &st)
{
    double v0, v1;
    int s1 = 0;
    sym_eval_pair<KIND>(pt, dist, g, inner, qpark, n, lobe, n, lobe, false, v0, st, v1, s1);
    return v0;
}

// Continuation of the phase that posted B, given the results (lane k holds the value of request k).
__device__ __forceinline__ void sym_consume(const SymPoint &pt, const GKLane &g, const IStore &outer, TaskState &T,
                                            const SymBatch &B, double gval, int batch_status)
{
    const int lane = g.lane;
    const int phase = B.phase;
    const bool is_v = pt.stokes == STOKES_V;
    T.status |= batch_status;

    if (phase == PH_DISCRETE) {
        for (int k = 0; k < B.n_req; k++) T.ans += readlane_d(gval, k);
        const int per = is_v ? 2 : 1;
        T.disc_next += B.n_req / per;
        if (T.disc_next < T.disc_hi) return;
        if (!rim_isfinite(T.ans)) { T.ans = RIM_NAN; T.phase = PH_DONE; return; }
        sym_start_lobe(pt, T, 0);
        T.phase = sym_chunk_loop_head(pt, T);
        return;
    }

    if (phase == PH_DERIV1 || phase == PH_DERIV2) {
        const double fm1 = readlane_d(gval, 0), fp1 = readlane_d(gval, 1);
        const double fmh = readlane_d(gval, 2), fph = readlane_d(gval, 3);
        const double h = T.dr_h;
        const double x = T.n_start;
        const double r3 = 0.5 * (fp1 - fm1);
        const double r5 = (4.0 / 3.0) * (fph - fmh) - (1.0 / 3.0) * r3;
        const double e3 = (rim_fabs(fp1) + rim_fabs(fm1)) * RIM_DBL_EPSILON;
        const double e5 = 2.0 * (rim_fabs(fph) + rim_fabs(fmh)) * RIM_DBL_EPSILON + e3;
        const double dy = rim_max(rim_fabs(r3 / h), rim_fabs(r5 / h)) * (rim_fabs(x) / h) * RIM_DBL_EPSILON;
        const double res = r5 / h;
        const double trunc = rim_fabs((r5 - r3) / h);
        const double round = rim_fabs(e5 / h) + dy;

        if (phase == PH_DERIV1) {
            T.dr_r0 = res;
            T.dr_err = round + trunc;
            if (round < trunc && (round > 0 && trunc > 0)) {
                T.dr_h = (1e-10 * T.n_start) * rim_pow(round / (2.0 * trunc), 1.0 / 3.0);
                T.phase = PH_DERIV2;
                return;
            }
        } else {
            const double error_opt = round + trunc;
            if (error_opt < T.dr_err && rim_fabs(res - T.dr_r0) < 4.0 * T.dr_err) {
                T.dr_r0 = res;
                T.dr_err = error_opt;
            }
        }
        const double deriv = T.dr_r0;

        if (deriv == 0. || (T.contrib != 0. && rim_fabs(deriv / T.contrib) < 1e-5)) T.delta_n *= T.incr;
        if (T.delta_n < T.n_start / T.incr) T.delta_n *= T.incr;

        T.qa = T.n_start;
        T.qb = T.n_start + T.delta_n;
        qag_begin(T.oq, 0., 1e-3, 1000);
        T.phase = PH_QAG_FIRST;
        return;
    }

    // outer GK31 over the values now sitting in the lanes
    bool chunk_done;
    if (phase == PH_QAG_FIRST) {
        const double hl = 0.5 * (T.qb - T.qa);
        const GKRes r = wave_gk31(gval, hl, g);
        chunk_done = qag_after_first(T.oq, outer, lane, T.qa, T.qb, readlane_d(r.result, 0), readlane_d(r.abserr, 0),
                                     readlane_d(r.resabs, 0), readlane_d(r.resasc, 0));
    } else {
        const double la = g.half ? T.oq.a2 : T.oq.a1;
        const double lb = g.half ? T.oq.b2 : T.oq.b1;
        const double hl = 0.5 * (lb - la);
        const GKRes r = wave_gk31(gval, hl, g);
        chunk_done = qag_after_bisect(T.oq, outer, lane,
                                      readlane_d(r.result, 0), readlane_d(r.abserr, 0), readlane_d(r.resasc, 0),
                                      readlane_d(r.result, 32), readlane_d(r.abserr, 32), readlane_d(r.resasc, 32));
    }
    if (!chunk_done) { T.phase = PH_QAG_BISECT; return; }

    if (T.oq.status != QAG_SUCCESS) {
        // `?` in n_integration (symphony.rs:269): the whole n integration is an Err
        T.status |= ST_OUTER_FAIL;
        if (T.oq.status == QAG_ESTORE) T.status |= ST_STORE_FULL;
        T.ni_failed = 1;
        T.phase = sym_end_of_lobe(pt, T);
        if (T.phase != PH_DONE) T.phase = sym_chunk_loop_head(pt, T);
        return;
    }
    T.contrib = T.oq.result;
    T.ni_ans += T.contrib;
    T.n_start += T.delta_n;
    if (T.n_start > 1e13) T.incr = 1.;
    T.phase = sym_chunk_loop_head(pt, T);
}

// Final NaN policy and the dimensional prefactor (symphony.rs:144-183).
__device__ __forceinline__ double sym_result(const SymPoint &pt, const TaskState &T, int &status_out)
{
    status_out = T.status;
    double ans = T.ans;
    if (!rim_isfinite(ans)) {
        status_out |= ST_NONFINITE;
        return RIM_NAN;
    }
    const double tpe = RIM_TWO_PI * RIM_ELECTRON_CHARGE;
    const double acos_th = rim_fabs(pt.cos_th);
    if (pt.coeff == COEFF_EMISSION)
        ans = ans * ((tpe * tpe) / (RIM_SPEED_LIGHT * acos_th));
    else
        ans = ans * (-1. * (tpe * tpe) / (2. * RIM_MASS_ELECTRON * RIM_SPEED_LIGHT * acos_th));
    return ans;
}

// The whole coefficient on one wave (no cooperation): used by the wavefront emulator and kept as
// the readable statement of the control flow.
template <int KIND>
__device__ __forceinline__ double symphony_coefficient(const SymPoint &pt, const DistParams &dist, const GKLane &g,
                                                       const IStore &inner, const IStore &outer, TaskState *park,
                                                       QagPark *qpark, int &status_out)
{
    const int lane = g.lane;
    TaskState T;
    sym_begin(pt, T);
    while (T.phase != PH_DONE) {
        SymBatch B;
        if (!sym_post(pt, g, outer, T, B)) break;
        if (lane == 0) *park = T;        // park the uniform state in LDS while the batch runs
        int batch_status = 0;
        double gval = 0.;
        unsigned long long mask = wv_ballot(B.req_active);
        while (mask) {
            const int k = __builtin_ffsll((long long) mask) - 1;
            mask &= mask - 1;
            const double n = readlane_d(B.req_n, k);
            const int lb = wv_readlane(B.req_lobe, k);
            int k2 = -1;
            double n2 = n;
            int lb2 = lb;
            if (mask) {
                k2 = __builtin_ffsll((long long) mask) - 1;
                mask &= mask - 1;
                n2 = readlane_d(B.req_n, k2);
                lb2 = wv_readlane(B.req_lobe, k2);
            }
            double val, val2;
            int st = 0, st2 = 0;
            sym_eval_pair<KIND>(pt, dist, g, inner, qpark, n, lb, n2, lb2, k2 >= 0, val, st, val2, st2);
            batch_status |= st | st2;
            if (lane == k) gval = val;
            if (lane == k2) gval = val2;
        }
        wv_sync();
        T = *park;
        task_uniformize(T);
        sym_consume(pt, g, outer, T, B, gval, uni(batch_status));
    }
    return sym_result(pt, T, status_out);
}

}  // namespace rim
#endif
