// heyvaerts_wave.h -- one wavefront computes one Faraday coefficient (rho_Q or rho_V).
//
//   heyvaerts::CalculationState::compute   heyvaerts.rs:81-191
//     non-resonant part: outer QAG over pomega, inner QAG over sigma   :85-151, 204-250
//     quasi-resonant part: outer QAG over sigma, inner QAG over pomega :156-185, 253-296
//     chunk marching with gsl::deriv_central(h = 1e-6) step control    :102-185
//
// Same structure as symphony_wave.h: the inner integrals are the only place with
// lane parallelism, so the wave runs them one after another and everything above
// them is a wave-uniform state machine whose phases post batches of outer
// abscissae (one per lane).  Operation order follows oracle/rimo_heyvaerts.c.
#ifndef RIM_HEYVAERTS_WAVE_H
#define RIM_HEYVAERTS_WAVE_H

#include "symphony_wave.h"
#include "dev_heyvaerts.h"

namespace rim {

#define RIM_HEY_MAX_STEPS 4096

enum HeyStage { HS_NR_FIRST = 0, HS_NR_RIGHT, HS_NR_LEFT, HS_QR, HS_DONE };
enum HeyPhase { HP_DERIV1 = 0, HP_DERIV2, HP_QAG_FIRST, HP_QAG_BISECT,
                HP_QAG_STASHED };   // (batch phase only) a bisection whose children's rule sums were on file: no requests

struct HeyTask {
    double nr_val, qr_val;
    double pomega_left, pomega_right, delta_left, delta_right, sigma_low, delta_sigma;
    double dr_x, dr_h, dr_r0, dr_err;
    double qa, qb;
    double result;
    int stage, phase, steps, status, batches, failed;
    QagState oq;
};

__device__ __forceinline__ void hey_uniformize(HeyTask &T)
{
    T.nr_val = uni(T.nr_val); T.qr_val = uni(T.qr_val);
    T.pomega_left = uni(T.pomega_left); T.pomega_right = uni(T.pomega_right);
    T.delta_left = uni(T.delta_left); T.delta_right = uni(T.delta_right);
    T.sigma_low = uni(T.sigma_low); T.delta_sigma = uni(T.delta_sigma);
    T.dr_x = uni(T.dr_x); T.dr_h = uni(T.dr_h); T.dr_r0 = uni(T.dr_r0); T.dr_err = uni(T.dr_err);
    T.qa = uni(T.qa); T.qb = uni(T.qb); T.result = uni(T.result);
    T.stage = uni(T.stage); T.phase = uni(T.phase); T.steps = uni(T.steps); T.status = uni(T.status);
    T.batches = uni(T.batches); T.failed = uni(T.failed);
    qag_uniformize(T.oq);
}

// ---- rounds: the long outer quadratures, several bisections per batch -----------------------------------------
// An outer quadrature that does not converge works through its list breadth first: measured with the oracle on the
// three 4096-subinterval quadratures of the bench tables (profiles/r4_outer_pick_statistics.txt), the interval qag.c picks
// next is a child of the one just bisected in 1-2 % of the steps.  The rule sums of an interval's children are a function
// of the interval alone, so they can be computed BEFORE qag.c asks for them: when the picked interval's sums are not on
// file, the batch carries -- besides its two children -- the children of the next RIM_TURBO_MAX - 1 intervals in qag.c's
// own order (descending error, stamp), and their sums are filed in the wave's HeyStash; a later pick that finds its sums
// there is booked without a batch.  qag.c's sequence of picks, sums, round-off counters and decisions is untouched (the
// mechanism of symphony_group.h's stash, one level up); a filed entry that is never picked -- the quadrature ended --
// is dropped together with its status bits and sample counts, so values, status words and work counters do not depend on
// whether rounds were used.  3.9 bisections per batch on those quadratures: the chain of sequential batches that bounds
// the end of a launch is that much shorter.
#define RIM_HEY_STASH 8
#define RIM_ROUND_MIN_SIZE 48
struct HeyStash {
    double a[RIM_HEY_STASH], b[RIM_HEY_STASH];          // the interval (bit patterns compared); used bit in `used`
    double v[RIM_HEY_STASH][6];                         // area1, error1, resasc1, area2, error2, resasc2 (lane 0's view)
    int status[RIM_HEY_STASH];                          // status bits of the entry's 62 inner integrals
    unsigned long long samples[RIM_HEY_STASH];          // their integrand samples (work counters)
    unsigned used;
    // the batch in flight: the intervals besides the picked one whose children it evaluates (round_n = 1 + their number),
    // and the place of the picked interval's sums in the stash when they were on file (-1: not)
    int round_n, hit;
    double ra[RIM_TURBO_MAX - 1], rb[RIM_TURBO_MAX - 1];
};
__device__ __forceinline__ int hey_stash_find(const HeyStash *hs, double a, double b, int lane)
{
    const int k = lane & (RIM_HEY_STASH - 1);
    const unsigned long long hit = wv_ballot(lane < RIM_HEY_STASH && ((uni(hs->used) >> k) & 1u) &&
                                             rim_bits(hs->a[k]) == rim_bits(a) && rim_bits(hs->b[k]) == rim_bits(b));
    return hit ? __builtin_ffsll((long long) hit) - 1 : -1;
}

// ---- the coefficient as a resumable computation (same shape as symphony_wave.h) -----------
//
//   hey_begin         initialise the task state
//   hey_post          stage/phase -> a batch of up to 62 outer abscissae, one per lane (B.req_n = abscissa,
//                     B.req_lobe = 1 for the quasi-resonant regime)
//   hey_eval_request  nr_outer_integrand / qr_outer_integrand: THE site of the inner QAG and the integrand
//   hey_consume       continuation on the results; T.stage == HS_DONE at the end
//   hey_result        failure policy
// so that the cooperative kernel (rimphony_hip.hip) can hand the requests of a batch to other waves.

#define RIM_HEY_TOL 1e-5
#define RIM_HEY_DELTA_SCALE 5.

__device__ __forceinline__ void hey_start_chunk(HeyTask &T, double a, double b)
{
    T.qa = a; T.qb = b;
    qag_begin(T.oq, 0., 1e-3, 4096);
    T.phase = HP_QAG_FIRST;
}
__device__ __forceinline__ void hey_start_deriv(HeyTask &T, double x)
{
    T.dr_x = x;
    T.dr_h = 1e-6;
    T.phase = HP_DERIV1;
}
__device__ __forceinline__ void hey_fail(HeyTask &T) { T.failed = 1; T.stage = HS_DONE; }

// head of the marching loop of the current stage (the `while keep_going` tests of heyvaerts.rs:102,134,161)
__device__ __forceinline__ void hey_loop_head(const HeyPoint &pt, HeyTask &T)
{
    if (++T.steps > RIM_HEY_MAX_STEPS) { T.status |= ST_CHUNK_CAP; hey_fail(T); return; }
    if (T.stage == HS_NR_RIGHT) {
        if (T.nr_val != 0.) hey_start_deriv(T, T.pomega_right);
        else hey_start_chunk(T, T.pomega_right, T.pomega_right + T.delta_right);
    } else if (T.stage == HS_NR_LEFT) {
        hey_start_deriv(T, T.pomega_left);
    } else {   // HS_QR
        if (T.qr_val != 0.) hey_start_deriv(T, T.sigma_low);
        else if (hey_qr_is_endless(pt, T.sigma_low)) { T.status |= ST_CHUNK_CAP; hey_fail(T); }      // dev_heyvaerts.h
        else hey_start_chunk(T, T.sigma_low, T.sigma_low + T.delta_sigma);
    }
}

__device__ __forceinline__ void hey_begin(const HeyPoint &pt, HeyTask &T)
{
    T.nr_val = 0.; T.qr_val = 0.;
    T.pomega_left = -3. * pt.sigma0;
    T.pomega_right = 3. * pt.sigma0;
    T.delta_left = T.pomega_right;
    T.delta_right = T.pomega_right;
    T.sigma_low = 0.; T.delta_sigma = 0.;
    T.dr_x = 0.; T.dr_h = 0.; T.dr_r0 = 0.; T.dr_err = 0.;
    T.result = RIM_NAN;
    T.stage = HS_NR_FIRST;
    T.phase = HP_QAG_FIRST;
    T.steps = 0; T.status = ST_OK; T.batches = 0; T.failed = 0;
    T.qa = T.pomega_left;
    T.qb = T.pomega_right;
    qag_begin(T.oq, 0., 1e-3, 4096);
}

// The next `want` intervals of the outer list in qag.c's order after the picked one -- descending (error, stamp), the
// order of its sorted list -- that are not on file already: hs->ra / rb, hs->round_n = 1 + their number.
__device__ __forceinline__ void hey_round_select(const IStore &outer, const HeyTask &T, HeyStash *hs, int want, int lane)
{
    double pe = T.oq.e_i;                       // the picked interval: the largest (error, stamp)
    int pstamp = uni(ist_stamp(outer, T.oq.imax));
    int n = 1;
    for (int tries = 0; tries < want + RIM_HEY_STASH && n - 1 < want; tries++) {
        // wave-wide argmax over the entries strictly below (pe, pstamp)
        double be = -1.0;
        int bs = -1, bi = -1;
        for (int i = lane; i < T.oq.size; i += 64) {
            const double e = ist_e(outer, i);
            const int s = ist_stamp(outer, i);
            const bool below = e < pe || (e == pe && s < pstamp);
            if (below && (e > be || (e == be && s > bs))) { be = e; bs = s; bi = i; }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const double oe = wv_shfl_xor(be, m);
            const int os = wv_shfl_xor(bs, m);
            const int oi = wv_shfl_xor(bi, m);
            if (oe > be || (oe == be && os > bs)) { be = oe; bs = os; bi = oi; }
        }
        const int idx = wv_readfirstlane(bi);
        if (idx < 0) break;                     // the list has no further entry
        pe = uni(be); pstamp = wv_readfirstlane(bs);
        const double a = uni(ist_a(outer, idx)), b = uni(ist_b(outer, idx));
        if (hey_stash_find(hs, a, b, lane) >= 0) continue;      // its children's sums are on file
        if (lane == 0) { hs->ra[n - 1] = a; hs->rb[n - 1] = b; }
        n++;
    }
    if (lane == 0) hs->round_n = n;
    wv_sync();
}

// Post the next batch.  Returns false (and ends the task) when the batch backstop is hit.
// hs / rounds: the wave's stash and how many further intervals this batch may evaluate besides the picked one (0: none,
// and hs may be null) -- the Faraday kernel's long outer quadratures, see HeyStash above.
__device__ __forceinline__ bool hey_post(const HeyPoint &pt, const GKLane &g, const IStore &outer, HeyTask &T, SymBatch &B,
                                         HeyStash *hs = nullptr, int rounds = 0)
{
    const int lane = g.lane;
    B.req_n = 0.; B.req_lobe = T.stage == HS_QR ? 1 : 0; B.req_active = false; B.n_req = 0; B.phase = T.phase;
    if (hs && (uni(hs->round_n) != 1 || uni(hs->hit) >= 0)) {       // whatever the previous batch was: this one starts plain
        wv_sync();
        if (lane == 0) { hs->round_n = 1; hs->hit = -1; }
        wv_sync();
    }
    if (++T.batches > RIM_MAX_BATCHES) { T.status |= ST_CHUNK_CAP; hey_fail(T); return false; }
    if (T.phase == HP_DERIV1 || T.phase == HP_DERIV2) {
        B.n_req = 4;
        if (lane < 4) {
            const double h = T.dr_h, x = T.dr_x;
            B.req_n = (lane == 0) ? x - h : (lane == 1) ? x + h : (lane == 2) ? x - h / 2 : x + h / 2;
            B.req_active = true;
        }
    } else if (T.phase == HP_QAG_FIRST) {
        const double center = 0.5 * (T.qa + T.qb);
        const double hl = 0.5 * (T.qb - T.qa);
        B.req_n = center + hl * gk_t(g);
        B.req_active = g.node && g.half == 0;
    } else {
        qag_pick(T.oq, outer, lane);
        if (hs) {
            const int hit = uni(hs->used) ? hey_stash_find(hs, T.oq.a1, T.oq.b2, lane) : -1;
            if (hit >= 0) {                                               // booked without a batch
                wv_sync();
                if (lane == 0) hs->hit = hit;
                wv_sync();
                B.phase = HP_QAG_STASHED;
                return true;
            }
        }
        const double la = g.half ? T.oq.a2 : T.oq.a1;
        const double lb = g.half ? T.oq.b2 : T.oq.b1;
        const double center = 0.5 * (la + lb);
        const double hl = 0.5 * (lb - la);
        B.req_n = center + hl * gk_t(g);
        B.req_active = g.node;
        // (only a quadrature that has grown past RIM_ROUND_MIN_SIZE subintervals: one that converges would leave the
        // further intervals' sums unused)
        if (hs && rounds > 0 && T.oq.size >= RIM_ROUND_MIN_SIZE) {
            const int room = RIM_HEY_STASH - __builtin_popcount(uni(hs->used));
            const int want = rounds < room ? rounds : room;
            if (want > 0) hey_round_select(outer, T, hs, want, lane);
        }
    }
    return true;
}

// The abscissa of the lane's request for further interval j (1 <= j < round_n) of a round: the same expression as
// hey_post's for the picked interval.
__device__ __forceinline__ double hey_round_request(const HeyStash *hs, const GKLane &g, int j)
{
    const double a = hs->ra[j - 1], b = hs->rb[j - 1];
    const double mid = 0.5 * (a + b);
    const double la = g.half ? mid : a;
    const double lb = g.half ? b : mid;
    const double center = 0.5 * (la + lb);
    const double hl = 0.5 * (lb - la);
    return center + hl * gk_t(g);
}

// Where the results of a round's further intervals are read from (the owner's board slot): request `rank` of further
// interval j sits at per * j + rank.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RIM_WAVE_EMU)
#define HEY_IO_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define HEY_IO_LOAD(p) (*(p))
#endif
struct HeyRoundIO {
    const unsigned long long *res;
    const int *res_status;
    const unsigned *res_samples;
    int rank, per;
};

// One or two requests: the inner integral at outer abscissa u (heyvaerts.rs:213-250 non-resonant, 262-296
// quasi-resonant).  THE site of the inner QAG and the integrand.  Two requests share their first rule
// application (wave_qag_pair); a request whose sigma range is empty contributes 0 without a quadrature
// (heyvaerts.rs:224-226), and the other one then runs alone.
template <int KIND>
__device__ __forceinline__ void hey_eval_pair(const HeyPoint &pt, const DistParams &dist, const HeyConsts &hc,
                                              const GKLane &g, const IStore &inner, QagParkBase *qpark,
                                              double u0, int qr0, double u1, int qr1, bool have1,
                                              double &val0, int &st0, double &val1, int &st1)
{
    // every lane sets up the request of its half
    const bool sec = g.half != 0 && have1;
    const double u_l = sec ? u1 : u0;
    const bool qr_l = (sec ? qr1 : qr0) != 0;
    double lo_l, hi_l;
    bool empty_l = false;
    if (!qr_l) {
        const double sigma_min = rim_sqrt(u_l * u_l + pt.sigma0_sq);
        const double sigma_max = RIM_INVERSE_SQRT_3 * rim_pow15(sigma_min);       // sigma_min = sqrt(..) >= 0
        empty_l = sigma_max <= sigma_min;
        lo_l = sigma_min; hi_l = sigma_max;
    } else {
        const double pomega_max_phys = rim_sqrt(RIM_THREE_TWO_THIRDS * rim_pow43(u_l) - pt.sigma0_sq);
        const double pomega_max_qr = rim_sqrt(u_l * u_l - pt.sigma0_sq);
        const double pomega_max = rust_min(pomega_max_phys, pomega_max_qr);
        lo_l = -pomega_max; hi_l = pomega_max;
    }
    const double a0 = readlane_d(lo_l, 0), b0 = readlane_d(hi_l, 0);
    const double a1 = readlane_d(lo_l, 32), b1 = readlane_d(hi_l, 32);
    const bool todo0 = wv_readlane(empty_l ? 1 : 0, 0) == 0;
    const bool todo1 = have1 && wv_readlane(empty_l ? 1 : 0, 32) == 0;
    u0 = uni(u0); u1 = uni(u1);
    const bool q0 = uni(qr0) != 0, q1 = uni(qr1) != 0;
    val0 = 0.; val1 = 0.;
    if (!todo0 && !todo1) return;

    auto f = [&](double v, bool active, bool second) __attribute__((always_inline)) -> double {
        return active ? hey_element<KIND>(pt, dist, hc, second ? q1 : q0, second ? u1 : u0, v) : 0.;
    };
    // both: one joint run.  Only one of them: it runs as "integral 0" (flip maps it to its own data).
    const bool both = todo0 && todo1;
    const bool flip = !todo0;
    double r0, r1;
    int qs0, qs1;
    wave_qag_pair(f, g, inner, flip ? a1 : a0, flip ? b1 : b0, a1, b1, both, flip, 0., 1e-3, 4096, qpark, r0, qs0, r1, qs1);
    if (flip) { r1 = r0; qs1 = qs0; }
    if (todo0) {
        val0 = r0;
        if (qs0 != QAG_SUCCESS) {
            val0 = RIM_NAN;
            st0 |= ST_INNER_FAIL;
            if (qs0 == QAG_ESTORE) st0 |= ST_STORE_FULL;
        }
    }
    if (todo1) {
        val1 = r1;
        if (qs1 != QAG_SUCCESS) {
            val1 = RIM_NAN;
            st1 |= ST_INNER_FAIL;
            if (qs1 == QAG_ESTORE) st1 |= ST_STORE_FULL;
        }
    }
}

template <int KIND>
__device__ __forceinline__ double hey_eval_request(const HeyPoint &pt, const DistParams &dist, const HeyConsts &hc,
                                                   const GKLane &g, const IStore &inner, QagParkBase *qpark,
                                                   double u, int qr_flag, int &st)
{
    double v0, v1;
    int s1 = 0;
    hey_eval_pair<KIND>(pt, dist, hc, g, inner, qpark, u, qr_flag, u, qr_flag, false, v0, st, v1, s1);
    return v0;
}

// Continuation of the phase that posted B (lane k holds the value of request k).
// hs: the wave's stash (rounds; null: none).  stash_samples: out, the integrand samples of a bisection that was booked
// from the stash (they count for the coefficient when its sums are consumed); dropped_samples: out, += the samples of
// filed sums that the end of the quadrature left unused.
__device__ __forceinline__ void hey_consume(const HeyPoint &pt, const GKLane &g, const IStore &outer, HeyTask &T,
                                            const SymBatch &B, double gval, int batch_status, HeyStash *hs = nullptr,
                                            unsigned long long *stash_samples = nullptr, unsigned long long *dropped_samples = nullptr,
                                            const HeyRoundIO *io = nullptr)
{
    const int lane = g.lane;
    const int phase = B.phase;
    T.status |= batch_status;

    if (phase == HP_DERIV1 || phase == HP_DERIV2) {
        const double fm1 = readlane_d(gval, 0), fp1 = readlane_d(gval, 1);
        const double fmh = readlane_d(gval, 2), fph = readlane_d(gval, 3);
        const double h = T.dr_h;
        const double x = T.dr_x;
        const double r3 = 0.5 * (fp1 - fm1);
        const double r5 = (4.0 / 3.0) * (fph - fmh) - (1.0 / 3.0) * r3;
        const double e3 = (rim_fabs(fp1) + rim_fabs(fm1)) * RIM_DBL_EPSILON;
        const double e5 = 2.0 * (rim_fabs(fph) + rim_fabs(fmh)) * RIM_DBL_EPSILON + e3;
        const double dy = rim_max(rim_fabs(r3 / h), rim_fabs(r5 / h)) * (rim_fabs(x) / h) * RIM_DBL_EPSILON;
        const double res = r5 / h;
        const double trunc = rim_fabs((r5 - r3) / h);
        const double round = rim_fabs(e5 / h) + dy;

        if (phase == HP_DERIV1) {
            T.dr_r0 = res;
            T.dr_err = round + trunc;
            if (round < trunc && (round > 0 && trunc > 0)) {
                T.dr_h = 1e-6 * rim_pow(round / (2.0 * trunc), 1.0 / 3.0);
                T.phase = HP_DERIV2;
                return;
            }
        } else {
            const double error_opt = round + trunc;
            if (error_opt < T.dr_err && rim_fabs(res - T.dr_r0) < 4.0 * T.dr_err) {
                T.dr_r0 = res;
                T.dr_err = error_opt;
            }
        }
        const double rel_deriv = T.dr_r0;

        if (T.stage == HS_NR_RIGHT) {
            if (rel_deriv == 0. || rim_fabs(1. / (rel_deriv * T.delta_right)) > RIM_HEY_DELTA_SCALE)
                T.delta_right *= RIM_HEY_DELTA_SCALE;
            hey_start_chunk(T, T.pomega_right, T.pomega_right + T.delta_right);
        } else if (T.stage == HS_NR_LEFT) {
            if (rel_deriv == 0. || rim_fabs(1. / (rel_deriv * T.delta_left)) > RIM_HEY_DELTA_SCALE)
                T.delta_left *= RIM_HEY_DELTA_SCALE;
            hey_start_chunk(T, T.pomega_left - T.delta_left, T.pomega_left);
        } else {
            if (rel_deriv == 0. || rim_fabs(1. / (rel_deriv * T.delta_sigma)) > RIM_HEY_DELTA_SCALE) {
                if (T.delta_sigma < 1e6 * pt.sigma0) T.delta_sigma *= RIM_HEY_DELTA_SCALE;
            }
            hey_start_chunk(T, T.sigma_low, T.sigma_low + T.delta_sigma);
        }
        return;
    }

    bool chunk_done;
    if (phase == HP_QAG_FIRST) {
        const double hl = 0.5 * (T.qb - T.qa);
        const GKRes r = wave_gk31(gval, hl, g);
        chunk_done = qag_after_first(T.oq, outer, lane, T.qa, T.qb, readlane_d(r.result, 0), readlane_d(r.abserr, 0),
                                     readlane_d(r.resabs, 0), readlane_d(r.resasc, 0));
    } else {
        // A bisection.  The children's rule sums of the picked interval come from this batch's values -- or from the
        // stash, when an earlier round evaluated them (HP_QAG_STASHED) -- and go to qag.c's loop body; a round's
        // further intervals (j >= 1: values from the board) get the same arithmetic and go to the stash.  One site of
        // the rule sums and one of the loop body for all of them.
        const bool stashed = phase == HP_QAG_STASHED;
        const int nr = (hs && io && !stashed) ? uni(hs->round_n) : 1;
        chunk_done = false;
        unsigned long long unused = 0;          // per lane: samples of further intervals that nobody will ask for
        for (int j = 0; j < nr; j++) {
            double a = T.oq.a1, b = T.oq.b2, gv = gval;
            int stl = 0;
            unsigned smp = 0;
            if (j > 0) {
                a = uni(hs->ra[j - 1]); b = uni(hs->rb[j - 1]);
                gv = 0.;
                if (g.node) {
                    gv = rim_frombits(HEY_IO_LOAD(&io->res[io->per * j + io->rank]));
                    stl = HEY_IO_LOAD(&io->res_status[io->per * j + io->rank]);
                    smp = HEY_IO_LOAD(&io->res_samples[io->per * j + io->rank]);
                }
                if (chunk_done) { unused += smp; continue; }      // the quadrature ended with the picked interval
            }
            double a1, e1, s1, a2, e2, s2;
            if (stashed) {
                const int k = uni(hs->hit);
                a1 = uni(hs->v[k][0]); e1 = uni(hs->v[k][1]); s1 = uni(hs->v[k][2]);
                a2 = uni(hs->v[k][3]); e2 = uni(hs->v[k][4]); s2 = uni(hs->v[k][5]);
                T.status |= uni(hs->status[k]);
                if (stash_samples) {
                    const unsigned long long sm = hs->samples[k];
                    *stash_samples = ((unsigned long long) (unsigned) wv_readfirstlane((int) (unsigned) (sm >> 32)) << 32) |
                                     (unsigned) wv_readfirstlane((int) (unsigned) sm);
                }
                const unsigned used = uni(hs->used);
                wv_sync();
                if (lane == 0) hs->used = used & ~(1u << k);
                wv_sync();
            } else {
                const double mid = 0.5 * (a + b);
                const double la = g.half ? mid : a;
                const double lb = g.half ? b : mid;
                const double hl = 0.5 * (lb - la);
                const GKRes r = wave_gk31(gv, hl, g);
                a1 = readlane_d(r.result, 0); e1 = readlane_d(r.abserr, 0); s1 = readlane_d(r.resasc, 0);
                a2 = readlane_d(r.result, 32); e2 = readlane_d(r.abserr, 32); s2 = readlane_d(r.resasc, 32);
            }
            if (j == 0) {
                chunk_done = qag_after_bisect(T.oq, outer, lane, a1, e1, s1, a2, e2, s2);
                continue;
            }
            // file the further interval's sums, with the status bits and the sample count of its 62 inner integrals
            int st = 0;
            if (wv_ballot((stl & ST_INNER_FAIL) != 0)) st |= ST_INNER_FAIL;
            if (wv_ballot((stl & ST_STORE_FULL) != 0)) st |= ST_STORE_FULL;
            unsigned long long tot = smp;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const unsigned lo = (unsigned) wv_shfl_xor((int) (unsigned) tot, m), hi = (unsigned) wv_shfl_xor((int) (unsigned) (tot >> 32), m);
                tot += ((unsigned long long) hi << 32) | lo;
            }
            const unsigned used = uni(hs->used);
            const int pos = __builtin_ffs((int) (~used & ((1u << RIM_HEY_STASH) - 1u))) - 1;      // hey_post left room
            wv_sync();
            if (lane == 0 && pos >= 0) {
                hs->a[pos] = a; hs->b[pos] = b;
                hs->v[pos][0] = a1; hs->v[pos][1] = e1; hs->v[pos][2] = s1;
                hs->v[pos][3] = a2; hs->v[pos][4] = e2; hs->v[pos][5] = s2;
                hs->status[pos] = st;
                hs->samples[pos] = tot;
                hs->used = used | (1u << pos);
            }
            wv_sync();
        }
        if (nr > 1 && chunk_done && dropped_samples) {
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const unsigned lo = (unsigned) wv_shfl_xor((int) (unsigned) unused, m), hi = (unsigned) wv_shfl_xor((int) (unsigned) (unused >> 32), m);
                unused += ((unsigned long long) hi << 32) | lo;
            }
            *dropped_samples += ((unsigned long long) (unsigned) wv_readfirstlane((int) (unsigned) (unused >> 32)) << 32) |
                                (unsigned) wv_readfirstlane((int) (unsigned) unused);
        }
    }
    if (!chunk_done) { T.phase = HP_QAG_BISECT; return; }
    if (hs && uni(hs->used)) {
        // the quadrature is over: whatever is still on file belongs to it and is dropped -- the samples that went into it
        // were never part of the coefficient (the launch's sample count is the reference's: dropped_samples come off it)
        const unsigned used = uni(hs->used);
        unsigned long long drop = 0;
        for (int k = 0; k < RIM_HEY_STASH; k++)
            if ((used >> k) & 1u) drop += hs->samples[k];
        if (dropped_samples) *dropped_samples += ((unsigned long long) (unsigned) wv_readfirstlane((int) (unsigned) (drop >> 32)) << 32) |
                                                  (unsigned) wv_readfirstlane((int) (unsigned) drop);
        wv_sync();
        if (lane == 0) hs->used = 0;
        wv_sync();
    }

    // .unwrap_or(NAN) of the outer integral (heyvaerts.rs:204-211, 253-260)
    double contrib = T.oq.result;
    if (T.oq.status != QAG_SUCCESS) {
        contrib = RIM_NAN;
        T.status |= ST_OUTER_FAIL;
        if (T.oq.status == QAG_ESTORE) T.status |= ST_STORE_FULL;
    }
    if (rim_isnan(contrib)) { hey_fail(T); return; }

    if (T.stage == HS_NR_FIRST) {
        T.nr_val = contrib;
        T.stage = HS_NR_RIGHT;
        T.steps = 0;
        hey_loop_head(pt, T);
        return;
    }
    bool keep_going = true;
    if (T.stage == HS_NR_RIGHT) {
        if (T.nr_val != 0.) keep_going = rim_fabs(contrib / T.nr_val) > RIM_HEY_TOL;
        T.nr_val += contrib;
        T.pomega_right += T.delta_right;
        if (!keep_going) { T.stage = HS_NR_LEFT; T.steps = 0; }
        hey_loop_head(pt, T);
    } else if (T.stage == HS_NR_LEFT) {
        keep_going = rim_fabs(contrib / T.nr_val) > RIM_HEY_TOL;
        T.nr_val += contrib;
        T.pomega_left -= T.delta_left;
        if (!keep_going) {
            T.stage = HS_QR;
            T.steps = 0;
            T.qr_val = 0.;
            T.sigma_low = rust_max(pt.sigma0, RIM_INVERSE_SQRT_3 * rim_pow(pt.sigma0, 1.5));
            T.delta_sigma = pt.sigma0;
        }
        hey_loop_head(pt, T);
    } else {   // HS_QR
        if (T.qr_val != 0.) keep_going = rim_fabs(contrib / T.qr_val) > RIM_HEY_TOL;
        T.qr_val += contrib;
        T.sigma_low += T.delta_sigma;
        if (!keep_going) {
            const double ssin = pt.s * pt.sin_th;
            T.result = 2. * (RIM_ELECTRON_CHARGE * RIM_ELECTRON_CHARGE) * (T.nr_val + T.qr_val)
                / (RIM_MASS_ELECTRON * (ssin * ssin));
            T.stage = HS_DONE;
        } else {
            hey_loop_head(pt, T);
        }
    }
}

__device__ __forceinline__ double hey_result(const HeyTask &T, int &status_out)
{
    status_out = T.status;
    if (T.failed) {
        status_out |= ST_NONFINITE;
        return RIM_NAN;
    }
    if (rim_isnan(T.result)) status_out |= ST_NONFINITE;
    return T.result;
}

// The whole coefficient on one wave (no cooperation): the readable statement of the control flow.
template <int KIND>
__device__ __forceinline__ double heyvaerts_coefficient(const HeyPoint &pt, const DistParams &dist, const HeyConsts &hc,
                                                        const GKLane &g, const IStore &inner, const IStore &outer,
                                                        HeyTask *park, QagParkBase *qpark, int &status_out)
{
    const int lane = g.lane;
    HeyTask T;
    hey_begin(pt, T);
    while (T.stage != HS_DONE) {
        SymBatch B;
        if (!hey_post(pt, g, outer, T, B)) break;
        if (lane == 0) {
            *park = T;
            if (qpark->hb) {
                hb_store(qpark->hb + 1, (unsigned long long) T.batches);
                hb_store(qpark->hb + 2, (unsigned long long) (100 * T.stage + T.phase));
                hb_store(qpark->hb + 5, (unsigned long long) T.steps);
            }
        }
        int batch_status = 0;
        double gval = 0.;
        unsigned long long mask = wv_ballot(B.req_active);
        while (mask) {
            const int k = __builtin_ffsll((long long) mask) - 1;
            mask &= mask - 1;
            const double u = readlane_d(B.req_n, k);
            int k2 = -1;
            double u2 = u;
            if (mask) {
                k2 = __builtin_ffsll((long long) mask) - 1;
                mask &= mask - 1;
                u2 = readlane_d(B.req_n, k2);
            }
            double val, val2;
            int st = 0, st2 = 0;
            hey_eval_pair<KIND>(pt, dist, hc, g, inner, qpark, u, B.req_lobe, u2, B.req_lobe, k2 >= 0, val, st, val2, st2);
            batch_status |= st | st2;
            if (lane == k) gval = val;
            if (lane == k2) gval = val2;
        }
        wv_sync();
        T = *park;
        hey_uniformize(T);
        hey_consume(pt, g, outer, T, B, gval, uni(batch_status));
    }
    return hey_result(T, status_out);
}

}  // namespace rim
#endif
