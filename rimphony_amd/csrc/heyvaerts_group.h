// heyvaerts_group.h -- rho_Q and rho_V of ONE parameter point advancing in lock-step on one wavefront.
//
// The reference computes the two Faraday coefficients by two independent calls (lib.rs:178-191 ->
// heyvaerts.rs:60-78), each a double integral over the same (sigma, pomega) domain whose inner integrand is one of the
// elements h_* (rho_Q) or f_* (rho_V) of heyvaerts.rs:302-468.  The elements of one sample share almost everything: the
// coordinates (fill_coord_vars, :194-201), d f / d sigma with the distribution's power and exponential (:472-493), and
// the Bessel functions -- I_{+-1/3}, I_{+-2/3}(g), or J and Y of orders sigma, sigma - 1 -- that are 60 % of the kernel's
// time; they differ in a closing formula of a dozen operations (dev_heyvaerts.h: hey_shared / hey_member).
//
// Same machinery as the Symphony groups (symphony_group.h): the two members' state machines (heyvaerts_wave.h:
// hey_post / hey_consume, unchanged) run side by side, identical outer abscissae merge into one entry, and an entry's
// inner quadratures run in lock-step (wave_qag_group): shared first rule application, each member bisecting what ITS
// error list says, one pass serving both where they agree, rule sums stashed for the other where they do not.  Every
// member sees the arithmetic of a solo run; values, status words and work counters are the reference's, bit for bit.
#ifndef RIM_HEYVAERTS_GROUP_H
#define RIM_HEYVAERTS_GROUP_H

#include "heyvaerts_wave.h"
#include "symphony_group.h"

namespace rim {

// Output slots of the members: 6 = rho_Q (Stokes Q), 7 = rho_V (Stokes V); group_slot() keeps 3 bits, which is enough.
__device__ __forceinline__ int hey_slot_stokes(int slot) { return slot == 6 ? STOKES_Q : STOKES_V; }

// the elements of the members at one sample: the shared part once (dev_heyvaerts.h)
template <int KIND>
struct HeyGroupF {
    const HeyPoint *pt;             // wave-uniform (pt->stokes is not used)
    const DistParams *d;
    const HeyConsts *hc;
    double u0, u1;
    bool q0, q1;
    unsigned slots;
    bool qr;                        // of the current pass
    HeyShared sh;

    __device__ __forceinline__ void shared(double v, bool active, bool second, unsigned members)
    {
        bool want_h = false;
        for (unsigned rem = members; rem; rem &= rem - 1)
            if (group_slot(slots, __builtin_ctz(rem)) == 6) want_h = true;
        qr = second ? q1 : q0;
        const double fixed = second ? u1 : u0;
        sh = HeyShared();
        if (active) sh = hey_shared<KIND>(*pt, *d, *hc, qr, fixed, v, want_h);
    }
    __device__ __forceinline__ double member(int m, bool active) const
    {
        if (!active) return 0.;
        return hey_member(*pt, hey_slot_stokes(group_slot(slots, m)), qr, sh);
    }
};

// One or two entries: the inner integral at outer abscissa u (heyvaerts.rs:213-250 non-resonant, 262-296
// quasi-resonant) for the members of mask0 (and of mask1 at (u1, qr1) if mask1 != 0).  THE integrand site of the Faraday
// group kernel.  An entry whose sigma range is empty is 0 without a quadrature (heyvaerts.rs:224-226) and the other
// one then runs alone.  Results: gp->mem[m].res[e] / .qst[e] / .samples.
template <int KIND>
__device__ __forceinline__ void hey_eval_group(const HeyPoint &pt, const DistParams &dist, const HeyConsts &hc, unsigned slots,
                                               const GKLane &g, double *inner_lds, double *inner_spill, GroupParkBase *gp,
                                               double u0, int qr0, unsigned mask0, double u1, int qr1, unsigned mask1)
{
    const int lane = g.lane;
    const bool have1 = mask1 != 0;
    // every lane sets up the entry of its half
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

    HeyGroupF<KIND> f;
    f.pt = &pt; f.d = &dist; f.hc = &hc;
    f.slots = slots;
    f.qr = false;
    wv_sync();                       // nobody still reads the previous evaluation's member records
    if (todo0 || todo1) {
        // both: one joint run.  Only one of them: it runs as integral A (its own data in the functor's first place)
        const bool flip = !todo0;
        f.u0 = uni(flip ? u1 : u0); f.u1 = uni(u1);
        f.q0 = (flip ? uni(qr1) : uni(qr0)) != 0; f.q1 = uni(qr1) != 0;
        wave_qag_group(f, g, inner_lds, inner_spill, flip ? a1 : a0, flip ? b1 : b0, a1, b1,
                       flip ? mask1 : mask0, (todo0 && todo1) ? mask1 : 0u, 1e-3, 4096, gp);
        if (flip && lane == 0) {
            for (unsigned rem = mask1; rem; rem &= rem - 1) {
                const int m = __builtin_ctz(rem);
                gp->mem[m].res[1] = gp->mem[m].res[0];
                gp->mem[m].qst[1] = gp->mem[m].qst[0];
            }
        }
    }
    // entries with an empty range: 0, no samples
    if (lane == 0) {
        if (!todo0)
            for (unsigned rem = mask0; rem; rem &= rem - 1) {
                const int m = __builtin_ctz(rem);
                gp->mem[m].res[0] = 0.; gp->mem[m].qst[0] = QAG_SUCCESS;
                if (!(todo1 && ((mask1 >> m) & 1u))) gp->mem[m].samples = 0u;
            }
        if (have1 && !todo1)
            for (unsigned rem = mask1; rem; rem &= rem - 1) {
                const int m = __builtin_ctz(rem);
                gp->mem[m].res[1] = 0.; gp->mem[m].qst[1] = QAG_SUCCESS;
                if (!(todo0 && ((mask0 >> m) & 1u))) gp->mem[m].samples = 0u;
            }
    }
    wv_sync();
}

// Whose turn is it (see symphony_group.h: group_turn)?  Behind = earlier stage (first non-resonant chunk, marching
// right, marching left, quasi-resonant), then the smaller distance marched, then the earlier phase of the chunk.
__device__ __forceinline__ unsigned hey_group_turn(const HeyTask *park, unsigned alive)
{
    unsigned turn = 0;
    int best_stage = 0, best_phase = 0;
    double best_pos = 0.;
    for (unsigned rem = alive; rem; rem &= rem - 1) {
        const int m = __builtin_ctz(rem);
        const int stage = uni(park[m].stage), phase = uni(park[m].phase);
        const double pos = stage == HS_NR_RIGHT ? uni(park[m].pomega_right) : stage == HS_NR_LEFT ? -uni(park[m].pomega_left)
                         : stage == HS_QR ? uni(park[m].sigma_low) : 0.;
        int cmp = 0;
        if (!turn) cmp = -1;
        else if (stage != best_stage) cmp = stage < best_stage ? -1 : 1;
        else if (pos != best_pos) cmp = pos < best_pos ? -1 : 1;
        else if (phase != best_phase) cmp = phase < best_phase ? -1 : 1;
        if (cmp < 0) { turn = 1u << m; best_stage = stage; best_phase = phase; best_pos = pos; }
        else if (cmp == 0) turn |= 1u << m;
    }
    return turn;
}

// ---- the Faraday pair on one wave, no cooperation between waves (wavefront emulator; the readable statement) --------
template <int KIND>
__device__ __forceinline__ void heyvaerts_group(const HeyPoint &pt0, const DistParams &dist, const HeyConsts &hc, unsigned slots, int nmem,
                                                const GKLane &g, double *inner_lds, double *inner_spill, double *outer_spill,
                                                HeyTask *park, GroupParkBase *gp, double *vals, int *stats)
{
    const int lane = g.lane;
    unsigned alive = 0;
    for (int m = 0; m < nmem; m++) {
        HeyPoint pt = pt0;
        pt.stokes = hey_slot_stokes(group_slot(slots, m));
        HeyTask T;
        hey_begin(pt, T);
        wv_sync();
        if (lane == 0) park[m] = T;
        if (T.stage != HS_DONE) alive |= 1u << m;
    }
    wv_sync();
    while (alive) {
        double rq0 = 0., rq1 = 0., rq2 = 0., rq3 = 0.;
        int rl0 = 0, rl1 = 0, rl2 = 0, rl3 = 0;
        unsigned act = 0, posted = 0;
        int nq0 = 0, nq1 = 0, nq2 = 0, nq3 = 0;
        int ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0;
        const unsigned turn = hey_group_turn(park, alive);
        for (unsigned rem = turn; rem; rem &= rem - 1) {
            const int m = __builtin_ctz(rem);
            HeyPoint pt = pt0;
            pt.stokes = hey_slot_stokes(group_slot(slots, m));
            const IStore outer = group_store(nullptr, 0, outer_spill, SPILL_HEYGOUTER, m);
            HeyTask T = park[m];
            hey_uniformize(T);
            SymBatch B;
            const bool ok = hey_post(pt, g, outer, T, B);
            wv_sync();
            if (lane == 0) park[m] = T;
            if (!ok) { alive &= ~(1u << m); continue; }
            posted |= 1u << m;
            put4(m, B.req_n, rq0, rq1, rq2, rq3);
            put4(m, B.req_lobe, rl0, rl1, rl2, rl3);
            put4(m, B.n_req, nq0, nq1, nq2, nq3);
            put4(m, B.phase, ph0, ph1, ph2, ph3);
            if (B.req_active) act |= 1u << m;
        }
        wv_sync();
        const GroupClasses gc = group_classify(act, rq0, rq1, rq2, rq3, rl0, rl1, rl2, rl3);
        double gv0 = 0., gv1 = 0., gv2 = 0., gv3 = 0.;
        int bs0 = 0, bs1 = 0, bs2 = 0, bs3 = 0;
        unsigned long long lanes = wv_ballot(gc.reps != 0);
        int cur_k = -1;
        unsigned cur_bits = 0;
        for (;;) {
            int ek0 = -1, ek1 = -1, em0 = 0, em1 = 0;
            for (int e = 0; e < 2; e++) {
                if (cur_bits == 0) {
                    if (!lanes) break;
                    cur_k = __builtin_ffsll((long long) lanes) - 1;
                    lanes &= lanes - 1;
                    cur_bits = (unsigned) wv_readlane((int) gc.reps, cur_k);
                }
                if (e == 0) { ek0 = cur_k; em0 = __builtin_ctz(cur_bits); }
                else { ek1 = cur_k; em1 = __builtin_ctz(cur_bits); }
                cur_bits &= cur_bits - 1;
            }
            if (ek0 < 0) break;
            const double n0 = readlane_d(sel4(em0, rq0, rq1, rq2, rq3), ek0);
            const int lb0 = wv_readlane(sel4(em0, rl0, rl1, rl2, rl3), ek0);
            const unsigned mk0 = ((unsigned) wv_readlane((int) gc.cmask, ek0) >> (4 * em0)) & 15u;
            double n1 = n0;
            int lb1 = lb0;
            unsigned mk1 = 0;
            if (ek1 >= 0) {
                n1 = readlane_d(sel4(em1, rq0, rq1, rq2, rq3), ek1);
                lb1 = wv_readlane(sel4(em1, rl0, rl1, rl2, rl3), ek1);
                mk1 = ((unsigned) wv_readlane((int) gc.cmask, ek1) >> (4 * em1)) & 15u;
            }
            hey_eval_group<KIND>(pt0, dist, hc, slots, g, inner_lds, inner_spill, gp, n0, lb0, mk0, n1, lb1, mk1);
            for (int e = 0; e < 2; e++) {
                const unsigned mk = e ? mk1 : mk0;
                const int ek = e ? ek1 : ek0;
                for (unsigned rem = mk; rem; rem &= rem - 1) {
                    const int m = __builtin_ctz(rem);
                    int st = 0;
                    const double v = group_entry_value(gp, m, e, st);
                    put4(m, sel4(m, bs0, bs1, bs2, bs3) | st, bs0, bs1, bs2, bs3);
                    if (lane == ek) put4(m, v, gv0, gv1, gv2, gv3);
                }
            }
        }
        wv_sync();
        for (unsigned rem = posted; rem; rem &= rem - 1) {
            const int m = __builtin_ctz(rem);
            HeyPoint pt = pt0;
            pt.stokes = hey_slot_stokes(group_slot(slots, m));
            const IStore outer = group_store(nullptr, 0, outer_spill, SPILL_HEYGOUTER, m);
            HeyTask T = park[m];
            hey_uniformize(T);
            SymBatch B;
            B.req_n = sel4(m, rq0, rq1, rq2, rq3);
            B.req_lobe = sel4(m, rl0, rl1, rl2, rl3);
            B.req_active = ((act >> m) & 1u) != 0;
            B.n_req = sel4(m, nq0, nq1, nq2, nq3);
            B.phase = sel4(m, ph0, ph1, ph2, ph3);
            hey_consume(pt, g, outer, T, B, sel4(m, gv0, gv1, gv2, gv3), uni(sel4(m, bs0, bs1, bs2, bs3)));
            wv_sync();
            if (lane == 0) park[m] = T;
            if (T.stage == HS_DONE) alive &= ~(1u << m);
        }
        wv_sync();
    }
    for (int m = 0; m < nmem; m++) {
        HeyTask T = park[m];
        hey_uniformize(T);
        int st = 0;
        vals[m] = hey_result(T, st);
        stats[m] = st;
    }
}

}  // namespace rim
#endif
