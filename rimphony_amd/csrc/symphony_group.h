// symphony_group.h -- the Symphony coefficients of ONE parameter point advancing in lock-step on one wavefront.
//
// The reference evaluates the eight coefficients of a point by eight independent calls (lib.rs:178-191), each with its
// own CalculationState (symphony.rs:66-187).  But J_n(z) and J'_n(z) of an integrand sample -- more than half of its
// cost -- depend on (s, theta, n, gamma) only (symphony.rs:406-442); j_I, alpha_I, j_Q, alpha_Q integrate over the same
// gamma-ranges of the same harmonics, start every gamma-integral from the same rule application and mostly bisect the
// same intervals; j_V and alpha_V do the same on the two lobes.  Measured with the oracle (profiles/r2_sample_overlap.txt):
// the six coefficients evaluate 3.6 x as many samples as there are distinct (n, gamma) pairs.
//
// Here a task is (point, GROUP): group IQ = {j_I, alpha_I, j_Q, alpha_Q} and group V = {j_V, alpha_V} (whatever the
// coefficient mask selects of them).  One wave runs the members' state machines (symphony_wave.h: sym_post / sym_consume,
// unchanged) side by side:
//   * every round each unfinished member posts its batch of (n, lobe) requests; requests that are identical between
//     members (same bits of n, same lobe) are merged -- lane by lane, a lane-local comparison -- into one ENTRY
//     (n, lobe, member mask);
//   * an entry is ONE gamma-integral evaluated for all its members together (wave_qag_group): the members' adaptive
//     quadratures run in lock-step.  They share the first rule application; afterwards every member picks the interval
//     ITS OWN error list says, the members that picked the same interval are served by one pass -- kinematics and the
//     Bessel pair once per sample, the polarisation and distribution terms per member --, and a member that picked
//     another interval gets a pass of its own.  No table, no anticipation: a sample is evaluated when some member's
//     quadrature asks for it, and every member sees exactly the arithmetic of a solo run (its own rule sums from its
//     own integrand values in the same reduction tree, its own bisection sequence, its own round-off counters), so
//     results, status words and the per-coefficient work counters are the reference's, bit for bit.
//
// Nothing here changes a rounding: tests/test_gpu_parity.py compares group runs with the oracle's per-coefficient
// values and with solo runs (coefficient masks that leave a member alone).
#ifndef RIM_SYMPHONY_GROUP_H
#define RIM_SYMPHONY_GROUP_H

#include "symphony_wave.h"

namespace rim {

#define RIM_GROUP 4
// LDS entries of each member's subinterval lists; longer lists continue in the wave's spill region (global memory),
// as in wave_qag.h.  Lengths on the bench tables: a gamma-integral takes 6.9 passes on average.
#ifndef CAP_GINNER
#define CAP_GINNER 24
#endif
#ifndef CAP_GOUTER
#define CAP_GOUTER 16
#endif
// per-member global spill behind the LDS lists: the GSL limits of the path (symphony.rs:375-380: 5000; :264-269: 1000)
// (the Faraday group -- heyvaerts_group.h -- uses the same stores: its limits are 4096 / 4096, heyvaerts.rs:82-83)
#define SPILL_GINNER 5000
#define SPILL_GOUTER 1000               // symphony.rs:264-269: the n-chunk quadrature's limit
#define SPILL_HEYGOUTER 4096            // heyvaerts.rs:82-83 (the Faraday pair in lock-step, heyvaerts_group.h)
#define SPILL_GROUP_DOUBLES_PER_WAVE (RIM_GROUP * (RIM_ISTORE_DOUBLES(SPILL_GINNER) + RIM_ISTORE_DOUBLES(SPILL_GOUTER)))
#define SPILL_HEYGROUP_DOUBLES_PER_WAVE (RIM_GROUP * (RIM_ISTORE_DOUBLES(SPILL_GINNER) + RIM_ISTORE_DOUBLES(SPILL_HEYGOUTER)))

// A member is identified by its output slot 0..5 (lib.rs:176-177): coefficient = slot & 1 (emission, absorption),
// Stokes parameter = slot >> 1.  `slots` packs the members' slots, 4 bits each.
__device__ __forceinline__ int group_slot(unsigned slots, int m) { return (int) ((slots >> (4 * m)) & 7u); }

// uniform selection among four per-lane values (m is wave-uniform): the per-member per-lane data of the group driver
// lives in named registers, and this is how a loop over members reaches it
__device__ __forceinline__ double sel4(int m, double a0, double a1, double a2, double a3)
{ return m == 0 ? a0 : m == 1 ? a1 : m == 2 ? a2 : a3; }
__device__ __forceinline__ int sel4(int m, int a0, int a1, int a2, int a3)
{ return m == 0 ? a0 : m == 1 ? a1 : m == 2 ? a2 : a3; }
__device__ __forceinline__ void put4(int m, double v, double &a0, double &a1, double &a2, double &a3)
{ if (m == 0) a0 = v; else if (m == 1) a1 = v; else if (m == 2) a2 = v; else a3 = v; }
__device__ __forceinline__ void put4(int m, int v, int &a0, int &a1, int &a2, int &a3)
{ if (m == 0) a0 = v; else if (m == 1) a1 = v; else if (m == 2) a2 = v; else a3 = v; }

// ---- per-member state of the inner (gamma) quadratures, wave-private LDS ------------------------------------
// The STASH: when a pass bisects interval X because some member's quadrature picked it, every other member of the
// integral whose list holds X as an entry (it will most likely bisect X too, a few steps later: the members' trees of
// intervals are nearly the same, the order in which they grow them is not) gets ITS OWN rule sums of the two children
// computed from the samples that are in the registers anyway (its polarisation and distribution terms: a fifth of a
// pass) and filed here under the entry's index.  When that member's quadrature picks the entry, the sums are taken
// from the stash and no pass is executed.  The sums are the ones a pass of the member alone would produce (same
// samples, same reduction tree), a stashed entry that is never picked costs nothing but the fifth of a pass, and
// samples are counted for a member when its quadrature consumes them: values, status words and counters do not
// depend on whether, or when, anything was stashed.
#define RIM_STASH 8
struct GroupMember {
    double area, errsum;        // running sums of the member's QAG (qag.c)
    double fb[2][4];            // first-rule sums (result, abserr, resabs, resasc) of integrals A and B
    double res[2];              // results of the member's integrals A and B
    int qst[2];                 // their GSL status
    int iteration, size, rt1, rt2, imax;
    unsigned samples;           // integrand samples of this member during the current evaluation
    // children's rule sums filed ahead of the member's pick: entry index (-1: free), then per child result / abserr,
    // flags bit 0 / 1: resasc != abserr of child 1 / 2
    int sk[RIM_STASH], sf[RIM_STASH];
    double sr1[RIM_STASH], se1[RIM_STASH], sr2[RIM_STASH], se2[RIM_STASH];
    int snext;                  // replacement cursor
};

struct GroupParkBase {
    GroupMember mem[RIM_GROUP];
    WaveCounters ctr;           // samples / inner_qags: summed over members (the reference's counts); steps: passes executed
    unsigned long long member_passes;   // passes a solo run of every member would have executed (sharing = member_passes / steps)
    unsigned long long stash_filed, stash_used;
    unsigned long long *hb;
};
struct GroupPark : GroupParkBase {
    LeungOrder ord[4];          // Symphony: order records (n, n + 1) of the one or two gamma-integrals in flight
};

__device__ __forceinline__ IStore group_store(double *lds_base, int cap, double *spill_base, int gcap, int m)
{
    return istore_carve(lds_base ? lds_base + (size_t) m * RIM_ISTORE_DOUBLES(cap) : nullptr, lds_base ? cap : 0,
                        spill_base ? spill_base + (size_t) m * RIM_ISTORE_DOUBLES(gcap) : nullptr, spill_base ? gcap : 0);
}

// position of entry idx in member M's stash, or -1
__device__ __forceinline__ int stash_find(const GroupMember *M, int idx, int lane)
{
    const unsigned long long hit = wv_ballot(lane < RIM_STASH && M->sk[lane & (RIM_STASH - 1)] == idx);
    return hit ? __builtin_ffsll((long long) hit) - 1 : -1;
}

// ---- the adaptive quadratures of up to RIM_GROUP members over one or two intervals, in lock-step ---------------

// qag.c's loop body for member M after the two children of its picked entry `imax` = [a_i, b_i] have been evaluated:
// area / error of child 1 and 2 in lane 0's view (every lane executes, decisions are lane 0's), ne_both = "resasc !=
// error for both children", r_own / e_own = the sums of THIS lane's child (lanes 0 and 32 file their own child).
// Returns true when the member's integral is finished (result and status filed in M->res[cur] / M->qst[cur]).
__device__ __forceinline__ bool group_book(GroupMember *M, const IStore &st, const GKLane &g, int cur, int imax,
                                           double a_i, double b_i, double area1, double error1, double area2, double error2,
                                           bool ne_both, double r_own, double e_own, double epsrel, int limit)
{
    const int lane = g.lane;
    int iteration = uni(M->iteration), size = uni(M->size), rt1 = uni(M->rt1), rt2 = uni(M->rt2);
    const IEntry en = ist_entry(st, imax);
    const double r_i = en.r, e_i = en.e;
    const double mid = 0.5 * (a_i + b_i);
    const double area12 = area1 + area2;
    const double error12 = error1 + error2;
    const double errsum = M->errsum + (error12 - e_i);
    const double area = M->area + (area12 - r_i);
    if (ne_both) {
        const double delta = r_i - area12;
        if (lane0_and(rim_fabs(delta) <= 1.0e-5 * rim_fabs(area12), error12 >= 0.99 * e_i)) rt1++;
        if (iteration >= 10 && lane0(error12 > e_i)) rt2++;
    }
    const double tolerance = epsrel * rim_fabs(area);          // = rim_max(0, x) for x >= 0 or NaN: epsabs is 0 on this path
    const bool more = lane0(errsum > tolerance);
    int error_type = 0;
    if (more) {
        if (rt1 >= 6 || rt2 >= 20) error_type = 2;
        const double tmp = (1 + 100 * RIM_DBL_EPSILON) * (rim_fabs(mid) + 1000 * RIM_DBL_MIN);
        if (lane0_and(rim_fabs(a_i) <= tmp, rim_fabs(b_i) <= tmp)) error_type = 3;
    }
    // update(): the child with the larger error keeps the parent's slot; lanes 0 and 32 file their own child
    const bool c2gt = lane0(error2 > error1);
    const bool overflow = size >= istore_capacity(st);
    wv_sync();               // every lane has read the parent's entry and the member's sums
    if (!overflow) {
        if ((lane & 31) == 0) {
            const bool keep = (g.half != 0) == c2gt;
            const int slot = keep ? imax : size;
            const int stamp = 2 * iteration + (keep ? 0 : 1);
            const double la = g.half ? mid : a_i, lb = g.half ? b_i : mid;
            if (slot < st.cap) {
                st.a[slot] = la; st.b[slot] = lb; st.r[slot] = r_own; st.e[slot] = e_own;
                st.stamp[slot] = stamp;
            } else {         // beyond the LDS part: the wave's spill region in global memory
                const int j = slot - st.cap;
                st.g[j] = la; st.g[st.gcap + j] = lb; st.g[2 * st.gcap + j] = r_own; st.g[3 * st.gcap + j] = e_own;
                ((int *) (st.g + 4 * st.gcap))[j] = stamp;
            }
        }
        size++;
    }
    iteration++;
    const bool finished = !(!overflow && iteration < limit && !error_type && more);
    if (lane == 0) {
        M->area = area; M->errsum = errsum;
        M->iteration = iteration; M->size = size; M->rt1 = rt1; M->rt2 = rt2;
        M->samples += 62u;
    }
    if (finished) {
        wv_sync();
        double sum = 0;
        if (size <= st.cap) { for (int k = 0; k < size; k++) sum += st.r[k]; }
        else { for (int k = 0; k < size; k++) sum += ist_r(st, k); }
        int status;
        if (lane0(errsum <= tolerance)) status = QAG_SUCCESS;      // (not "!more": a NaN error sum is a failure)
        else if (overflow) status = QAG_ESTORE;
        else if (error_type == 2) status = QAG_EROUND;
        else if (error_type == 3) status = QAG_ESING;
        else if (iteration == limit) status = QAG_EMAXITER;
        else status = QAG_EFAILED;
        if (lane == 0) { M->res[cur] = sum; M->qst[cur] = status; }
        RIM_HIT(22);
    }
    return finished;
}

// Integral A on [a0, b0] for the members of maskA, integral B on [a1, b1] for those of maskB (maskB == 0: none); as in
// wave_qag_pair the two share their first rule application (A on lanes 0..30, B on 32..62), then A's members run to
// the end, then B's.  The integrand is a functor with two steps:
//   f.shared(x, active, second, members)   everything of the sample that does not depend on the member (per lane:
//                                          x and `active`; `second` selects integral B's order data; `members` says whose
//                                          terms will be asked for)
//   f.member(m, active) -> double          member m's integrand value at the lane's sample (0 on inactive lanes)
// Each member's arithmetic is that of wave_qag_pair / qag.c: results in gp->mem[m].res[] / .qst[].  One loop, one
// site of the integrand: a turn of the loop is the joint first rule application (cur < 0), a pass over the two
// children of an interval, or the booking of members whose pick was on file in their stash.
template <class F>
__device__ __forceinline__ void wave_qag_group(F &f, const GKLane &g, double *inner_lds, double *inner_spill,
                                               double a0, double b0, double a1, double b1, unsigned maskA, unsigned maskB,
                                               double epsrel, int limit, GroupParkBase *gp)
{
    const int lane = g.lane;
    // The masks, limits and tolerances are wave-uniform by construction; saying so keeps the member loops below -- their
    // masks, member indices, list addresses and counters -- in scalar registers (the compiler cannot see it through the
    // board path of the caller and otherwise runs all of it on the vector unit, spilling to scratch).
    maskA = uni(maskA); maskB = uni(maskB);
    epsrel = uni(epsrel); limit = uni(limit);
    a0 = uni(a0); b0 = uni(b0); a1 = uni(a1); b1 = uni(b1);
    const bool have1 = maskB != 0;
    unsigned n_pass = 0, n_member_pass = 0, n_samp_all = 0, n_qags = 0, n_filed = 0, n_used = 0;
    int cur = -1;                       // -1: the joint first rule application; then integral 0 (A) and 1 (B)
    unsigned active = 0, need_pick = 0;
    unsigned stashed = 0;               // bit m: member m's stash may hold entries (else no look-up is needed)

    wv_sync();                          // nobody still reads the previous evaluation's member records
    for (;;) {
        unsigned serve = 0;             // members whose rule sums this turn's pass produces
        unsigned booknow = 0;           // ... of them, those that picked the pass's interval: booked from the registers
        unsigned from_stash = 0;        // members whose pick is on file: booked without a pass
        // entry of the pass's interval in each served member's list: four 16-bit fields of one wave-uniform word (list
        // indices stay below the GSL limits, 5000 / 4096) -- as four ints indexed by the member they lived in a scratch
        // array and came back as vector registers
        unsigned long long idxp = 0;
        double a_i = 0., b_i = 0.;      // the pass's interval (uniform)
        double la = 0., lb = 0.;        // per lane: the interval this lane's sample belongs to
        bool active_lane = false, second = false;

        if (cur < 0) {
            second = g.half != 0 && have1;
            la = second ? a1 : a0; lb = second ? b1 : b0;
            serve = maskA | maskB;
            active_lane = g.node && (g.half == 0 || have1);
            RIM_HIT(1);
        } else {
            // every member whose list changed picks its next interval; a pick whose children's sums are on file in the
            // member's stash is booked at once
            wv_sync();                  // list entries and member records written in the last turn are visible
            RIM_PROF_T(t_picks);
            for (unsigned rem = need_pick; rem; rem &= rem - 1) {
                const int m = __builtin_ctz(rem);
                GroupMember *const M = gp->mem + m;
                const int size = uni(M->size);
                int imax = 0;           // qpsrt: slot 0 while the list has <= 2 entries
                if (size > 2) {
                    const IStore st = group_store(inner_lds, CAP_GINNER, inner_spill, SPILL_GINNER, m);
                    RIM_PROF_T(t_pick);
                    imax = (size <= 64 && size <= st.cap) ? lean_pick(st, size, lane) : spill_pick(st, size, lane);
                    RIM_PROF_ADD(13, t_pick);
                }
                if (lane == 0) M->imax = imax;
                if (((stashed >> m) & 1u) && stash_find(M, imax, lane) >= 0) from_stash |= 1u << m;
            }
            if (need_pick) wv_sync();
            need_pick = 0;
            RIM_PROF_ADD(18, t_picks);
            RIM_PROF_T(t_prep);
            if (!from_stash) {
                // a pass: the interval of the first active member; it serves the members that picked this interval and
                // the members that hold it as an entry of their list (their sums go to the stash)
                const int m0 = __builtin_ctz(active);
                {
                    const IStore st = group_store(inner_lds, CAP_GINNER, inner_spill, SPILL_GINNER, m0);
                    const int imax = uni(gp->mem[m0].imax);
                    if (imax < st.cap) { a_i = st.a[imax]; b_i = st.b[imax]; }
                    else { a_i = st.g[imax - st.cap]; b_i = st.g[st.gcap + (imax - st.cap)]; }
                    a_i = uni(a_i); b_i = uni(b_i);
                    idxp = (unsigned long long) (unsigned) imax * 0x0001000100010001ull;
                }
                const unsigned long long abits = rim_bits(a_i), bbits = rim_bits(b_i);
                serve = booknow = 1u << m0;
                for (unsigned rem = active & ~serve; rem; rem &= rem - 1) {
                    const int m = __builtin_ctz(rem);
                    GroupMember *const M = gp->mem + m;
                    const IStore st = group_store(inner_lds, CAP_GINNER, inner_spill, SPILL_GINNER, m);
                    const int imax = uni(M->imax), size = uni(M->size);
                    double am, bm;
                    if (imax < st.cap) { am = st.a[imax]; bm = st.b[imax]; }
                    else { am = st.g[imax - st.cap]; bm = st.g[st.gcap + (imax - st.cap)]; }
                    int idx = -1;
                    if (rim_bits(uni(am)) == abits && rim_bits(uni(bm)) == bbits) { idx = imax; booknow |= 1u << m; }
                    else if (size <= 64 && size <= st.cap) {
                        const int l = lane < size ? lane : 0;
                        const unsigned long long hit = wv_ballot(lane < size && rim_bits(st.a[l]) == abits && rim_bits(st.b[l]) == bbits);
                        if (hit) {
                            idx = __builtin_ffsll((long long) hit) - 1;
                            if (((stashed >> m) & 1u) && stash_find(M, idx, lane) >= 0) idx = -1;       // already on file
                        }
                    }
                    if (idx >= 0) {
                        serve |= 1u << m;
                        idxp = (idxp & ~(0xffffull << (16 * m))) | ((unsigned long long) (unsigned) idx << (16 * m));
                    }
                }
                const double mid = 0.5 * (a_i + b_i);
                la = g.half ? mid : a_i;
                lb = g.half ? b_i : mid;
                active_lane = g.node;
                second = cur != 0;
            }
            RIM_PROF_ADD(19, t_prep);
        }

        if (serve) {
            // ---- one pass: the integrand's shared part once, then every served member's terms and rule sums ----
            const double hl = 0.5 * (lb - la);
            const double x = 0.5 * (la + lb) + hl * gk_t(g);
            RIM_HIT(0);
            RIM_PROF_T(t_int);
            f.shared(x, active_lane, second, serve);
            RIM_PROF_ADD(1, t_int);
            n_pass += 1;
#if defined(RIM_WAVE_EMU)
            if (lane == 0) { extern unsigned long long g_emu_hist[64]; g_emu_hist[(cur < 0 ? 0 : 8) + __builtin_popcount(cur < 0 ? serve : booknow)]++;
                             if (cur >= 0) g_emu_hist[16 + __builtin_popcount(active)]++; }
#endif
            for (unsigned rem = serve; rem; rem &= rem - 1) {
                const int m = __builtin_ctz(rem);
                GroupMember *const M = gp->mem + m;
                const bool inA = ((maskA >> m) & 1u) != 0, inB = ((maskB >> m) & 1u) != 0;
                const bool act_m = cur < 0 ? (g.node && (g.half == 0 ? inA : inB)) : g.node;
                RIM_PROF_T(t_mem);
                const double fv = f.member(m, act_m);
                RIM_PROF_ADD(20, t_mem);
                RIM_PROF_T(t_gk);
                const GKRes r = wave_gk31(fv, hl, g);
                RIM_PROF_ADD(11, t_gk);
                if (cur < 0) {
                    // the joint first application: file the first-rule sums of the member's integral(s)
                    if ((lane == 0 && inA) || (lane == 32 && inB)) {
                        double *fb = M->fb[g.half];
                        fb[0] = r.result; fb[1] = r.abserr; fb[2] = r.resabs; fb[3] = r.resasc;
                    }
                    const unsigned ns = (inA ? 31u : 0u) + (inB ? 31u : 0u);
                    if (lane == 0) M->samples = ns;
                    n_samp_all += ns;
                    n_member_pass += 1;
                    n_qags += (inA ? 1u : 0u) + (inB ? 1u : 0u);
                } else {
                    const unsigned long long ne = wv_ballot(r.resasc != r.abserr);
                    if ((booknow >> m) & 1u) {
                        RIM_PROF_T(t_ab);
                        const IStore st = group_store(inner_lds, CAP_GINNER, inner_spill, SPILL_GINNER, m);
                        const bool fin = group_book(M, st, g, cur, (int) ((idxp >> (16 * m)) & 0xffffull), a_i, b_i,
                                                    r.result, r.abserr, readlane_d(r.result, 32), readlane_d(r.abserr, 32),
                                                    (ne & 1ull) && ((ne >> 32) & 1ull), r.result, r.abserr, epsrel, limit);
                        n_samp_all += 62u;
                        n_member_pass += 1;
                        if (fin) active &= ~(1u << m); else need_pick |= 1u << m;
                        RIM_PROF_ADD(12, t_ab);
                    } else {
                        // file the children's sums in the member's stash (a free place, else the next one in turn)
                        RIM_PROF_T(t_file);
                        const int idx = (int) ((idxp >> (16 * m)) & 0xffffull);
                        const unsigned long long freep = wv_ballot(lane < RIM_STASH && M->sk[lane & (RIM_STASH - 1)] < 0);
                        const int pos = freep ? __builtin_ffsll((long long) freep) - 1 : uni(M->snext);
                        wv_sync();
                        if (lane == 0) {
                            M->sk[pos] = idx;
                            M->sf[pos] = (int) (ne & 1ull) | (int) (((ne >> 32) & 1ull) << 1);
                            M->sr1[pos] = r.result; M->se1[pos] = r.abserr;
                            if (!freep) M->snext = (pos + 1) & (RIM_STASH - 1);
                        }
                        if (lane == 32) { M->sr2[pos] = r.result; M->se2[pos] = r.abserr; }
                        stashed |= 1u << m;
                        n_filed += 1;
                        RIM_PROF_ADD(21, t_file);
                    }
                }
            }
        }

        if (from_stash) {
            // ---- members whose pick has its children's sums on file: qag.c's loop body without a pass ----
            RIM_PROF_T(t_fs);
            for (unsigned rem = from_stash; rem; rem &= rem - 1) {
                const int m = __builtin_ctz(rem);
                GroupMember *const M = gp->mem + m;
                const IStore st = group_store(inner_lds, CAP_GINNER, inner_spill, SPILL_GINNER, m);
                const int imax = uni(M->imax);
                const int pos = stash_find(M, imax, lane);
                const double area1 = M->sr1[pos], error1 = M->se1[pos], area2 = M->sr2[pos], error2 = M->se2[pos];
                const int sfl = uni(M->sf[pos]);
                const IEntry en = ist_entry(st, imax);
                wv_sync();
                if (lane == 0) M->sk[pos] = -1;
                const bool fin = group_book(M, st, g, cur, imax, uni(en.a), uni(en.b), area1, error1, area2, error2, sfl == 3,
                                            g.half ? area2 : area1, g.half ? error2 : error1, epsrel, limit);
                n_samp_all += 62u;
                n_member_pass += 1;
                n_used += 1;
                if (fin) active &= ~(1u << m); else need_pick |= 1u << m;
            }
            RIM_PROF_ADD(22, t_fs);
        }

        // ---- the next integral, once the current one has no unfinished member ----
        RIM_PROF_T(t_next);
        while (cur < 0 || !active) {
            cur += 1;
            if (cur > (have1 ? 1 : 0)) break;
            const unsigned mask = cur ? maskB : maskA;
            const double fa = cur ? a1 : a0, fbnd = cur ? b1 : b0;
            wv_sync();                          // the first-rule sums are visible to every lane
            stashed = 0;
            for (unsigned rem = mask; rem; rem &= rem - 1) {
                // qag_after_first on the member's first-rule sums
                const int m = __builtin_ctz(rem);
                GroupMember *const M = gp->mem + m;
                const double f_res = M->fb[cur][0], f_err = M->fb[cur][1], f_abs = M->fb[cur][2], f_asc = M->fb[cur][3];
                int status = QAG_SUCCESS;
                bool finished = true;
                if (epsrel < 50 * RIM_DBL_EPSILON || epsrel < 0.5e-28) {
                    if (lane == 0) { M->res[cur] = 0.; M->qst[cur] = QAG_EBADTOL; }
                    continue;
                }
                const double tolerance = epsrel * rim_fabs(f_res);          // = rim_max(0, x): epsabs is 0 on this path
                const double round_off = 50 * RIM_DBL_EPSILON * f_abs;
                if (lane0_and(f_err <= round_off, f_err > tolerance)) status = QAG_EROUND;
                else if (lane0_and(f_err <= tolerance, f_err != f_asc) || lane0(f_err == 0.0)) status = QAG_SUCCESS;
                else if (limit == 1) status = QAG_EMAXITER;
                else finished = false;
                if (finished) {
                    if (lane == 0) { M->res[cur] = f_res; M->qst[cur] = status; }
                } else {
                    const IStore st = group_store(inner_lds, CAP_GINNER, inner_spill, SPILL_GINNER, m);
                    if (lane == 0) {
                        st.a[0] = fa; st.b[0] = fbnd; st.r[0] = f_res; st.e[0] = f_err; st.stamp[0] = 0;
                        M->area = f_res; M->errsum = f_err;
                        M->iteration = 1; M->size = 1; M->rt1 = 0; M->rt2 = 0; M->imax = 0; M->snext = 0;
                    }
                    if (lane < RIM_STASH) M->sk[lane] = -1;
                    active |= 1u << m;
                }
            }
            need_pick = active;
        }
        RIM_PROF_ADD(23, t_next);
        if (cur > (have1 ? 1 : 0)) break;
    }
    if (lane == 0) {
        gp->ctr.samples += n_samp_all;
        gp->ctr.steps += n_pass;
        gp->ctr.inner_qags += n_qags;
        gp->member_passes += n_member_pass;
        gp->stash_filed += n_filed;
        gp->stash_used += n_used;
        if (gp->hb) hb_store(gp->hb + 3, gp->ctr.steps);
    }
    wv_sync();                                  // results and sample counts are visible to every lane
}

// ---- gamma_integrand for the members of a group: the shared part once per sample --------------------------------
template <int KIND>
struct SymGroupF {
    double s, cos_th, sin_th;       // wave-uniform
    const DistParams *d;
    const LeungOrder *ord;          // LDS: records of orders n0, n0 + 1, n1, n1 + 1
    double n0, n1;
    int fl0, fl1;
    unsigned slots;
    // per lane, valid between shared() and the member() calls of a pass
    GiShared sh;
    double g2, f_em, f_ab;

    __device__ __forceinline__ void shared(double x, bool active, bool second, unsigned members)
    {
        SymOrder so;
        const int fl = second ? fl1 : fl0;
        so.n = second ? n1 : n0;
        so.small = (fl & 1) != 0; so.np1_small = (fl & 2) != 0; so.dj_nan = (fl & 4) != 0;
        so.o = ord + (second ? 2 : 0);
        bool need_em = false, need_ab = false;
        for (unsigned rem = members; rem; rem &= rem - 1) {
            if (group_slot(slots, __builtin_ctz(rem)) & 1) need_ab = true; else need_em = true;
        }
        sh.gamma = 0.; sh.beta = 0.; sh.cos_xi = 0.; sh.mj = 0.; sh.njp = 0.;
        g2 = 0.; f_em = 0.; f_ab = 0.;
        if (active) {
            sh = gamma_integrand_shared<0>(s, cos_th, sin_th, so, x);
            RIM_PROF_T(t_f);
            g2 = x * x;
#if !defined(RIM_NO_FTERM_SHARE)           // (A/B knob of tools/build_variant.sh)
            if (need_em && need_ab) gamma_integrand_f_terms<KIND>(*d, cos_th, sh, f_em, f_ab);     // what the two share, once
            else
#endif
            {
                if (need_em) f_em = gamma_integrand_f_term<KIND, 0>(COEFF_EMISSION, *d, cos_th, sh);
                if (need_ab) f_ab = gamma_integrand_f_term<KIND, 0>(COEFF_ABSORPTION, *d, cos_th, sh);
            }
            RIM_PROF_ADD(6, t_f);
        }
    }
    __device__ __forceinline__ double member(int m, bool active) const
    {
        const int slot = group_slot(slots, m);
        if (!active) return 0.;
        const double pol_term = gamma_integrand_pol_term(slot >> 1, sh.mj, sh.njp);
        return g2 * pol_term * ((slot & 1) ? f_ab : f_em);
    }
};

// One or two entries: gamma_integral(n) (symphony.rs:312-389) for the members of mask0 (and of mask1 at (n1, lobe1)
// if mask1 != 0).  THE integrand site of the group kernel.  Results: gp->mem[m].res[e] / .qst[e] / .samples.
// `pt.stokes` only says whether this is the V group (the lobes' limits); pt.coeff is not used.
template <int KIND>
__device__ __forceinline__ void sym_eval_group(const SymPoint &pt, const DistParams &dist, unsigned slots, const GKLane &g,
                                               double *inner_lds, double *inner_spill, GroupPark *gp,
                                               double n0, int lobe0, unsigned mask0, double n1, int lobe1, unsigned mask1)
{
    const int lane = g.lane;
    const bool have1 = mask1 != 0;
    RIM_HIT(21);
    RIM_PROF_T(t_setup);
    // setup as in sym_eval_pair: even lanes the order record of n, odd lanes that of n + 1, of their half's request
    const bool second = g.half != 0 && have1;
    const double n_l = second ? n1 : n0;
    const double n_mine = (lane & 1) ? n_l + 1. : n_l;
    LeungOrder ord_mine = LeungOrder();
    if (!(n_mine < 30.)) ord_mine = leung_order(n_mine);
    const GammaLimits L = gamma_limits(pt, n_l, second ? lobe1 : lobe0);
    wv_sync();                       // nobody is still reading the previous requests' records
    if ((lane & 30) == 0) gp->ord[(lane >> 4) | (lane & 1)] = ord_mine;          // lanes 0, 1, 32, 33 -> records 0, 1, 2, 3
    wv_sync();
    const double a0 = readlane_d(L.g0, 0), b0 = readlane_d(L.g1, 0);
    const double a1 = readlane_d(L.g0, 32), b1 = readlane_d(L.g1, 32);
    const int fl_l = (n_l < 30. ? 1 : 0) | ((n_l + 1.) < 30. ? 2 : 0) | (n_l >= 1e15 ? 4 : 0);
    SymGroupF<KIND> f;
    f.s = pt.s; f.cos_th = pt.cos_th; f.sin_th = pt.sin_th;
    f.d = &dist;
    f.ord = gp->ord;
    f.n0 = uni(n0); f.n1 = uni(n1);
    f.fl0 = wv_readlane(fl_l, 0); f.fl1 = wv_readlane(fl_l, 32);
    f.slots = slots;
    RIM_PROF_ADD(7, t_setup);
    wave_qag_group(f, g, inner_lds, inner_spill, a0, b0, a1, b1, mask0, mask1, 1e-3, 5000, gp);
}

// value and status bits of member m's entry e after sym_eval_group (symphony.rs:375-380: an Err is a NaN sample)
__device__ __forceinline__ double group_entry_value(const GroupParkBase *gp, int m, int e, int &st)
{
    const int qs = uni(gp->mem[m].qst[e]);
    if (qs != QAG_SUCCESS) {
        st |= ST_INNER_FAIL;
        if (qs == QAG_ESTORE) st |= ST_STORE_FULL;
        return RIM_NAN;
    }
    return uni(gp->mem[m].res[e]);
}

// ---- lane-local merging of the members' requests ----------------------------------------------------------------
// Lane k holds request k of every member's batch (rq*/rl*: n and lobe; act bit m: member m posted one on this lane).
// Members with the same request share a class; the class is represented by its lowest member.
//   reps   bit m: member m represents a class on this lane
//   cmask  4 bits per member m: the members of the class m represents (0 if it represents none)
struct GroupClasses { unsigned reps, cmask; };

__device__ __forceinline__ GroupClasses group_classify(unsigned act, double rq0, double rq1, double rq2, double rq3,
                                                       int rl0, int rl1, int rl2, int rl3)
{
    const unsigned long long b0 = rim_bits(rq0), b1 = rim_bits(rq1), b2 = rim_bits(rq2), b3 = rim_bits(rq3);
    const bool a0 = act & 1u, a1 = (act & 2u) != 0, a2 = (act & 4u) != 0, a3 = (act & 8u) != 0;
    // first[m]: the lowest active member with member m's request
    const int f1 = (a1 && a0 && b1 == b0 && rl1 == rl0) ? 0 : 1;
    const int f2 = (a2 && a0 && b2 == b0 && rl2 == rl0) ? 0 : (a2 && a1 && b2 == b1 && rl2 == rl1) ? 1 : 2;
    const int f3 = (a3 && a0 && b3 == b0 && rl3 == rl0) ? 0 : (a3 && a1 && b3 == b1 && rl3 == rl1) ? 1
                 : (a3 && a2 && b3 == b2 && rl3 == rl2) ? 2 : 3;
    unsigned c0 = a0 ? 1u : 0u, c1 = (a1 && f1 == 1) ? 2u : 0u, c2 = (a2 && f2 == 2) ? 4u : 0u, c3 = (a3 && f3 == 3) ? 8u : 0u;
    if (a1 && f1 == 0) c0 |= 2u;
    if (a2) { if (f2 == 0) c0 |= 4u; else if (f2 == 1) c1 |= 4u; }
    if (a3) { if (f3 == 0) c0 |= 8u; else if (f3 == 1) c1 |= 8u; else if (f3 == 2) c2 |= 8u; }
    GroupClasses gc;
    gc.reps = (c0 ? 1u : 0u) | (c1 ? 2u : 0u) | (c2 ? 4u : 0u) | (c3 ? 8u : 0u);
    gc.cmask = c0 | (c1 << 4) | (c2 << 8) | (c3 << 12);
    return gc;
}

// ---- whose turn is it? ------------------------------------------------------------------------------------------
// The members' state machines need different numbers of batches for the same stretch of harmonics (one needs the second
// finite-difference probe of gsl_deriv_central, gsl.rs:233-257, the other does not; one n-chunk takes one bisection, the
// other three), so "every member posts one batch per round" drifts them apart by a batch -- and then they ask for the
// same gamma-integrals one round after each other and nothing is shared.  Only the members that are FURTHEST BEHIND
// post: behind = earlier lobe, then smaller chunk start n_start (symphony.rs:225-276), then earlier phase of the chunk
// (discrete harmonics, probe, second probe, first rule application, bisections).  The others wait; whoever catches up
// posts together with them, and identical requests merge.  Only the order of evaluation changes.
__device__ __forceinline__ unsigned group_turn(const TaskState *park, unsigned alive)
{
    unsigned turn = 0;
    int best_lobe = 0, best_phase = 0;
    double best_n = 0.;
    for (unsigned rem = alive; rem; rem &= rem - 1) {
        const int m = __builtin_ctz(rem);
        const int lobe = uni(park[m].lobe), phase = uni(park[m].phase);
        const double n_start = uni(park[m].n_start);
        int cmp = 0;                             // < 0: this member is further behind than the best so far
        if (!turn) cmp = -1;
        else if (lobe != best_lobe) cmp = lobe < best_lobe ? -1 : 1;
        else if (phase == PH_DISCRETE || best_phase == PH_DISCRETE) cmp = (phase == PH_DISCRETE ? 0 : 1) - (best_phase == PH_DISCRETE ? 0 : 1);
        else if (n_start != best_n) cmp = n_start < best_n ? -1 : 1;
        else if (phase != best_phase) cmp = phase < best_phase ? -1 : 1;
        if (cmp < 0) { turn = 1u << m; best_lobe = lobe; best_phase = phase; best_n = n_start; }
        else if (cmp == 0) turn |= 1u << m;
    }
    return turn;
}

// ---- the whole group on one wave (no cooperation between waves) -----------------------------------------------
// Used by the wavefront emulator and kept as the readable statement of the control flow; the kernel
// (rimphony_hip.hip, group_kernel) runs the same steps with the cooperative tail in between.
//   park[RIM_GROUP]    the members' task states (LDS)
//   outer_lds          RIM_GROUP outer subinterval stores of CAP_GOUTER entries (LDS), outer_spill behind them
//   vals / stats       [nmem] results (lane-uniform)
template <int KIND>
__device__ __forceinline__ void symphony_group(const SymPoint &pt0, const DistParams &dist, unsigned slots, int nmem,
                                               const GKLane &g, double *inner_lds, double *inner_spill,
                                               double *outer_lds, double *outer_spill, TaskState *park, GroupPark *gp,
                                               double *vals, int *stats)
{
    const int lane = g.lane;
    unsigned alive = 0;
    for (int m = 0; m < nmem; m++) {
        SymPoint pt = pt0;
        pt.coeff = group_slot(slots, m) & 1; pt.stokes = group_slot(slots, m) >> 1;
        TaskState T;
        sym_begin(pt, T);
        wv_sync();
        if (lane == 0) park[m] = T;
        if (T.phase != PH_DONE) alive |= 1u << m;
    }
    wv_sync();
    while (alive) {
        // ---- every unfinished member posts its batch ----
        double rq0 = 0., rq1 = 0., rq2 = 0., rq3 = 0.;
        int rl0 = 0, rl1 = 0, rl2 = 0, rl3 = 0;
        unsigned act = 0;                       // per lane: bit m = member m posted a request on this lane
        unsigned posted = 0;
        int nq0 = 0, nq1 = 0, nq2 = 0, nq3 = 0;                              // uniform: B.n_req of each member
        int ph0 = PH_DONE, ph1 = PH_DONE, ph2 = PH_DONE, ph3 = PH_DONE;      // uniform: the phase that posted
        const unsigned turn = group_turn(park, alive);
        for (unsigned rem = turn; rem; rem &= rem - 1) {
            const int m = __builtin_ctz(rem);
            SymPoint pt = pt0;
            pt.coeff = group_slot(slots, m) & 1; pt.stokes = group_slot(slots, m) >> 1;
            const IStore outer = group_store(outer_lds, CAP_GOUTER, outer_spill, SPILL_GOUTER, m);
            TaskState T = park[m];
            task_uniformize(T);
            SymBatch B;
            const bool ok = sym_post(pt, g, outer, T, B);
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

        // ---- merge identical requests (lane-local) and evaluate the entries, two at a time ----
        const GroupClasses gc = group_classify(act, rq0, rq1, rq2, rq3, rl0, rl1, rl2, rl3);
        double gv0 = 0., gv1 = 0., gv2 = 0., gv3 = 0.;
        int bs0 = 0, bs1 = 0, bs2 = 0, bs3 = 0;                              // uniform: status bits of each member's batch
        unsigned long long lanes = wv_ballot(gc.reps != 0);
        int cur_k = -1;
        unsigned cur_bits = 0;
        for (;;) {
            // gather up to two entries
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
#if defined(RIM_WAVE_EMU)
            if (lane == 0) { extern unsigned long long g_emu_hist[64]; g_emu_hist[24 + __builtin_popcount(mk0)]++; if (mk1) g_emu_hist[24 + __builtin_popcount(mk1)]++;
                             g_emu_hist[32 + sel4(em0, ph0, ph1, ph2, ph3)] += __builtin_popcount(mk0) == 1; }
#endif
#if defined(RIM_WAVE_EMU) && defined(RIM_GROUP_TRACE)
            if (lane == 0) printf("  entry k=%d m=%d n=%.17g lobe=%d mask=%x | k=%d m=%d n=%.17g mask=%x  phases %d %d %d %d\n", ek0, em0, n0, lb0, mk0, ek1, em1, n1, mk1, ph0, ph1, ph2, ph3);
#endif
            sym_eval_group<KIND>(pt0, dist, slots, g, inner_lds, inner_spill, gp, n0, lb0, mk0, n1, lb1, mk1);
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

        // ---- every member that posted consumes its results ----
        for (unsigned rem = posted; rem; rem &= rem - 1) {
            const int m = __builtin_ctz(rem);
            SymPoint pt = pt0;
            pt.coeff = group_slot(slots, m) & 1; pt.stokes = group_slot(slots, m) >> 1;
            const IStore outer = group_store(outer_lds, CAP_GOUTER, outer_spill, SPILL_GOUTER, m);
            TaskState T = park[m];
            task_uniformize(T);
            SymBatch B;
            B.req_n = sel4(m, rq0, rq1, rq2, rq3);
            B.req_lobe = sel4(m, rl0, rl1, rl2, rl3);
            B.req_active = ((act >> m) & 1u) != 0;
            B.n_req = sel4(m, nq0, nq1, nq2, nq3);
            B.phase = sel4(m, ph0, ph1, ph2, ph3);
            sym_consume(pt, g, outer, T, B, sel4(m, gv0, gv1, gv2, gv3), uni(sel4(m, bs0, bs1, bs2, bs3)));
            wv_sync();
            if (lane == 0) park[m] = T;
            if (T.phase == PH_DONE) alive &= ~(1u << m);
        }
        wv_sync();
    }
    for (int m = 0; m < nmem; m++) {
        SymPoint pt = pt0;
        pt.coeff = group_slot(slots, m) & 1; pt.stokes = group_slot(slots, m) >> 1;
        TaskState T = park[m];
        task_uniformize(T);
        int st = 0;
        vals[m] = sym_result(pt, T, st);
        stats[m] = st;
    }
}

}  // namespace rim
#endif
