// rimphony_group.hip -- group_kernel<KIND>: the Symphony coefficients of a parameter point in lock-step (gfx950 only).
//
// Task = (point, group); group IQ = the selected ones of {j_I, alpha_I, j_Q, alpha_Q}, group V = those of {j_V, alpha_V}
// (symphony_group.h says why and what is shared).  The kernel has the shape of coop_kernel (rimphony_hip.hip): a
// persistent grid of single-wave workgroups pulling tasks from one atomic counter, and a cooperative tail -- once the
// queue is empty, a wave that still owns a task publishes the merged ENTRIES (n, lobe, member mask) of each round on
// its board slot, idle waves claim single entries, run the members' gamma-integrals in lock-step and hand back one
// value, status and sample count per member.  An entry is a pure function of (point, group, n, lobe, mask), so who
// evaluates it changes no bit; the members' sample counts travel with the values and are booked by the owner when
// it consumes the batch, so the per-coefficient work counters do not depend on who helped either.
//
// A translation unit of its own: hipcc's code generation for a kernel of this size depends on what else is in the
// unit (rimphony_internal.h).
#include <hip/hip_runtime.h>
#include "../../include/rimphony_hip.h"
#include "symphony_group.h"
#include "heyvaerts_group.h"
#include "rimphony_internal.h"
#include "coop_common.h"
#include "group_launch.h"

using namespace rim;

// ---- the two problems the group kernel runs ---------------------------------------------------------------------
// A problem supplies the context of a task, the members' state machines (begin / post / consume / result: the solo
// code of symphony_wave.h / heyvaerts_wave.h, one instance per member), whose turn it is, and the lock-step
// evaluation of one or two merged entries.  A member is named by its output slot (lib.rs:176-177).
template <int KIND>
struct SymGroupProblem {
    typedef TaskState Task;
    typedef GroupPark Park;
    struct Ctx { SymPoint pt; DistParams d; };
    enum : int { QUEUE = 0, TAIL_WORD = 14, STATS_WORD = 8, WAVES = RIM_GROUP_WAVES, EXTRA_LDS_DOUBLES = 1,
                 GSPILL_OUTER = SPILL_GOUTER, SPILL_PER_WAVE = SPILL_GROUP_DOUBLES_PER_WAVE };
    static __device__ __forceinline__ void init(const SymArgs &, Ctx &c, double *)
    { c.pt.s = 0.; c.pt.cos_th = 0.; c.pt.sin_th = 0.; c.pt.coeff = 0; c.pt.stokes = 0; }
    static __device__ __forceinline__ void load(const SymArgs &a, size_t i, unsigned slots, Ctx &c, double &norm)
    { load_context<KIND>(a, i, group_slot(slots, 0), c.pt, c.d, norm); }
    static __device__ __forceinline__ SymPoint member_point(const Ctx &c, int slot)
    { SymPoint pt = c.pt; pt.coeff = slot & 1; pt.stokes = slot >> 1; return pt; }
    static __device__ __forceinline__ void begin(const Ctx &c, int slot, Task &T) { sym_begin(member_point(c, slot), T); }
    static __device__ __forceinline__ void uniformize(Task &T) { task_uniformize(T); }
    static __device__ __forceinline__ bool done(const Task &T) { return T.phase == PH_DONE; }
    static __device__ __forceinline__ int batches(const Task &T) { return T.batches; }
    static __device__ __forceinline__ bool post(const Ctx &c, int slot, const GKLane &g, const IStore &outer, Task &T, SymBatch &B)
    { return sym_post(member_point(c, slot), g, outer, T, B); }
    static __device__ __forceinline__ void consume(const Ctx &c, int slot, const GKLane &g, const IStore &outer, Task &T,
                                                   const SymBatch &B, double gval, int bst)
    { sym_consume(member_point(c, slot), g, outer, T, B, gval, bst); }
    static __device__ __forceinline__ double result(const Ctx &c, int slot, const Task &T, int &st)
    { return sym_result(member_point(c, slot), T, st); }
    static __device__ __forceinline__ unsigned turn(const Task *park, unsigned alive) { return group_turn(park, alive); }
    static __device__ __forceinline__ void eval(const Ctx &c, unsigned slots, const GKLane &g, double *lds, double *spill, Park *gp,
                                                double x0, int t0, unsigned m0, double x1, int t1, unsigned m1)
    { sym_eval_group<KIND>(c.pt, c.d, slots, g, lds, spill, gp, x0, t0, m0, x1, t1, m1); }
};

template <int KIND>
struct HeyGroupProblem {
    typedef HeyTask Task;
    typedef GroupParkBase Park;
    struct Ctx { HeyPoint pt; DistParams d; HeyConsts hc; };
    enum : int { QUEUE = 4, TAIL_WORD = 15, STATS_WORD = 10, WAVES = RIM_HEY_GROUP_WAVES, EXTRA_LDS_DOUBLES = 1,
                 GSPILL_OUTER = SPILL_HEYGOUTER, SPILL_PER_WAVE = SPILL_HEYGROUP_DOUBLES_PER_WAVE };
    static __device__ __forceinline__ void init(const SymArgs &a, Ctx &c, double *extra_lds)
    {
        c.hc = hey_consts();
        c.pt.s = 0.; c.pt.cos_th = 0.; c.pt.sin_th = 0.; c.pt.sigma0 = 0.; c.pt.sigma0_sq = 0.; c.pt.dinv = 0.; c.pt.endless_gamma = RIM_INF; c.pt.stokes = STOKES_Q;
    }
    static __device__ __forceinline__ void load(const SymArgs &a, size_t i, unsigned, Ctx &c, double &norm)
    {
        HeyPoint &pt = c.pt;
        pt.s = uni(a.s[i]);
        rim_sincos(a.theta[i], &pt.sin_th, &pt.cos_th);
        pt.sin_th = uni(pt.sin_th);
        pt.cos_th = uni(pt.cos_th);
        hey_point_derive(pt);
        pt.sigma0 = uni(pt.sigma0); pt.sigma0_sq = uni(pt.sigma0_sq); pt.dinv = uni(pt.dinv);
        pt.stokes = STOKES_Q;
        load_params<KIND>(a.pp, i, c.d);
        norm = uni(a.norm[i]);
        dist_prepare<KIND>(c.d, norm);
#pragma unroll
        for (int k = 0; k < 5; k++) c.d.par[k] = uni(c.d.par[k]);
        c.d.inv_gamma_cutoff = uni(c.d.inv_gamma_cutoff);
        c.d.inv_kappa_width = uni(c.d.inv_kappa_width);
        c.d.neg_inverse_t = uni(c.d.neg_inverse_t);
        hey_point_endless<KIND>(pt, c.d);
        pt.endless_gamma = uni(pt.endless_gamma);
    }
    static __device__ __forceinline__ HeyPoint member_point(const Ctx &c, int slot)
    { HeyPoint pt = c.pt; pt.stokes = hey_slot_stokes(slot); return pt; }
    static __device__ __forceinline__ void begin(const Ctx &c, int slot, Task &T) { hey_begin(member_point(c, slot), T); }
    static __device__ __forceinline__ void uniformize(Task &T) { hey_uniformize(T); }
    static __device__ __forceinline__ bool done(const Task &T) { return T.stage == HS_DONE; }
    static __device__ __forceinline__ int batches(const Task &T) { return T.batches; }
    static __device__ __forceinline__ bool post(const Ctx &c, int slot, const GKLane &g, const IStore &outer, Task &T, SymBatch &B)
    { return hey_post(member_point(c, slot), g, outer, T, B); }
    static __device__ __forceinline__ void consume(const Ctx &c, int slot, const GKLane &g, const IStore &outer, Task &T,
                                                   const SymBatch &B, double gval, int bst)
    { hey_consume(member_point(c, slot), g, outer, T, B, gval, bst); }
    static __device__ __forceinline__ double result(const Ctx &, int, const Task &T, int &st) { return hey_result(T, st); }
    static __device__ __forceinline__ unsigned turn(const Task *park, unsigned alive) { return hey_group_turn(park, alive); }
    static __device__ __forceinline__ void eval(const Ctx &c, unsigned slots, const GKLane &g, double *lds, double *spill, Park *gp,
                                                double x0, int t0, unsigned m0, double x1, int t1, unsigned m1)
    { hey_eval_group<KIND>(c.pt, c.d, c.hc, slots, g, lds, spill, gp, x0, t0, m0, x1, t1, m1); }
};

// claim the next unevaluated entry of batch `seq` on a group slot; -1 if there is none (or the batch is over)
__device__ __forceinline__ int group_claim(GroupSlot *slot, unsigned seq, int lane)
{
    int k = -1;
    if (lane == 0) {
        for (int tries = 0; tries < 64; tries++) {
            const unsigned long long c = __hip_atomic_load(&slot->claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned) (c >> 32) != seq || !claim_open(c)) break;
            unsigned long long expect = c;
            if (__hip_atomic_compare_exchange_strong(&slot->claim, &expect, c + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) { k = (int) (c & 0xff); break; }
        }
    }
    return __builtin_amdgcn_readfirstlane(k);
}

template <class P>
__global__ __launch_bounds__(64, P::WAVES) void group_kernel(GroupArgs ga)
{
    const SymArgs &a = ga.base;
    __shared__ double s_tab[96];
    __shared__ double s_ginner[RIM_GROUP * RIM_ISTORE_DOUBLES(CAP_GINNER)];
    __shared__ typename P::Task s_park[RIM_GROUP];
    __shared__ typename P::Park s_gp;
    __shared__ double s_extra[P::EXTRA_LDS_DOUBLES];
    const GKLane g = gk_lane_init(s_tab);
    const int lane = g.lane;
    double *const inner_spill = ga.gspill + (size_t) blockIdx.x * P::SPILL_PER_WAVE;
    double *const outer_spill = inner_spill + RIM_GROUP * RIM_ISTORE_DOUBLES(SPILL_GINNER);
    if (threadIdx.x == 0) {
        s_gp.ctr = WaveCounters{0, 0, 0};
        s_gp.member_passes = 0; s_gp.stash_filed = 0; s_gp.stash_used = 0;
        s_gp.hb = nullptr;
    }
#if defined(RIM_PROF) && defined(__HIP_DEVICE_COMPILE__)
    if (threadIdx.x < 32) rim_prof_lds[threadIdx.x] = 0;
    __syncthreads();
#endif
    RIM_PROF_T(t_kernel);
#if defined(RIM_PROF) && !defined(RIM_PROF_COUNTS) && !defined(RIM_ISA_MARKS) && defined(__HIP_DEVICE_COMPILE__)
    unsigned long long t_helper = 0;
#endif

    GroupSlot *const my = ga.gboard + blockIdx.x;
    unsigned *const flag_exhausted = a.board_flags + BOARD_FLAG_EXHAUSTED;
    unsigned *const flag_active = a.board_flags + BOARD_FLAG_ACTIVE;
    unsigned *const flag_idle = a.board_flags + BOARD_FLAG_IDLE;
    unsigned *const hints = a.board_flags + BOARD_HINTS;
    const bool coop = ga.gboard != nullptr && a.board_flags != nullptr && ga.coop != 0;
    const unsigned nboard = gridDim.x;
    unsigned seq = 0;                      // sequence number of this wave's published batches
    bool counted_idle = false;
    int backoff = 1;
    unsigned long long idle_since = 0;
    unsigned n_polls = 0;

    const unsigned long long ntasks = (unsigned long long) a.n * (unsigned long long) ga.ngroups;
    typename P::Ctx cx;                    // context of the entries being evaluated (own task or a helped one)
    P::init(a, cx, s_extra);
    unsigned slots = 0;
    unsigned long long *const queue = a.queue + P::QUEUE;
    size_t own_i = 0;
    unsigned own_slots = 0;
    int own_nmem = 0, own_group = 0;
    unsigned alive = 0;
    bool have_task = false, helper = false, board_dead = false;
    unsigned last_hint = 0;

    __builtin_amdgcn_s_setprio(3);
    for (;;) {
        // per-lane requests of the round (owner only)
        double rq0 = 0., rq1 = 0., rq2 = 0., rq3 = 0.;
        int rl0 = 0, rl1 = 0, rl2 = 0, rl3 = 0;
        unsigned act = 0, posted = 0;
        int nq0 = 0, nq1 = 0, nq2 = 0, nq3 = 0;
        int ph0 = PH_DONE, ph1 = PH_DONE, ph2 = PH_DONE, ph3 = PH_DONE;
        GroupClasses gc;
        gc.reps = 0; gc.cmask = 0;
        int ebase = 0, ecount = 0;             // this lane's first entry ordinal, number of entries of the round
        GroupSlot *src = my;
        unsigned src_seq = 0;
        bool shared = false;

        if (!helper) {
            // ---------- owner: next round of the current task (fetching a task first if needed) ----------
            if (!have_task) {
                const unsigned long long t = wave_next_task(queue, lane);
                if (t >= ntasks) {
                    helper = true;
#if defined(RIM_TAIL_DIAG)     // (tools/tail_times.py: when did the queue run dry, when did the launch end)
                    if (lane == 0) atomicCAS(a.queue + 12, 0ull, wall_clock64());
#endif
                    RIM_PROF_ADD(27, t_kernel);         // owner phase of this wave
#if defined(RIM_PROF) && !defined(RIM_PROF_COUNTS) && !defined(RIM_ISA_MARKS) && defined(__HIP_DEVICE_COMPILE__)
                    t_helper = __builtin_readcyclecounter();
#endif
                    if (lane == 0) {
                        __hip_atomic_store(flag_exhausted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_sub(flag_active, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (coop) __hip_atomic_fetch_add(flag_idle, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    counted_idle = true;
                    __builtin_amdgcn_s_setprio(0);
                    if (!coop) break;
                    continue;
                }
                const size_t seqidx = (size_t) (t / (unsigned) ga.ngroups);
                own_i = a.perm ? (size_t) a.perm[seqidx] : seqidx;
                own_group = (int) (t % (unsigned) ga.ngroups);
                own_slots = own_group ? ga.gslots[1] : ga.gslots[0];
                own_nmem = own_group ? ga.gnmem[1] : ga.gnmem[0];
                double norm;
                P::load(a, own_i, own_slots, cx, norm);
                slots = own_slots;
                __syncthreads();
                if (!(norm == norm)) {
                    if (lane == 0) {
                        for (int m = 0; m < own_nmem; m++) {
                            const int sl = group_slot(own_slots, m);
                            a.out[own_i * 8 + sl] = RIM_NAN;
                            if (a.status) a.status[own_i * 8 + sl] = ST_NORM_FAIL | ST_NONFINITE;
                        }
                    }
                    continue;
                }
                alive = 0;
                for (int m = 0; m < own_nmem; m++) {
                    typename P::Task T0;
                    P::begin(cx, group_slot(own_slots, m), T0);
                    __syncthreads();
                    if (lane == 0) s_park[m] = T0;
                    if (!P::done(T0)) alive |= 1u << m;
                }
                __syncthreads();
                have_task = true;
            }
            if (!alive) {
                // every member has finished: results (symphony.rs:144-183)
                for (int m = 0; m < own_nmem; m++) {
                    typename P::Task T = s_park[m];
                    P::uniformize(T);
                    int st = 0;
                    const int sl = group_slot(own_slots, m);
                    const double val = P::result(cx, sl, T, st);
                    if (lane == 0) {
                        a.out[own_i * 8 + sl] = val;
#if defined(RIM_TAIL_DIAG)     // (tools/tail_times.py: the member's number of batches in the upper half of the status word)
                        if (a.status) a.status[own_i * 8 + sl] = st | ((P::batches(T) < 0x7fff ? P::batches(T) : 0x7fff) << 16);
#else
                        if (a.status) a.status[own_i * 8 + sl] = st;
#endif
                        // the heaviest member of the launch: its sequential chain of batches bounds the launch's tail
                        atomicMax(a.queue + P::TAIL_WORD, ((unsigned long long) P::batches(T) << 40) | ((unsigned long long) own_i & 0xffffffffffull));
                    }
                }
                __syncthreads();
                have_task = false;
                continue;
            }
            // the members that are furthest behind post their batches (symphony_group.h: group_turn)
            RIM_PROF_T(t_post);
            const unsigned turn = P::turn(s_park, alive);
            for (unsigned rem = turn; rem; rem &= rem - 1) {
                const int m = __builtin_ctz(rem);
                const IStore outer = group_store(nullptr, 0, outer_spill, P::GSPILL_OUTER, m);
                typename P::Task T = s_park[m];
                P::uniformize(T);
                SymBatch B;
                const bool ok = P::post(cx, group_slot(own_slots, m), g, outer, T, B);
                __syncthreads();
                if (lane == 0) s_park[m] = T;
                if (!ok) { alive &= ~(1u << m); continue; }
                posted |= 1u << m;
                put4(m, B.req_n, rq0, rq1, rq2, rq3);
                put4(m, B.req_lobe, rl0, rl1, rl2, rl3);
                put4(m, B.n_req, nq0, nq1, nq2, nq3);
                put4(m, B.phase, ph0, ph1, ph2, ph3);
                if (B.req_active) act |= 1u << m;
            }
            __syncthreads();
            RIM_PROF_ADD(24, t_post);
            if (!posted) continue;
            gc = group_classify(act, rq0, rq1, rq2, rq3, rl0, rl1, rl2, rl3);
            {
                const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
                for (int j = 0; j < RIM_GROUP; j++) {
                    const unsigned long long bj = wv_ballot(((gc.reps >> j) & 1u) != 0);
                    ebase += __builtin_popcountll(bj & below);
                    ecount += __builtin_popcountll(bj);
                }
            }

            // publish the round when some wave is idle
            unsigned idle = 0, actw = 1;
            if (lane == 0 && coop) {
                if (__hip_atomic_load(flag_exhausted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    idle = __hip_atomic_load(flag_idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    actw = __hip_atomic_load(flag_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            idle = (unsigned) __builtin_amdgcn_readfirstlane((int) idle);
            actw = (unsigned) __builtin_amdgcn_readfirstlane((int) actw);
            shared = idle != 0 && ecount >= 2 && ecount <= RIM_GROUP_ENTRIES && !board_dead;
            if (shared) {
                seq += 1;
                src_seq = seq;
#pragma unroll
                for (int m = 0; m < RIM_GROUP; m++) {
                    if ((gc.reps >> m) & 1u) {
                        const int e = ebase + __builtin_popcount(gc.reps & ((1u << m) - 1u));
                        bput(&my->req_n[e], rim_bits(sel4(m, rq0, rq1, rq2, rq3)));
                        bput(&my->req_tag[e], sel4(m, rl0, rl1, rl2, rl3) | (int) (((gc.cmask >> (4 * m)) & 15u) << 1));
                    }
                }
                if (lane == 0) {
                    bput(&my->point, (unsigned long long) own_i);
                    bput(&my->slots, own_slots);
                    bput(&my->done, 0u);
                }
                drain_vmem();
                __syncthreads();
                if (lane == 0)
                    __hip_atomic_store(&my->claim, ((unsigned long long) seq << 32) | ((unsigned long long) ecount << 8),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                drain_vmem();          // the claim word is out before anybody can see the hint
                __syncthreads();
                {
                    const unsigned span = hint_span(actw);
                    const unsigned channel = ((unsigned) blockIdx.x + seq) & (span - 1u);
                    if (((unsigned) lane & (span - 1u)) == channel)
                        __hip_atomic_store(&hints[(unsigned) lane * BOARD_HINT_STRIDE], (seq << 16) | ((unsigned) blockIdx.x + 1u),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        } else {
            // ---------- helper: find a published round through this wave's hint line ----------
            unsigned h = 0, actw = 1;
            int leave = 0;
            unsigned long long c = 0;
            if (lane == 0) {
                if ((n_polls & 15u) == 0) {
                    actw = __hip_atomic_load(flag_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // more idle waves than the remaining owners can feed only add polling traffic: the surplus leaves
                    const unsigned keep = actw * RIM_GROUP_HELPERS_PER_OWNER + 32u;
                    if (counted_idle && actw != 0 &&
                        __hip_atomic_load(flag_idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > keep) {
                        const unsigned before = __hip_atomic_fetch_sub(flag_idle, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (before > keep) leave = 1;
                        else __hip_atomic_fetch_add(flag_idle, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (!leave)
                    h = __hip_atomic_load(&hints[((unsigned) blockIdx.x & 63u) * BOARD_HINT_STRIDE], __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
                const unsigned hs = h & 0xffffu;
                if (h != last_hint && hs != 0 && hs <= nboard) {
                    c = __hip_atomic_load(&ga.gboard[hs - 1].claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!claim_open(c)) { last_hint = h; h = 0; }      // nothing left of that round: wait for a new hint
                } else {
                    h = 0;
                }
            }
            h = (unsigned) __builtin_amdgcn_readfirstlane((int) h);
            actw = (unsigned) __builtin_amdgcn_readfirstlane((int) actw);
            if (__builtin_amdgcn_readfirstlane(leave)) break;
            n_polls += 1;
            if (h == 0) {
                if (actw == 0) break;       // every task is finished
                // a wave that has seen nothing to do for 2 s leaves: it holds no claim, so leaving is always safe, and
                // it bounds every wait of this kernel
                const unsigned long long now = wall_clock64();
                if (idle_since == 0) idle_since = now;
                else if (now - idle_since > a.idle_ticks) {
                    if (counted_idle && lane == 0)
                        __hip_atomic_fetch_sub(flag_idle, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                for (int w = 0; w < backoff; w++) __builtin_amdgcn_s_sleep(127);
                if (backoff < 16) backoff *= 2;
                continue;
            }
            h &= 0xffffu;
            const unsigned long long cw = bcast_u64(c);
            src = ga.gboard + (h - 1u);
            src_seq = (unsigned) (cw >> 32);
            shared = true;
        }

        // ---------- evaluate entries of `src` (the integrand lives here, once) ----------
        double gv0 = 0., gv1 = 0., gv2 = 0., gv3 = 0.;
        unsigned bsp = 0;                   // status bits of each member's batch, 8 bits per member (wave-uniform)
        bool ctx_loaded = !helper;
        int got = 0;
        size_t work_i = own_i;
        bool redo;
        do {
        redo = false;
        unsigned long long lanes = helper ? 0ull : wv_ballot(gc.reps != 0);
        int cur_k = -1;
        unsigned cur_bits = 0;
        for (;;) {
            double n0 = 0., n1 = 0.;
            int lb0 = 0, lb1 = 0, ek0 = -1, ek1 = -1, eidx = -1;
            unsigned mk0 = 0, mk1 = 0;
            if (shared) {
                eidx = group_claim(src, src_seq, lane);      // ordinal of the entry within the round
                if (eidx < 0) break;
                if (helper && !ctx_loaded) {
                    double norm;
                    const size_t hi = (size_t) bcast_u64(bget(&src->point));
                    slots = (unsigned) __builtin_amdgcn_readfirstlane((int) bget(&src->slots));
                    P::load(a, hi, slots, cx, norm);
                    work_i = hi;
                    ctx_loaded = true;
                    if (counted_idle) {
                        if (lane == 0) __hip_atomic_fetch_sub(flag_idle, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        counted_idle = false;
                    }
                    backoff = 1;
                    idle_since = 0;
                }
                n0 = uni(rim_frombits(bget(&src->req_n[eidx])));
                const int tag = __builtin_amdgcn_readfirstlane(bget(&src->req_tag[eidx]));
                lb0 = tag & 1;
                mk0 = ((unsigned) tag >> 1) & 15u;
                n1 = n0; lb1 = lb0;
            } else {
                // the wave's own round: two entries at a time share their first rule application
                for (int e = 0; e < 2; e++) {
                    if (cur_bits == 0) {
                        if (!lanes) break;
                        cur_k = __builtin_ffsll((long long) lanes) - 1;
                        lanes &= lanes - 1;
                        cur_bits = (unsigned) wv_readlane((int) gc.reps, cur_k);
                    }
                    const int em = __builtin_ctz(cur_bits);
                    cur_bits &= cur_bits - 1;
                    const double nn = readlane_d(sel4(em, rq0, rq1, rq2, rq3), cur_k);
                    const int ll = wv_readlane(sel4(em, rl0, rl1, rl2, rl3), cur_k);
                    const unsigned mm = ((unsigned) wv_readlane((int) gc.cmask, cur_k) >> (4 * em)) & 15u;
                    if (e == 0) { ek0 = cur_k; n0 = nn; lb0 = ll; mk0 = mm; n1 = nn; lb1 = ll; }
                    else { ek1 = cur_k; n1 = nn; lb1 = ll; mk1 = mm; }
                }
                if (ek0 < 0) break;
            }
            // (wave-uniform by construction -- readlane / readfirstlane results merged over the two paths above; saying so
            // keeps the member loops below on the scalar unit)
            mk0 = uni(mk0); mk1 = uni(mk1); ek0 = uni(ek0); ek1 = uni(ek1);
            got += 1;
            RIM_PROF_T(t_req);
            P::eval(cx, slots, g, s_ginner, inner_spill, &s_gp, n0, lb0, mk0, n1, lb1, mk1);
            RIM_PROF_ADD(9, t_req);
            if (helper) RIM_PROF_ADD(28, t_req);        // ... of them, for other waves' tasks
            if (shared) {
                // hand the members' values, status bits and sample counts back through the board
                if (lane == 0) {
                    int stw = 0;
                    for (unsigned rem = mk0; rem; rem &= rem - 1) {
                        const int m = __builtin_ctz(rem);
                        int st = 0;
                        const double v = group_entry_value(&s_gp, m, 0, st);
                        bput(&src->res[eidx * RIM_GROUP + m], rim_bits(v));
                        bput(&src->res_samples[eidx * RIM_GROUP + m], s_gp.mem[m].samples);
                        stw |= (st & 0xff) << (8 * m);
                    }
                    bput(&src->res_status[eidx], stw);
                    drain_vmem();
                    __hip_atomic_fetch_add(&src->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                for (int e = 0; e < 2; e++) {
                    const unsigned mk = e ? mk1 : mk0;
                    const int ek = e ? ek1 : ek0;
                    for (unsigned rem = mk; rem; rem &= rem - 1) {
                        const int m = __builtin_ctz(rem);
                        int st = 0;
                        const double v = group_entry_value(&s_gp, m, e, st);
                        bsp |= (unsigned) (st & 0xff) << (8 * m);
                        if (lane == ek) put4(m, v, gv0, gv1, gv2, gv3);
                    }
                }
                if (a.work && lane == 0) {
                    // (both entries of a pair are booked in mem[m].samples of their members)
                    for (unsigned rem = mk0 | mk1; rem; rem &= rem - 1) {
                        const int m = __builtin_ctz(rem);
                        atomicAdd(a.work + work_i * 8 + (size_t) group_slot(slots, m), (unsigned long long) s_gp.mem[m].samples);
                    }
                }
            }
        }
        if (helper) break;

        // ---------- owner: collect a shared round ----------
        if (shared) {
            const unsigned want = (unsigned) ecount;
            bool complete = false;
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                unsigned dn = 0;
                if (lane == 0) dn = __hip_atomic_load(&my->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dn = (unsigned) __builtin_amdgcn_readfirstlane((int) dn);
                if (dn >= want) { complete = true; break; }
                if (wall_clock64() - t0 > a.owner_ticks) break;
                __builtin_amdgcn_s_sleep(32);
            }
            if (lane == 0)
                __hip_atomic_store(&my->claim, (unsigned long long) seq << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (complete) {
                // lane k reads the results of its own entries: member mm's value comes from the entry of its class
                int st_l0 = 0, st_l1 = 0, st_l2 = 0, st_l3 = 0;
#pragma unroll
                for (int mm = 0; mm < RIM_GROUP; mm++) {
                    if ((act >> mm) & 1u) {
                        int r = 0;           // the member that represents mm's class on this lane
#pragma unroll
                        for (int q = 0; q < RIM_GROUP; q++)
                            if ((gc.cmask >> (4 * q)) & (1u << mm)) r = q;
                        const int e = ebase + __builtin_popcount(gc.reps & ((1u << r) - 1u));
                        const double v = rim_frombits(bget(&my->res[e * RIM_GROUP + mm]));
                        const int stl = (bget(&my->res_status[e]) >> (8 * mm)) & 0xff;
                        put4(mm, v, gv0, gv1, gv2, gv3);
                        put4(mm, stl, st_l0, st_l1, st_l2, st_l3);
                        if (a.work) atomicAdd(a.work + own_i * 8 + (size_t) group_slot(own_slots, mm),
                                              (unsigned long long) bget(&my->res_samples[e * RIM_GROUP + mm]));
                    }
                }
#pragma unroll
                for (int mm = 0; mm < RIM_GROUP; mm++) {
                    const int stl = sel4(mm, st_l0, st_l1, st_l2, st_l3);
                    int bst = 0;
                    if (wv_ballot((stl & ST_INNER_FAIL) != 0)) bst |= ST_INNER_FAIL;
                    if (wv_ballot((stl & ST_STORE_FULL) != 0)) bst |= ST_STORE_FULL;
                    bsp = (bsp & ~(0xffu << (8 * mm))) | ((unsigned) bst << (8 * mm));
                }
            } else {
                // a claimed entry has not come back within the bound: the owner closes the round, evaluates ALL of it
                // itself (same bits, only later; nothing of the failed attempt has been booked) and never publishes again
                board_dead = true;
                shared = false;
                gv0 = gv1 = gv2 = gv3 = 0.;
                bsp = 0;
                redo = true;
            }
        }
        } while (redo);
        if (helper) {
            if (got == 0) {
                for (int w = 0; w < backoff; w++) __builtin_amdgcn_s_sleep(127);
                if (backoff < 16) backoff *= 2;
            } else {
                if (lane == 0) __hip_atomic_fetch_add(flag_idle, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                counted_idle = true;
            }
            continue;
        }
        __syncthreads();
        // ---------- every member that posted consumes its results ----------
        RIM_PROF_T(t_cons);
        for (unsigned rem = posted; rem; rem &= rem - 1) {
            const int m = __builtin_ctz(rem);
            const IStore outer = group_store(nullptr, 0, outer_spill, P::GSPILL_OUTER, m);
            typename P::Task T = s_park[m];
            P::uniformize(T);
            SymBatch B;
            B.req_n = sel4(m, rq0, rq1, rq2, rq3);
            B.req_lobe = sel4(m, rl0, rl1, rl2, rl3);
            B.req_active = ((act >> m) & 1u) != 0;
            B.n_req = sel4(m, nq0, nq1, nq2, nq3);
            B.phase = sel4(m, ph0, ph1, ph2, ph3);
            P::consume(cx, group_slot(own_slots, m), g, outer, T, B, sel4(m, gv0, gv1, gv2, gv3), (int) ((uni(bsp) >> (8 * m)) & 0xffu));
            __syncthreads();
            if (lane == 0) s_park[m] = T;
            if (P::done(T)) alive &= ~(1u << m);
        }
        __syncthreads();
        RIM_PROF_ADD(26, t_cons);
    }

    RIM_PROF_ADD(0, t_kernel);
#if defined(RIM_TAIL_DIAG)
    if (g.lane == 0) atomicMax(a.queue + 13, wall_clock64());
#endif
#if defined(RIM_PROF) && !defined(RIM_PROF_COUNTS) && !defined(RIM_ISA_MARKS) && defined(__HIP_DEVICE_COMPILE__)
    if (t_helper) RIM_PROF_ADD(29, t_helper);           // helper phase of this wave (helping + polling)
#endif
    __syncthreads();
#if defined(RIM_PROF) && defined(__HIP_DEVICE_COMPILE__)
    if (ga.prof && threadIdx.x < 32) ga.prof[(size_t) blockIdx.x * 32 + threadIdx.x] += rim_prof_lds[threadIdx.x];
#endif
    if (g.lane == 0) {
        atomicAdd(queue + 1, s_gp.ctr.samples);
        atomicAdd(queue + 2, s_gp.ctr.steps);
        atomicAdd(queue + 3, s_gp.ctr.inner_qags);
        atomicAdd(a.queue + P::STATS_WORD, s_gp.member_passes);
        atomicAdd(a.queue + P::STATS_WORD + 1, s_gp.stash_filed);
    }
}

// ---- launch interface (group_launch.h) -----------------------------------------------------------------------
template <class P>
static const void *kernel_ptr() { return reinterpret_cast<const void *>(group_kernel<P>); }

const void *rim_group_kernel(int kind, int faraday)
{
    if (faraday) {
        switch (kind) {
        case 0: return kernel_ptr<HeyGroupProblem<0>>();
        case 1: return kernel_ptr<HeyGroupProblem<1>>();
        case 2: return kernel_ptr<HeyGroupProblem<2>>();
        default: return kernel_ptr<HeyGroupProblem<3>>();
        }
    }
    switch (kind) {
    case 0: return kernel_ptr<SymGroupProblem<0>>();
    case 1: return kernel_ptr<SymGroupProblem<1>>();
    case 2: return kernel_ptr<SymGroupProblem<2>>();
    default: return kernel_ptr<SymGroupProblem<3>>();
    }
}

int rim_group_waves(int faraday) { return faraday ? RIM_HEY_GROUP_WAVES : RIM_GROUP_WAVES; }

template <class P>
static void launch(unsigned grid, hipStream_t st, const GroupArgs &ga)
{
    hipLaunchKernelGGL(group_kernel<P>, dim3(grid), dim3(64), RIM_DYN_LDS, st, ga);
}

int rim_group_launch(int kind, int faraday, unsigned grid, hipStream_t st, const GroupArgs &ga)
{
    if (faraday) {
        switch (kind) {
        case 0: launch<HeyGroupProblem<0>>(grid, st, ga); break;
        case 1: launch<HeyGroupProblem<1>>(grid, st, ga); break;
        case 2: launch<HeyGroupProblem<2>>(grid, st, ga); break;
        default: launch<HeyGroupProblem<3>>(grid, st, ga); break;
        }
    } else {
        switch (kind) {
        case 0: launch<SymGroupProblem<0>>(grid, st, ga); break;
        case 1: launch<SymGroupProblem<1>>(grid, st, ga); break;
        case 2: launch<SymGroupProblem<2>>(grid, st, ga); break;
        default: launch<SymGroupProblem<3>>(grid, st, ga); break;
        }
    }
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return RIMPHONY_OK;
    rim_set_last_error("group_kernel launch", hipGetErrorString(e));
    return RIMPHONY_EHIP;
}
