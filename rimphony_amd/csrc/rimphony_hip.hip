// rimphony_hip.hip -- kernels and C ABI of librimphony_hip.so (gfx950 only).
//
// Kernels (one 64-lane wavefront per workgroup; each wave owns its LDS):
//   norm_kernel<KIND>      full_calculation(): per-point normalisation integral
//   coop_kernel<P>         (point, coefficient) tasks pulled from an atomic queue, with a cooperative tail;
//                          P = SymphonyProblem<KIND> (j, alpha) or HeyvaertsProblem<KIND> (rho_Q, rho_V)
//   highfreq_kernel        closed-form high-frequency rho_Q, rho_V
//   bessel_kernel, integrand_kernel, gamma_integral_kernel, qag_selftest_kernel
//                          unit seams used by the parity tests
//
// Launch geometry: a persistent grid of RIM_WAVES_PER_CU x 256 single-wave
// workgroups; per-task cost spans orders of magnitude, so
// tasks are handed out dynamically, one atomicAdd per task.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <mutex>
#include <thread>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
#include "../../include/rimphony_hip.h"
#include "symphony_wave.h"
#include "heyvaerts_wave.h"
#include "highfreq.h"

using namespace rim;

#include "rimphony_internal.h"
#include "coop_common.h"
#include "group_launch.h"

// ------------------------------------------------------------------------------
// normalisation integrands (power_law.rs:95-96, pitchy_kappa.rs:100-104; the
// thermal one is the substituted form documented in oracle/rimo_dist.c)
// ------------------------------------------------------------------------------

template <int KIND>
__device__ inline double norm_integrand(const DistParams &d, double g)
{
    if (KIND == DIST_POWER_LAW || KIND == DIST_PITCHY_PL) {
        return rim_pow(g, -d.par[0]) * rim_exp(-g * d.inv_gamma_cutoff);
    } else if (KIND == DIST_PITCHY_KAPPA) {
        return g * rim_sqrt(g * g - 1.) *
            rim_pow(1. + (g - 1.) * d.inv_kappa_width, -(d.par[0] + 1.)) *
            rim_exp(-g * d.inv_gamma_cutoff);
    } else {
        const double u = g;
        const double u2 = u * u;
        const double gg = 1. + u2;
        return gg * (u * rim_sqrt(u2 + 2.)) * rim_exp(d.neg_inverse_t * gg) * (2. * u);
    }
}

__device__ inline double hyperg_2F1_at_1(double a, double b, double c)
{
    const double lc = rim_lgamma_pos(c);
    const double lcab = rim_lgamma_pos(c - a - b);
    const double lca = rim_lgamma_pos(c - a);
    const double lcb = rim_lgamma_pos(c - b);
    return rim_exp(lc + lcab - lca - lcb);
}

template <int KIND>
__global__ __launch_bounds__(64) void norm_kernel(ParamPtrs pp, size_t n, double *norm, unsigned long long *queue,
                                                    double *spill_base)
{
    __shared__ double s_tab[96];
    __shared__ double s_store[RIM_ISTORE_DOUBLES(CAP_NORM)];
    const GKLane g = gk_lane_init(s_tab);
    const IStore st = istore_carve(s_store, CAP_NORM, spill_base + (size_t) blockIdx.x * SPILL_DOUBLES_PER_WAVE, SPILL_INNER);
    __shared__ QagPark s_qpark;
    if (threadIdx.x == 0) { s_qpark.ctr = WaveCounters{0, 0, 0}; s_qpark.hb = nullptr; }

    for (;;) {
        const unsigned long long t = wave_next_task(queue, g.lane);
        if (t >= n) break;
        const size_t i = (size_t) t;

        DistParams d;
        load_params<KIND>(pp, i, d);
        dist_prepare<KIND>(d, RIM_NAN);

        double lo, hi, epsrel, pa = 1.;
        if (KIND == DIST_POWER_LAW) { lo = d.par[1]; hi = d.par[2]; epsrel = 1e-8; }
        else if (KIND == DIST_PITCHY_PL) { lo = d.par[2]; hi = d.par[3]; epsrel = 1e-8; pa = hyperg_2F1_at_1(0.5, -0.5 * d.par[1], 1.5); }
        else if (KIND == DIST_PITCHY_KAPPA) {
            const double g_cut = 1. / d.inv_gamma_cutoff;
            lo = 1.; hi = 1e3 * g_cut; epsrel = 1e-8; pa = hyperg_2F1_at_1(0.5, -0.5 * d.par[2], 1.5);
        } else { lo = 0.; hi = rim_sqrt(60. * d.par[0] + 4.); epsrel = 1e-10; }

        auto f = [&](double x, bool active) -> double { return active ? norm_integrand<KIND>(d, x) : 0.; };
        QagState q;
        wave_qag(f, g, st, lo, hi, 0., epsrel, 1000, q, &s_qpark);
        double v = RIM_NAN;
        if (q.status == QAG_SUCCESS) {
            if (KIND == DIST_PITCHY_PL || KIND == DIST_PITCHY_KAPPA) v = 1. / (2. * RIM_TWO_PI * pa * q.result);
            else v = 1. / (2. * RIM_TWO_PI * q.result);
        }
        if (g.lane == 0) norm[i] = v;
    }
}

// ------------------------------------------------------------------------------
// symphony
// ------------------------------------------------------------------------------

#if defined(RIM_PROF)
#define RIM_PROF_ROWS 32768
__device__ unsigned long long g_rim_prof[RIM_PROF_ROWS * 32];
#endif

// ---- the two problems the cooperative kernel runs ------------------------------------------
// A problem supplies the uniform context of a task, the parked task state and the five steps of
// the resumable computation (begin / post / eval / consume / result).  Requests and results have
// the same shape in both: (double abscissa, int tag) -> (double value, int status).
template <int KIND, int PREC = 0>
struct SymphonyProblem {
    struct Ctx { SymPoint pt; DistParams d; };
    typedef TaskState Task;
    typedef QagPark Park;
    enum : unsigned long long { QUEUE = 0, WAVES = RIM_SYM_WAVES, HB_TAG = 0, EXTRA_LDS_DOUBLES = 1, EARLY_HELP = 0, EARLY_SQUAD = 0 };
    static __device__ __forceinline__ void init(const SymArgs &, Ctx &, double *) {}
    static __device__ __forceinline__ void load(const SymArgs &a, size_t i, int slot, Ctx &c, double &norm)
    { load_context<KIND>(a, i, slot, c.pt, c.d, norm); }
    static __device__ __forceinline__ void begin(const Ctx &c, Task &T) { sym_begin(c.pt, T); }
    static __device__ __forceinline__ int early_metric(const Task &) { return 0; }
    static __device__ __forceinline__ void uniformize(Task &T) { task_uniformize(T); }
    static __device__ __forceinline__ bool done(const Task &T) { return T.phase == PH_DONE; }
    struct Stash { unsigned used; };        // (rounds are the Faraday kernel's: HeyvaertsProblem)
    enum { TURBO = 0 };
    static __device__ __forceinline__ void post(const Ctx &c, const GKLane &g, const IStore &outer, Task &T, SymBatch &B, Stash *, int)
    { sym_post(c.pt, g, outer, T, B); }
    static __device__ __forceinline__ int round_n(const Stash &) { return 1; }
    static __device__ __forceinline__ void round_drop(Stash &) {}
    static __device__ __forceinline__ double round_request(const Stash &, const GKLane &, int) { return 0.; }

    // one or two requests ((x1, tag1) only if have1): two gamma-integrals share their first rule application
    static __device__ __forceinline__ void eval2(const Ctx &c, const GKLane &g, const IStore &inner, Park *qp,
                                                 double x0, int tag0, double x1, int tag1, bool have1,
                                                 double &v0, int &st0, double &v1, int &st1)
    { sym_eval_pair<KIND, PREC>(c.pt, c.d, g, inner, qp, x0, tag0, x1, tag1, have1, v0, st0, v1, st1); }
    static __device__ __forceinline__ void consume(const Ctx &c, const GKLane &g, const IStore &outer, Task &T,
                                                   const SymBatch &B, double gval, int bst, Stash *, unsigned long long *, unsigned long long *,
                                                   const AssistSlot *, int, int)
    { sym_consume(c.pt, g, outer, T, B, gval, bst); }
    static __device__ __forceinline__ double result(const Ctx &c, const Task &T, int &st) { return sym_result(c.pt, T, st); }
};

template <int KIND>
struct HeyvaertsProblem {
    struct Ctx { HeyPoint pt; DistParams d; HeyConsts hc; };
    typedef HeyTask Task;
    typedef QagParkBase Park;
    // EARLY_SQUAD: the default size of the squad that serves the longest outer quadrature from the start of a launch
    // (coop_common.h).  Outer quadratures that run to GSL's limit of 4096 bisections occur on the power-law table (one
    // task in ~1e5: 4122 / 3775 batches -- one title, 64 waves) and on the pitchy-kappa table (four of 2844 .. 4132
    // batches among the first 16793 rows and a dozen of 300 .. 760 -- four titles, 256 waves); the thermal table's long
    // chains were the endless marching loops that hey_qr_is_endless() now ends at once, the pitchy power-law table has
    // none past 244 batches -- there the squad would only cost its share of the grid.
    enum : unsigned long long { QUEUE = 4, WAVES = RIM_HEY_WAVES, HB_TAG = 1ull << 62, EXTRA_LDS_DOUBLES = 1, EARLY_HELP = 1,
                                EARLY_SQUAD = KIND == DIST_POWER_LAW ? RIM_EARLY_SQUAD_DEFAULT : KIND == DIST_PITCHY_KAPPA ? 4 * RIM_EARLY_SQUAD_DEFAULT : 0 };
    static __device__ __forceinline__ void init(const SymArgs &a, Ctx &c, double *extra_lds)
    {
        c.hc = hey_consts();
    }
    static __device__ __forceinline__ void load(const SymArgs &a, size_t i, int slot, Ctx &c, double &norm)
    {
        HeyPoint &pt = c.pt;
        pt.s = uni(a.s[i]);
        rim_sincos(a.theta[i], &pt.sin_th, &pt.cos_th);
        pt.sin_th = uni(pt.sin_th);
        pt.cos_th = uni(pt.cos_th);
        hey_point_derive(pt);
        pt.sigma0 = uni(pt.sigma0); pt.sigma0_sq = uni(pt.sigma0_sq); pt.dinv = uni(pt.dinv);
        pt.stokes = uni(c_slot_stokes[slot]);
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
    static __device__ __forceinline__ void begin(const Ctx &c, Task &T) { hey_begin(c.pt, T); }
    // what a task competes for early help with: the subintervals of the outer quadrature in progress (0 outside one).
    // A quadrature that converges stays below a few dozen; one on its way to GSL's limit of 4096 (heyvaerts.rs:82-83)
    // adds one per batch, for thousands of batches.
    static __device__ __forceinline__ int early_metric(const Task &T)
    { return (T.phase == HP_QAG_BISECT || T.phase == HP_QAG_FIRST) ? T.oq.size : 0; }
    static __device__ __forceinline__ void uniformize(Task &T) { hey_uniformize(T); }
    static __device__ __forceinline__ bool done(const Task &T) { return T.stage == HS_DONE; }
    typedef HeyStash Stash;
    // rounds (heyvaerts_wave.h): a long outer quadrature's batch carries the children of up to RIM_TURBO_MAX intervals.
    // Compiled in for the pitchy-kappa distribution only: that is where they pay (four quadratures of 2844 .. 4132 batches
    // and a dozen of 300 .. 760 in 65536 rows: -20 .. -26 % on its Faraday launches); the power-law table's long
    // quadratures are hidden by the squad already, the other two tables have none -- and the code costs a kernel that
    // never sees a round about 1 % (profiles/r4_ab_rounds.txt).
    enum { TURBO = KIND == DIST_PITCHY_KAPPA ? 1 : 0 };
    static __device__ __forceinline__ void post(const Ctx &c, const GKLane &g, const IStore &outer, Task &T, SymBatch &B, Stash *hs, int rounds)
    { hey_post(c.pt, g, outer, T, B, hs, rounds); }
    static __device__ __forceinline__ int round_n(const Stash &hs) { return uni(hs.round_n); }
    static __device__ __forceinline__ void round_drop(Stash &hs) { hs.round_n = 1; hs.hit = -1; }
    static __device__ __forceinline__ double round_request(const Stash &hs, const GKLane &g, int j) { return hey_round_request(&hs, g, j); }
    // file the sums of the round's further intervals: lane `rank`-th request of interval j sits at 62 j + rank of the slot
    static __device__ __forceinline__ void eval2(const Ctx &c, const GKLane &g, const IStore &inner, Park *qp,
                                                 double x0, int tag0, double x1, int tag1, bool have1,
                                                 double &v0, int &st0, double &v1, int &st1)
    { hey_eval_pair<KIND>(c.pt, c.d, c.hc, g, inner, qp, x0, tag0, x1, tag1, have1, v0, st0, v1, st1); }
    static __device__ __forceinline__ void consume(const Ctx &c, const GKLane &g, const IStore &outer, Task &T,
                                                   const SymBatch &B, double gval, int bst, Stash *hs, unsigned long long *stash_samples,
                                                   unsigned long long *dropped_samples, const AssistSlot *slot, int rank, int per)
    {
        // (a round's further intervals: their values, status bits and sample counts are read from the owner's board slot)
        HeyRoundIO io;
        io.res = slot->res; io.res_status = slot->res_status; io.res_samples = slot->res_samples; io.rank = rank; io.per = per;
        hey_consume(c.pt, g, outer, T, B, gval, bst, hs, stash_samples, dropped_samples, &io);
    }
    static __device__ __forceinline__ double result(const Ctx &, const Task &T, int &st) { return hey_result(T, st); }
};

template <class P>
__global__ __launch_bounds__(64, P::WAVES) void coop_kernel(SymArgs a)
{
#if defined(RIM_PROF) && defined(__HIP_DEVICE_COMPILE__)
    if (threadIdx.x < 32) rim_prof_lds[threadIdx.x] = 0;
    __syncthreads();
#endif
    RIM_PROF_T(t_kernel);
    __shared__ double s_tab[96];
    __shared__ double s_inner[RIM_ISTORE_DOUBLES(CAP_INNER)];
    __shared__ double s_outer[RIM_ISTORE_DOUBLES(CAP_OUTER)];
    __shared__ typename P::Task s_park;
    const GKLane g = gk_lane_init(s_tab);
    const int lane = g.lane;
    double *spill = a.spill + (size_t) blockIdx.x * SPILL_DOUBLES_PER_WAVE;
    const IStore inner = istore_carve(s_inner, CAP_INNER, spill, SPILL_INNER);
    const IStore outer = istore_carve(s_outer, CAP_OUTER, spill + RIM_ISTORE_DOUBLES(SPILL_INNER), SPILL_OUTER);
    __shared__ typename P::Park s_qpark;
    __shared__ double s_extra[P::EXTRA_LDS_DOUBLES];
    __shared__ typename P::Stash s_stash;       // rounds of a long outer quadrature (Faraday kernel; heyvaerts_wave.h)
    if (threadIdx.x == 0) { s_qpark.ctr = WaveCounters{0, 0, 0}; s_qpark.hb = nullptr; s_stash.used = 0; P::round_drop(s_stash); }
    bool last_shared = false;                   // the previous batch of the own task went over the board: the next may carry a round
    int round_n = 1;                            // intervals whose children the current batch evaluates

    AssistSlot *const my = a.board + blockIdx.x;
    unsigned *const flag_exhausted = a.board_flags + BOARD_FLAG_EXHAUSTED;
    unsigned *const flag_active = a.board_flags + BOARD_FLAG_ACTIVE;
    unsigned *const flag_idle = a.board_flags + BOARD_FLAG_IDLE;
    unsigned *const hints = a.board_flags + BOARD_HINTS;
    const unsigned nboard = gridDim.x;
    unsigned seq = 0;                      // sequence number of this wave's published batches
    bool counted_idle = false;             // this wave is currently counted in flags[IDLE]
    int backoff = 1;
    unsigned long long idle_since = 0;     // wall clock of the first empty poll since this wave last evaluated a request
    // diagnostics of the cooperative tail (queue words 8..13)
    unsigned n_polls = 0;
#if defined(RIM_COOP_DIAG)
    unsigned long long n_shared_batches = 0, n_helper_reqs = 0, n_owner_shared_reqs = 0, wait_ticks = 0,
                       n_empty_claims = 0, eval_ticks = 0, max_wait = 0, n_polls_total = 0;
#define COOP_DIAG(x) x
#else
#define COOP_DIAG(x)
#endif

    const unsigned long long ntasks = (unsigned long long) a.n * (unsigned long long) a.nslots;
    typename P::Ctx cx;                    // context of the requests being evaluated (own task or a helped one)
    P::init(a, cx, s_extra);
    unsigned long long *const queue = a.queue + P::QUEUE;
    // The own task's state lives in LDS (s_park) between the three places that touch it, so that it
    // never occupies registers while the integrand runs.
    size_t own_i = 0;
    int own_slot = 0;
    bool have_task = false, helper = false;
    bool board_dead = false;               // this wave once gave up waiting for helpers: it never publishes again (below)
    unsigned last_hint = 0;                // lane 0: the hint whose batch this wave has already seen exhausted

    // early help for the launch's longest chain (coop_common.h, SymArgs::early_squad)
    // (early_classes titles, each with its own 64 waves of the squad -- the ones that listen to the hint lines l with
    // l & (classes - 1) == title.  A task competes for one title at a time, starting with block & (classes - 1); when a
    // longer quadrature holds that one it moves to the title with the shortest holder, so that the titles end up with
    // the `classes` longest quadratures in progress and that many chains are served side by side.  Measured and not kept:
    // every champion publishing to the whole squad, eight titles on 256 waves -- the hint traffic slowed the bulk.)
    unsigned title = (unsigned) blockIdx.x & (a.early_classes - 1u);
    unsigned *const flag_champ0 = a.board_flags + BOARD_FLAG_CHAMP;
    const bool early_on = P::EARLY_HELP && a.early_squad != 0u && a.board != nullptr;
    const bool squad = early_on && (unsigned) blockIdx.x % a.early_stride == 0u && (unsigned) blockIdx.x / a.early_stride < a.early_squad;
    unsigned champ_mine = 0;               // the word this wave last entered in flags[CHAMP] (0: none)
    bool champion = false;                 // ... and it was the maximum: this task publishes while the queue is full
    bool seen_exhausted = false;           // (squad) the queue has run dry: from here on an ordinary helper of the tail
#if defined(RIM_TAIL_DIAG)
    unsigned long long task_t0 = 0;
    unsigned n_champ_batches = 0;
#endif

    // issue priorities: 3 the champion and the squad that serves it (the chain is the launch's critical path), 2 every
    // other owner, 0 the helpers of the tail
    __builtin_amdgcn_s_setprio(2);
    for (;;) {
        SymBatch B;
        B.req_n = 0.; B.req_lobe = 0; B.req_active = false; B.n_req = 0; B.phase = PH_DONE;
        AssistSlot *src = my;
        unsigned src_seq = 0;
        unsigned long long mask = 0;
        bool shared = false;

        if (!helper) {
            // ---------- owner: next batch of the current task (fetching a task first if needed) ----------
            if (!have_task) {
                // (a wave of the squad never fetches a task: it becomes a helper right here, without touching the queue --
                // the host left it out of flags[ACTIVE])
                const unsigned long long t = squad ? ~0ull : wave_next_task(queue, lane);
                if (t >= ntasks) {
                    helper = true;
                    if (lane == 0) {
                        if (!squad) {
#if defined(RIM_TAIL_DIAG)     // (tools/tail_times.py: when did the queue run dry, when did the launch end)
                            atomicCAS(a.queue + (P::QUEUE ? 10 : 12), 0ull, wall_clock64());
#endif
                            __hip_atomic_store(flag_exhausted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_fetch_sub(flag_active, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (a.board) __hip_atomic_fetch_add(flag_idle, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    counted_idle = true;      // flags[IDLE] = helper waves that are not evaluating a request
                    // Helpers run below the issue priority of waves that own a task: an owner's serial
                    // bookkeeping between batches is the critical path of the tail.  (The squad keeps the champion's
                    // priority until the queue is dry.)
                    if (squad) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
                    if (!a.board) break;     // cooperation disabled
                    continue;
                }
                const size_t seqidx = (size_t) (t / (unsigned) a.nslots);
                own_i = a.perm ? (size_t) a.perm[seqidx] : seqidx;
                own_slot = a.slot[(int) (t % (unsigned) a.nslots)];
                double norm;
                P::load(a, own_i, own_slot, cx, norm);
                if (lane == 0) {
                    s_qpark.hb = (a.heartbeat && (t | (unsigned long long) P::HB_TAG) == a.hb_task) ? a.heartbeat : nullptr;
                    if (s_qpark.hb) hb_store(s_qpark.hb + 0, t + 1);
                }
                __syncthreads();
                if (!(norm == norm)) {
                    if (lane == 0) {
                        a.out[own_i * 8 + own_slot] = RIM_NAN;
                        if (a.status) a.status[own_i * 8 + own_slot] = ST_NORM_FAIL | ST_NONFINITE;
                    }
                    continue;
                }
                typename P::Task T0;
                P::begin(cx, T0);
                if (lane == 0) { s_park = T0; s_stash.used = 0; }
                __syncthreads();
                have_task = true;
                last_shared = false;
#if defined(RIM_TAIL_DIAG)
                task_t0 = wall_clock64();
#endif
            }
            bool finished;
            int task_batches = 0;            // (the task's early-help metric after this post)
            {
                typename P::Task T = s_park;
                P::uniformize(T);
                if (!P::done(T)) P::post(cx, g, outer, T, B, P::TURBO ? &s_stash : nullptr, (P::TURBO && a.turbo && last_shared && !board_dead) ? RIM_TURBO_MAX - 1 : 0);
                finished = P::done(T);
                task_batches = P::early_metric(T);
                round_n = P::TURBO ? P::round_n(s_stash) : 1;
                if (finished) {
                    int st = 0;
                    const double val = P::result(cx, T, st);
                    // the heaviest task of the launch: its sequential chain of batches bounds the launch's tail
                    // (queue word 15 of the kernel's block: (batches << 40) | point index; rimphony_last_tail)
                    if (lane == 0) atomicMax(a.queue + (P::QUEUE ? 15 : 14), ((unsigned long long) T.batches << 40) | ((unsigned long long) own_i & 0xffffffffffull));
                    if (lane == 0) {
                        a.out[own_i * 8 + own_slot] = val;
#if defined(RIM_TAIL_DIAG)     // (tools/tail_times.py: when a chain of >= 512 batches began and ended -- the last one to end is reported)
                        if (P::QUEUE && T.batches >= 2048) { a.queue[8] = task_t0; a.queue[9] = wall_clock64(); a.queue[14] = n_champ_batches; }
                        n_champ_batches = 0;
#endif
#if defined(RIM_TAIL_DIAG)     // (tools/tail_times.py: the task's number of batches in the upper half of the status word)
                        if (a.status) a.status[own_i * 8 + own_slot] = st | ((T.batches < 0x7fff ? T.batches : 0x7fff) << 16);
#else
                        if (a.status) a.status[own_i * 8 + own_slot] = st;
#endif
                        if (s_qpark.hb) hb_store(s_qpark.hb + 10, 1ull);
                    }
                } else {
                    __syncthreads();              // everyone has read s_park
                    if (lane == 0) s_park = T;    // state after posting (batch counter, picked interval)
                }
            }
            if (finished) {
                if (champ_mine) {
                    // give the title back (if it is still this task's): the other candidates enter again with their next batch
                    if (lane == 0) {
                        unsigned expect = champ_mine;
                        __hip_atomic_compare_exchange_strong(flag_champ0 + 32u * title, &expect, 0u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT);
                    }
                    champ_mine = 0;
                    if (champion) __builtin_amdgcn_s_setprio(2);
                    champion = false;
                }
                have_task = false;
                continue;
            }
            mask = wv_ballot(B.req_active);

            // publish the batch when some wave is idle -- or, while the queue is still full, when this task holds the
            // most batches of all tasks in flight (the squad serves it)
            unsigned idle = 0, act = 1, exhausted = 0, champ_now = 0, title_next = title;
            const unsigned cand = (early_on && !board_dead && task_batches >= a.early_min)
                ? (((unsigned) (task_batches < 0xffff ? task_batches : 0xffff) << 16) | ((unsigned) blockIdx.x + 1u)) : 0u;
            if (lane == 0 && a.board) {
                exhausted = __hip_atomic_load(flag_exhausted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (exhausted) {
                    idle = __hip_atomic_load(flag_idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    act = __hip_atomic_load(flag_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (cand | champ_mine) {
                    if (cand < champ_mine) {
                        // the quadrature that competed for the title is over: give it back if it is still this task's
                        unsigned expect = champ_mine;
                        __hip_atomic_compare_exchange_strong(flag_champ0 + 32u * title, &expect, 0u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (cand) {
                        champ_now = __hip_atomic_load(flag_champ0 + 32u * title, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (champ_now > cand) {
                            // a longer quadrature holds this title (this task's own entry, if any, is gone with it): compete
                            // for the title whose holder is the shortest
                            for (unsigned k = 0; k < a.early_classes; k++) {
                                const unsigned v = __hip_atomic_load(flag_champ0 + 32u * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (v < champ_now) { champ_now = v; title_next = k; }
                            }
                        }
                        if (champ_now < cand) {
                            const unsigned old = __hip_atomic_fetch_max(flag_champ0 + 32u * title_next, cand, __ATOMIC_RELAXED,
                                                                        __HIP_MEMORY_SCOPE_AGENT);
                            champ_now = old > cand ? old : cand;
                        }
                    }
                }
            }
            title = (unsigned) __builtin_amdgcn_readfirstlane((int) title_next);
            idle = (unsigned) __builtin_amdgcn_readfirstlane((int) idle);
            act = (unsigned) __builtin_amdgcn_readfirstlane((int) act);
            exhausted = (unsigned) __builtin_amdgcn_readfirstlane((int) exhausted);
            champ_now = (unsigned) __builtin_amdgcn_readfirstlane((int) champ_now);
            if (early_on) {
                champ_mine = cand;
                const bool is_champ = cand != 0u && !exhausted && champ_now == cand;
                if (is_champ != champion) { if (is_champ) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
                champion = is_champ;
            }
            const int cnt = __builtin_popcountll(mask);
            shared = (exhausted ? idle != 0 : champion) && cnt >= 2 && !board_dead;
            if (P::TURBO) {
                if (cnt != 0) last_shared = shared;     // (a bisection booked from the stash posts nothing and changes nothing)
                if (round_n > 1 && !shared) {
                    // the round was planned for a batch that does not go over the board after all: the picked interval only
                    __syncthreads();
                    if (lane == 0) P::round_drop(s_stash);
                    __syncthreads();
                    round_n = 1;
                }
            }
#if defined(RIM_TAIL_DIAG)     // (tools/tail_times.py: when the longest outer quadrature passed 64 / 512 / 2048 subintervals, and how
                               // many of its batches were published as the champion's)
            if (P::QUEUE && lane == 0) {
                if (task_batches == 64) atomicCAS(a.queue + 12, 0ull, wall_clock64());
                if (task_batches == 512) atomicCAS(a.queue + 13, 0ull, wall_clock64());
                if (task_batches == 2048) a.queue[2] = wall_clock64();
            }
            if (shared && !exhausted) n_champ_batches += 1;
#endif
            if (shared) {
                seq += 1;
                src_seq = seq;
                COOP_DIAG(n_shared_batches += 1;)
                if (P::TURBO && round_n > 1) __syncthreads();      // lane 0's copy of the task state (the round's intervals) is visible
                if (B.req_active) {
                    const int rank = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                    bput(&my->req_n[rank], rim_bits(B.req_n));
                    bput(&my->req_lobe[rank], B.req_lobe);
                    // a round: the children of the further intervals, `cnt` requests each, behind the picked interval's
                    for (int j = 1; j < round_n; j++) {
                        bput(&my->req_n[cnt * j + rank], rim_bits(P::round_request(s_stash, g, j)));
                        bput(&my->req_lobe[cnt * j + rank], B.req_lobe);
                    }
                }
                if (lane == 0) {
                    bput(&my->point, (unsigned long long) own_i);
                    bput(&my->slot, own_slot);
                    bput(&my->done, 0u);
                }
                drain_vmem();
                __syncthreads();
                if (lane == 0)
                    __hip_atomic_store(&my->claim, ((unsigned long long) seq << 32) | ((unsigned long long) (cnt * round_n) << 8),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                drain_vmem();          // the claim word is out before anybody can see the hint
                __syncthreads();
                {
                    // (while the queue is full the only listeners are the squad: every hint line of the title)
                    const unsigned span = exhausted ? hint_span(act) : a.early_classes;
                    const unsigned channel = exhausted ? (((unsigned) blockIdx.x + seq) & (span - 1u)) : title;
                    if (((unsigned) lane & (span - 1u)) == channel)
                        __hip_atomic_store(&hints[(unsigned) lane * BOARD_HINT_STRIDE], (seq << 16) | ((unsigned) blockIdx.x + 1u),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        } else {
            // ---------- helper: find a published batch through this wave's hint line ----------
            unsigned h = 0, act = 1, exh = 0;
            int leave = 0;
            unsigned long long c = 0;
            if (lane == 0) {
                if ((n_polls & 15u) == 0) {
                    act = __hip_atomic_load(flag_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (squad && !seen_exhausted) exh = __hip_atomic_load(flag_exhausted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // More idle waves than the remaining owners can feed (a batch has <= 62 requests) only
                    // add polling traffic, which slows the waves that compute: the surplus leaves.
                    const unsigned keep = act * 64u + 32u;
                    if (counted_idle && act != 0 &&
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
                    c = __hip_atomic_load(&a.board[hs - 1].claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!claim_open(c)) { last_hint = h; h = 0; }      // nothing left of that batch: wait for a new hint
                } else {
                    h = 0;
                }
            }
            h = (unsigned) __builtin_amdgcn_readfirstlane((int) h);
            act = (unsigned) __builtin_amdgcn_readfirstlane((int) act);
            if (__builtin_amdgcn_readfirstlane(leave)) break;
            if (squad && !seen_exhausted && __builtin_amdgcn_readfirstlane((int) exh)) {
                seen_exhausted = true;
                __builtin_amdgcn_s_setprio(0);
            }
            n_polls += 1;
            COOP_DIAG(n_polls_total += 1;)
            if (h == 0) {
                if (act == 0) break;       // every task is finished
                if (squad && !seen_exhausted) idle_since = 0;      // the squad waits for a champion as long as the queue is full
                // A wave that has seen nothing to do for 2 s leaves.  It holds no claim, so leaving is always safe,
                // and it bounds every wait in this kernel: should part of the grid not be resident (the launch
                // sizes it so that it is), the waves waiting for a slot get one instead of being waited for.
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
            src = a.board + (h - 1u);
            src_seq = (unsigned) (cw >> 32);
            shared = true;
            // context of the helped task is loaded after the first successful claim (below)
        }

        // ---------- evaluate requests of `src` (the integrand lives here, once) ----------
        int batch_status = 0;
        double gval = 0.;
        bool ctx_loaded = !helper;
        unsigned long long local_mask = mask;
        int got = 0;
        bool redo;
        do {
        redo = false;
        for (;;) {
            int k;
            if (shared) {
                k = assist_claim(src, src_seq, lane);      // ordinal of the request within the batch
            } else {
                k = local_mask ? __builtin_ffsll((long long) local_mask) - 1 : -1;
                local_mask &= local_mask - 1;
            }
            if (k < 0) break;
            got += 1;
            COOP_DIAG(if (shared) { if (helper) n_helper_reqs += 1; else n_owner_shared_reqs += 1; })
            double n;
            int lb;
            if (helper) {
                if (!ctx_loaded) {
                    double norm;
                    const size_t hi = (size_t) bcast_u64(bget(&src->point));
                    const int hs = __builtin_amdgcn_readfirstlane(bget(&src->slot));
                    P::load(a, hi, hs, cx, norm);
                    ctx_loaded = true;
                    if (counted_idle) {
                        if (lane == 0) __hip_atomic_fetch_sub(flag_idle, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        counted_idle = false;
                    }
                    backoff = 1;
                    idle_since = 0;
                }
                n = uni(rim_frombits(bget(&src->req_n[k])));
                lb = __builtin_amdgcn_readfirstlane(bget(&src->req_lobe[k]));
            } else if (P::TURBO && shared && k >= __builtin_popcountll(mask)) {
                // a request of one of the round's further intervals: it only exists on the board
                n = uni(rim_frombits(bget(&my->req_n[k])));
                lb = __builtin_amdgcn_readfirstlane(bget(&my->req_lobe[k]));
            } else {
                const int kl = shared ? kth_set_bit(mask, k) : k;     // lane that posted the request
                n = readlane_d(B.req_n, kl);
                lb = wv_readlane(B.req_lobe, kl);
            }
            // a wave working through its own batch starts two requests together (P::eval2)
            int k2 = -1;
            double n2 = n;
            int lb2 = lb;
            if (!shared && local_mask) {
                k2 = __builtin_ffsll((long long) local_mask) - 1;
                local_mask &= local_mask - 1;
                n2 = readlane_d(B.req_n, k2);
                lb2 = wv_readlane(B.req_lobe, k2);
            }
            if (lane == 0 && s_qpark.hb) {
                hb_store(s_qpark.hb + 8, (unsigned long long) k);
                hb_store(s_qpark.hb + 9, rim_bits(n));
            }
            int st = 0, st2 = 0;
            double val, val2;
            COOP_DIAG(const unsigned long long e0 = wall_clock64();)
            RIM_PROF_T(t_req);
            unsigned long long samples_before = 0;
            if ((a.work || P::TURBO) && lane == 0) samples_before = s_qpark.ctr.samples;     // (rounds book a request's samples too)
            P::eval2(cx, g, inner, &s_qpark, n, lb, n2, lb2, k2 >= 0, val, st, val2, st2);
            // Work counters count what went INTO the stored value: a request evaluated for the board is booked by
            // the owner when it reads the result, so a batch the owner gives up on and evaluates again is counted once.
            unsigned long long samples_taken = 0;
            if ((a.work || P::TURBO) && lane == 0) {
                samples_taken = s_qpark.ctr.samples - samples_before;
                if (a.work && !shared) atomicAdd(a.work + own_i * 8 + (size_t) own_slot, samples_taken);
            }
            RIM_PROF_ADD(9, t_req);
            COOP_DIAG(eval_ticks += wall_clock64() - e0;)
            if (shared) {
                if (lane == 0) {
                    bput(&src->res[k], rim_bits(val));
                    bput(&src->res_status[k], st);
                    if (a.work || P::TURBO) bput(&src->res_samples[k], (unsigned) samples_taken);
                    drain_vmem();
                    __hip_atomic_fetch_add(&src->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (lane == k) gval = val;
                if (lane == k2) gval = val2;
                batch_status |= st | st2;
            }
        }
        if (helper) break;

        // ---------- owner: collect a shared batch ----------
        if (shared) {
            const unsigned want = (unsigned) (__builtin_popcountll(mask) * round_n);
            const int rank = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            bool complete = false;
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                unsigned dn = 0;
                if (lane == 0) dn = __hip_atomic_load(&my->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dn = (unsigned) __builtin_amdgcn_readfirstlane((int) dn);
                if (dn >= want) {
                    complete = true;
                    COOP_DIAG(const unsigned long long w = wall_clock64() - t0; wait_ticks += w; if (w > max_wait) max_wait = w;)
                    break;
                }
                if (wall_clock64() - t0 > a.owner_ticks) break;
                __builtin_amdgcn_s_sleep(32);
            }
            if (lane == 0)
                __hip_atomic_store(&my->claim, (unsigned long long) seq << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (complete) {
                int stl = 0;
                if ((mask >> lane) & 1ull) {
                    gval = rim_frombits(bget(&my->res[rank]));
                    stl = bget(&my->res_status[rank]);
                    if (a.work) atomicAdd(a.work + own_i * 8 + (size_t) own_slot, (unsigned long long) bget(&my->res_samples[rank]));
                }
                if (wv_ballot((stl & ST_INNER_FAIL) != 0)) batch_status |= ST_INNER_FAIL;
                if (wv_ballot((stl & ST_STORE_FULL) != 0)) batch_status |= ST_STORE_FULL;
            } else {
                // A claimed request has not come back within the bound (a helper wave that is not running: the GPU is
                // shared, or part of the grid is not resident).  The value of a request does not depend on who
                // evaluates it, so the owner closes the batch and evaluates ALL of it itself -- same bits, only later --
                // and never publishes again: whatever a late helper still writes to this slot is never read.
                board_dead = true;
                shared = false;
                if (P::TURBO && round_n > 1) {
                    __syncthreads();
                    if (lane == 0) P::round_drop(s_stash);
                    __syncthreads();
                    round_n = 1;
                }
                local_mask = mask;
                batch_status = 0;
                gval = 0.;
                redo = true;
            }
        }
        } while (redo);
        if (helper) {
            if (got == 0) {
                // every request of that batch was already taken: back off before looking again
                // (last_hint is not set here: the claim word is re-read on the next poll, and that
                // poll files the hint away once the batch shows no open request)
                COOP_DIAG(n_empty_claims += 1;)
                for (int w = 0; w < backoff; w++) __builtin_amdgcn_s_sleep(127);
                if (backoff < 16) backoff *= 2;
            } else {
                if (lane == 0) __hip_atomic_fetch_add(flag_idle, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                counted_idle = true;
            }
            continue;
        }
        __syncthreads();
        {
            typename P::Task T = s_park;
            P::uniformize(T);
            unsigned long long stash_samples = 0, dropped_samples = 0;
            P::consume(cx, g, outer, T, B, gval, uni(batch_status), P::TURBO ? &s_stash : nullptr, &stash_samples, &dropped_samples, my,
                       __builtin_popcountll(mask & ((1ull << lane) - 1ull)), __builtin_popcountll(mask));
            if (P::TURBO) {
                // (work counters count what went INTO the stored value: a bisection booked from the stash brings its samples)
                if (a.work && lane == 0 && stash_samples) atomicAdd(a.work + own_i * 8 + (size_t) own_slot, stash_samples);
                // the launch's sample count is the reference's: what was evaluated ahead and never asked for comes off it
                // (counted by whichever wave evaluated it; the counters are summed modulo 2^64 at the end of the launch)
                if (dropped_samples && lane == 0) s_qpark.ctr.samples -= dropped_samples;
            }
            __syncthreads();
            if (lane == 0) s_park = T;
            __syncthreads();
        }
    }

    RIM_PROF_ADD(0, t_kernel);
#if defined(RIM_TAIL_DIAG)
    if (g.lane == 0) atomicMax(a.queue + (P::QUEUE ? 11 : 13), wall_clock64());
#endif
    __syncthreads();
#if defined(RIM_PROF) && defined(__HIP_DEVICE_COMPILE__)
    if (threadIdx.x < 32) g_rim_prof[(size_t) blockIdx.x * 32 + threadIdx.x] += rim_prof_lds[threadIdx.x];
#endif
    if (g.lane == 0) {
        atomicAdd(queue + 1, s_qpark.ctr.samples);
        atomicAdd(queue + 2, s_qpark.ctr.steps);
        atomicAdd(queue + 3, s_qpark.ctr.inner_qags);
#if defined(RIM_COOP_DIAG)
        atomicAdd(a.queue + 8, n_shared_batches);
        atomicAdd(a.queue + 9, n_helper_reqs);
        atomicAdd(a.queue + 10, n_owner_shared_reqs);
        atomicAdd(a.queue + 11, wait_ticks);
        atomicAdd(a.queue + 12, n_polls_total);
        atomicAdd(a.queue + 13, n_empty_claims);
        atomicAdd(a.queue + 14, eval_ticks);
        (void) max_wait;          // [15] is the heaviest task: (batches << 40) | point index
#endif
    }
}

__global__ void board_init_kernel(unsigned *flags, unsigned active)
{
    const unsigned i = threadIdx.x;
    if (blockIdx.x == 0)
        for (unsigned k = i; k < BOARD_FLAG_WORDS; k += blockDim.x) flags[k] = (k == BOARD_FLAG_ACTIVE) ? active : 0u;
}

// ------------------------------------------------------------------------------
// heyvaerts (Faraday rho_Q = slot 6, rho_V = slot 7)
// ------------------------------------------------------------------------------

// ------------------------------------------------------------------------------
// expensive-first task order.  The cost of a point grows with s (about 3x from s < 10 to
// s > 100 on the bench tables), so points are bucketed by the binary exponent of s and the
// persistent grid starts with the large-s buckets: the cheap points then fill the tail of the
// launch.  Only the order of evaluation changes, never a result.
// ------------------------------------------------------------------------------
#define ORDER_BUCKETS 32

// The Faraday kernel's tasks have their own cost profile (per-coefficient batch counts of a -DRIM_TAIL_DIAG build, round 4,
// all four tables): the fewer harmonics -- the smaller sigma0 = s sin(theta) -- the MORE batches (mean 37 at the top decile
// of s, 67 at the bottom one; 200+ at sigma0 ~ 0.005), and the outer quadratures that run to GSL's limit of 4096
// bisections (heyvaerts.rs:82-83; one task in ~1e5, thousands of sequential batches) all sat at sigma0 in [1.15, 2.4]
// (power law: 2.15, 2.40; pitchy kappa: 1.15, 1.20, 1.41).  With the Symphony order (large s first) exactly these tasks
// started last.  Here: the window 0.9 <= sigma0 <= 3 first (10 % of the rows; a chain that starts with the launch has
// the whole launch to overlap with -- coop_common.h, early help), then ascending sigma0.
__device__ __forceinline__ int faraday_order_bucket(double s, double theta)
{
    double sn, cs;
    rim_sincos(theta, &sn, &cs);
    const double sigma0 = s * rim_fabs(sn);
    if (sigma0 >= 0.9 && sigma0 <= 3.) return 0;
    if (!(sigma0 > 0.)) return ORDER_BUCKETS - 1;
    const int e = (int) ((rim_bits(sigma0) >> 52) & 0x7ff) - 1023;     // floor(log2 sigma0) for normal sigma0
    const int b = e + 11;                                                // sigma0 < 2^-10 -> 1, ..., sigma0 >= 2^19 -> 31
    return b < 1 ? 1 : (b > ORDER_BUCKETS - 1 ? ORDER_BUCKETS - 1 : b);
}

__device__ __forceinline__ int order_bucket(double s, double theta, const double *gmin, const double *temp, size_t i, int faraday)
{
    if (faraday) return faraday_order_bucket(s, theta);
    // Thermal distribution: the long tasks are the COLD, LOW-FREQUENCY, SMALL-ANGLE points, not the large-s ones -- there
    // j_V / alpha_V march through ~300 batches of the tail integral before GSL's round-off detector ends them (NaN, as in
    // the reference), ten times the table's median.  Measured on 65536 rows of configs[2]'s table with the per-coefficient
    // batch counts of a -DRIM_TAIL_DIAG build (round 4): s sin(theta) T^2 < 1 holds for 18 % of the rows and 95 % of the
    // coefficients of 200 batches and more (< 0.01: 3 % of the rows, 43 % of them, 92 % pure).  They go first; with the
    // large-s-first order they started last and were 12 % of a 65536-row launch of the Symphony groups.
    if (temp) {
        double sn, cs;
        rim_sincos(theta, &sn, &cs);
        const double t = temp[i];
        const double key = s * rim_fabs(sn) * (t * t);
        if (key < 0.01) return 0;
        if (key < 0.1) return 1;
        if (key < 1.) return 2;
        const int e = (int) ((rim_bits(s) >> 52) & 0x7ff) - 1023;
        const int b = 17 - e;
        return b < 3 ? 3 : (b > ORDER_BUCKETS - 1 ? ORDER_BUCKETS - 1 : b);
    }
    // Points whose 30 discrete harmonics all lie below gamma_min (power-law families) start the n
    // integration from zero and must resolve the onset at gamma_min inside one wide chunk: they are
    // 100-1000x the mean cost (and usually end in a GSL round-off failure, as in the reference).
    // They go first.
    if (gmin) {
        double sn, cs;
        rim_sincos(theta, &sn, &cs);
        const double n = rim_floor(s * rim_fabs(sn) + 30.);
        const double nos = n / s;
        const double gp = (nos + rim_fabs(cs) * rim_sqrt(nos * nos - sn * sn)) / (sn * sn);
        if (gp < gmin[i]) return 0;
    }
    const int e = (int) ((rim_bits(s) >> 52) & 0x7ff) - 1023;     // floor(log2 s) for normal s
    int b = 17 - e;                                                 // large s -> small bucket index
    return b < 1 ? 1 : (b > ORDER_BUCKETS - 1 ? ORDER_BUCKETS - 1 : b);
}

__global__ void order_hist_kernel(const double *s, const double *theta, const double *gmin, const double *temp, size_t n, unsigned *hist,
                                  int faraday)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&hist[order_bucket(s[i], theta[i], gmin, temp, i, faraday)], 1u);
}

__global__ void order_scan_kernel(unsigned *hist)   // exclusive scan in place, one thread
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned acc = 0;
        for (int b = 0; b < ORDER_BUCKETS; b++) { const unsigned c = hist[b]; hist[b] = acc; acc += c; }
    }
}

__global__ void order_scatter_kernel(const double *s, const double *theta, const double *gmin, const double *temp, size_t n,
                                     unsigned *offsets, unsigned *perm, int faraday)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[atomicAdd(&offsets[order_bucket(s[i], theta[i], gmin, temp, i, faraday)], 1u)] = (unsigned) i;
}

// fills the slots that were not selected (or not yet available) with NaN
__global__ void fill_unselected_kernel(double *out, int32_t *status, size_t n, uint32_t computed_mask)
{
    const size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * 8) return;
    const int slot = (int) (idx & 7);
    if (!(computed_mask & (1u << slot))) {
        out[idx] = RIM_NAN;
        if (status) status[idx] = ST_NONFINITE | RIMPHONY_ST_NOT_COMPUTED;
    }
}

// ------------------------------------------------------------------------------
// unit seams
// ------------------------------------------------------------------------------

__global__ void bessel_kernel(size_t count, const double *n, const double *x, double *j, double *dj)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (j) j[i] = bessel_j(n[i], x[i]);
    if (dj) dj[i] = bessel_dj(n[i], x[i]);
}


template <int KIND>
__global__ __launch_bounds__(64) void gamma_integral_kernel(PointArgs pa, const double *norm_ptr, size_t count,
                                                            const double *nvals, double *out, double *spill_base)
{
    __shared__ double s_tab[96];
    __shared__ double s_inner[RIM_ISTORE_DOUBLES(CAP_INNER)];
    const GKLane g = gk_lane_init(s_tab);
    const IStore inner = istore_carve(s_inner, CAP_INNER, spill_base + (size_t) blockIdx.x * SPILL_DOUBLES_PER_WAVE, SPILL_INNER);
    __shared__ QagPark s_qpark;
    if (threadIdx.x == 0) { s_qpark.ctr = WaveCounters{0, 0, 0}; s_qpark.hb = nullptr; }
    SymPoint pt;
    pt.s = pa.s;
    rim_sincos(pa.theta, &pt.sin_th, &pt.cos_th);
    pt.coeff = pa.coeff;
    pt.stokes = pa.stokes;
    DistParams d;
    for (int k = 0; k < 5; k++) d.par[k] = pa.par[k];
    dist_prepare<KIND>(d, norm_ptr[0]);
    for (size_t i = blockIdx.x; i < count; i += gridDim.x) {
        const double n = nvals[i];
        LeungOrder ord_tmp[2];
        SymOrder so = sym_order(n, ord_tmp);
        __syncthreads();
        if (g.lane == 0) { s_qpark.ord[0] = ord_tmp[0]; s_qpark.ord[1] = ord_tmp[1]; }
        __syncthreads();
        so.o = s_qpark.ord;
        const GammaLimits L = gamma_limits(pt, n, pa.negative_lobe);
        auto f = [&](double x, bool active) -> double { return active ? gamma_integrand<KIND>(pt, d, so, x) : 0.; };
        QagState q;
        wave_qag(f, g, inner, L.g0, L.g1, 0., 1e-3, 5000, q, &s_qpark);
        if (g.lane == 0) out[i] = (q.status == QAG_SUCCESS) ? q.result : RIM_NAN;
    }
}

__device__ inline double selftest_integrand(int family, double p0, double p1, double x)
{
    switch (family) {
    case 0: { const double u = (x - p0) * p1; return 1. / (1. + u * u); }
    case 1: return rim_sqrt(rim_fabs(x - p0)) * p1;
    case 2: return x * x * (p0 + x * p1);
    case 3: return 1. / rim_sqrt(rim_fabs(x - p0) + p1);
    default: { const double u = x * p0; return p1 * rim_fabs((u - rim_floor(u)) - 0.5); }   // triangle wave: long lists
    }
}

__global__ __launch_bounds__(64) void qag_selftest_kernel(size_t count, const int32_t *family, const double *p0,
                                                          const double *p1, const double *a, const double *b,
                                                          double epsabs, double epsrel, int limit,
                                                          double *result, double *abserr, int32_t *qstatus, int32_t *size,
                                                          double *spill_base)
{
    __shared__ double s_tab[96];
    __shared__ double s_store[RIM_ISTORE_DOUBLES(CAP_NORM)];
    const GKLane g = gk_lane_init(s_tab);
    const IStore st = istore_carve(s_store, CAP_NORM, spill_base + (size_t) blockIdx.x * SPILL_DOUBLES_PER_WAVE, SPILL_INNER);
    __shared__ QagPark s_qpark;
    if (threadIdx.x == 0) { s_qpark.ctr = WaveCounters{0, 0, 0}; s_qpark.hb = nullptr; }
    // problems 2j and 2j + 1 run as a pair (wave_qag_pair: shared first rule application), as the
    // product kernels run their gamma-integrals; an odd last problem runs alone
    for (size_t i = 2 * (size_t) blockIdx.x; i < count; i += 2 * (size_t) gridDim.x) {
        const bool have1 = i + 1 < count;
        const size_t i1 = have1 ? i + 1 : i;
        const int fam0 = family[i], fam1 = family[i1];
        const double q00 = p0[i], q01 = p1[i], q10 = p0[i1], q11 = p1[i1];
        auto f = [&](double x, bool active, bool second) -> double {
            return active ? selftest_integrand(second ? fam1 : fam0, second ? q10 : q00, second ? q11 : q01, x) : 0.;
        };
        double r0, r1, ae[2];
        int s0, s1, sz[2];
        wave_qag_pair(f, g, st, a[i], b[i], a[i1], b[i1], have1, false, epsabs, epsrel, limit, &s_qpark, r0, s0, r1, s1, ae, sz);
        if (g.lane == 0) {
            result[i] = r0; abserr[i] = ae[0]; qstatus[i] = s0; size[i] = sz[0];
            if (have1) { result[i1] = r1; abserr[i1] = ae[1]; qstatus[i1] = s1; size[i1] = sz[1]; }
        }
    }
}

// ------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------

struct rimphony_ctx {
    int device;
    int n_cu;
    unsigned long long *d_queue;   // 4 words
    double *d_norm;
    unsigned *d_perm;               // [norm_cap] task order + ORDER_BUCKETS histogram words behind it
    double *d_spill;                // [spill_waves][SPILL_DOUBLES_PER_WAVE]
    size_t spill_waves;
    AssistSlot *d_board;            // [board_slots] + flags behind
    size_t board_slots;
    int no_assist;                  // RIMPHONY_NO_ASSIST=1: one wave per task to the end (A/B measurements)
    size_t norm_cap;
    // staging for the host-buffer entry point
    double *d_in;
    size_t in_cap;
    double *d_out;
    int32_t *d_status;
    size_t out_cap;
    // HIP events bracketing the most recent Symphony / Faraday launch
    hipEvent_t ev_start, ev_stop;   // Symphony launch
    int ev_valid;
    hipEvent_t ev_fstart, ev_fstop; // Faraday launch
    int evf_valid;
    // diagnostics: heartbeat words in host-mapped memory
    unsigned long long *hb_host;
    unsigned long long *hb_dev;
    unsigned long long hb_task;
    // resident workgroups per CU of coop_kernel<P>, by problem (0 Symphony, 1 Heyvaerts) and distribution kind:
    // queried once per context (= per device) and instantiation
    int resident[2][4];
    int resident_f32[4];            // ... and of the Symphony kernels' fp32-integrand variants
    // 0: this context has the GPU to itself (it holds the device's lock file); 1: another context or process had
    // the device first -- smaller persistent grids, no cooperative tail (see rimphony_ctx_create)
    int shared_mode;
    int lock_fd;
    unsigned long long ticks_per_s;  // wall_clock64 rate of the device
    unsigned long long owner_wait_ticks;   // how long an owner waits for its helpers before recomputing a batch itself
    // successive batch calls of a context share its workspace: each call's stream waits for the previous call's
    // work (ev_batch), and the calls themselves are serialised by mu
    hipEvent_t ev_batch;
    int ev_batch_valid;
    std::recursive_mutex *mu;
    // group kernel (rimphony_group.hip): per-wave spill regions and board, resident workgroups per CU by kind
    double *d_gspill;
    size_t gspill_waves;
    int gspill_faraday;             // the region is sized for the Faraday pair's longer outer lists
    GroupSlot *d_gboard;            // [gboard_slots] + flag words behind
    size_t gboard_slots;
    int resident_group[2][4];       // [Symphony groups / Faraday pair][kind]
    int f32_variant;                // RIMPHONY_F32_VARIANT=1: accept RIMPHONY_PRECISION_F32_INTEGRAND (measurement hook)
    int sym_solo;                   // RIMPHONY_SYM_SOLO=1: one wave per (point, coefficient), the round-2 kernel (A/B measurements)
    int faraday_symphony_order;     // RIMPHONY_FARADAY_ORDER=symphony: the Faraday launch visits the points in the Symphony order (A/B measurements)
    int rounds;                     // RIMPHONY_ROUNDS=0: a long outer quadrature evaluates one interval per batch, as until round 4 (A/B measurements)
    int early_squad;                // RIMPHONY_EARLY_SQUAD: waves of the Faraday grid that serve the longest chain from the start (0: off; -1: the kind's default)
    int early_min;                  // RIMPHONY_EARLY_MIN: batches after which a task competes for that help
    int faraday_group;              // RIMPHONY_FARADAY_GROUP=1: rho_Q and rho_V of a point in lock-step (measured slower: DESIGN.md section 5)
};

// ---- last error (thread-local text; the codes are in rimphony_hip.h) ---------------------------
static thread_local char t_last_error[256] = "";
void rim_set_last_error(const char *what, const char *detail)
{
    snprintf(t_last_error, sizeof t_last_error, "%s: %s", what ? what : "?", detail ? detail : "?");
}
extern "C" const char *rimphony_last_error(void) { return t_last_error; }
void rim_clear_last_error() { t_last_error[0] = 0; }

void rim_ctx_lock(rimphony_ctx *c) { c->mu->lock(); }
void rim_ctx_unlock(rimphony_ctx *c) { c->mu->unlock(); }

static const int NPARAMS[4] = { 4, 1, 5, 4 };

extern "C" int rimphony_dist_nparams(int kind)
{
    if (kind < 0 || kind > 3) return RIMPHONY_EINVAL;
    return NPARAMS[kind];
}

extern "C" const char *rimphony_strerror(int code)
{
    switch (code) {
    case RIMPHONY_OK: return "success";
    case RIMPHONY_EINVAL: return "invalid argument";
    case RIMPHONY_EHIP: return "HIP runtime error";
    case RIMPHONY_ENOMEM: return "out of memory";
    case RIMPHONY_ENODEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case RIMPHONY_EBUSY: return "the device is in use by another rimphony context (RIMPHONY_EXCLUSIVE=1)";
    case RIMPHONY_ENOTSUP: return "not supported";
    case RIMPHONY_ERCCL: return "RCCL call failed";
    default: return "unknown error";
    }
}

extern "C" const char *rimphony_version(void) { return "rimphony_hip 0.2 (gfx950)"; }

// One context per GPU is the supported configuration: the persistent grids of the two coefficient kernels fill the
// device and their idle waves help the busy ones (cooperative tail), which only works while every wave of the grid is
// resident.  Whether somebody else is on the device cannot be asked of the runtime, so contexts announce themselves:
// the first one on a device (any process) takes an exclusive flock on /dev/shm/rimphony_hip.<pci bus id>.lock and
// keeps it for its lifetime; a context that finds the lock taken runs in SHARED mode -- quarter-size grids, no
// cooperative tail, hence no wave ever waits for another -- or, with RIMPHONY_EXCLUSIVE=1, is refused (RIMPHONY_EBUSY).
static int take_device_lock(int device, int *fd_out)
{
    *fd_out = -1;
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int) sizeof bus, device) != hipSuccess) snprintf(bus, sizeof bus, "dev%d", device);
    for (char *q = bus; *q; q++) if (*q == ':' || *q == '/') *q = '_';
    char path[160];
    snprintf(path, sizeof path, "/dev/shm/rimphony_hip.%s.lock", bus);
    // /dev/shm is world-writable: never follow a planted symlink, only lock a regular file, and only widen the mode of
    // a file this process has just created (O_EXCL), so that another user's context can open it later.
    int fd = open(path, O_CREAT | O_EXCL | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);
    if (fd >= 0) (void) fchmod(fd, 0666);
    else fd = open(path, O_RDWR | O_CLOEXEC | O_NOFOLLOW);
    if (fd < 0) return 0;                       // cannot tell: assume the device is ours
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { close(fd); return 0; }
    if (flock(fd, LOCK_EX | LOCK_NB) == 0) { *fd_out = fd; return 0; }
    close(fd);
    return 1;                                   // somebody else holds it
}

static void ctx_free(rimphony_ctx *c)
{
    if (c->d_queue) (void) hipFree(c->d_queue);
    if (c->d_norm) (void) hipFree(c->d_norm);
    if (c->d_perm) (void) hipFree(c->d_perm);
    if (c->d_spill) (void) hipFree(c->d_spill);
    if (c->d_board) (void) hipFree(c->d_board);
    if (c->d_gspill) (void) hipFree(c->d_gspill);
    if (c->d_gboard) (void) hipFree(c->d_gboard);
    if (c->d_in) (void) hipFree(c->d_in);
    if (c->d_out) (void) hipFree(c->d_out);
    if (c->d_status) (void) hipFree(c->d_status);
    if (c->hb_host) (void) hipHostFree(c->hb_host);
    if (c->ev_start) (void) hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void) hipEventDestroy(c->ev_stop);
    if (c->ev_fstart) (void) hipEventDestroy(c->ev_fstart);
    if (c->ev_fstop) (void) hipEventDestroy(c->ev_fstop);
    if (c->ev_batch) (void) hipEventDestroy(c->ev_batch);
    if (c->lock_fd >= 0) close(c->lock_fd);     // releases the flock
    delete c->mu;
    delete c;
}

extern "C" int rimphony_ctx_create(int device, rimphony_ctx **out)
{
    if (!out) return RIMPHONY_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RIMPHONY_ENODEVICE;
    if (device < 0 || device >= ndev) return RIMPHONY_EINVAL;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    rimphony_ctx *c = new (std::nothrow) rimphony_ctx();
    if (!c) return RIMPHONY_ENOMEM;
    memset(c, 0, sizeof *c);
    c->device = device;
    c->lock_fd = -1;
    c->mu = new (std::nothrow) std::recursive_mutex();
    if (!c->mu) { delete c; return RIMPHONY_ENOMEM; }
    { const char *e = getenv("RIMPHONY_NO_ASSIST"); c->no_assist = (e && e[0] == '1'); }
    { const char *e = getenv("RIMPHONY_SYM_SOLO"); c->sym_solo = (e && e[0] == '1'); }
    { const char *e = getenv("RIMPHONY_FARADAY_ORDER"); c->faraday_symphony_order = (e && e[0] == 's'); }
    { const char *e = getenv("RIMPHONY_ROUNDS"); c->rounds = (e && e[0] == '0') ? 0 : 1; }
    { const char *e = getenv("RIMPHONY_EARLY_SQUAD"); c->early_squad = e ? atoi(e) : -1; if (c->early_squad > 1024) c->early_squad = 1024; }
    { const char *e = getenv("RIMPHONY_EARLY_MIN"); c->early_min = e ? atoi(e) : 16; if (c->early_min < 1) c->early_min = 1; }
    { const char *e = getenv("RIMPHONY_F32_VARIANT"); c->f32_variant = (e && e[0] == '1'); }
    { const char *e = getenv("RIMPHONY_FARADAY_GROUP"); c->faraday_group = (e && e[0] == '1'); }
    c->shared_mode = take_device_lock(device, &c->lock_fd);
    if (c->shared_mode) {
        const char *e = getenv("RIMPHONY_EXCLUSIVE");
        if (e && e[0] == '1') { ctx_free(c); return RIMPHONY_EBUSY; }
        c->no_assist = 1;
    }
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {   // rate of wall_clock64() (s_memrealtime): 100 MHz on gfx950, asked rather than assumed
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) khz = 100000;
        c->ticks_per_s = (unsigned long long) khz * 1000ull;
        c->owner_wait_ticks = 120ull * c->ticks_per_s;
        // test hook: RIMPHONY_OWNER_WAIT_US=<microseconds> shortens the bound so that the recompute path runs
        // (tests/test_gpu_boundary.py::test_owner_fallback_changes_no_bit)
        const char *e = getenv("RIMPHONY_OWNER_WAIT_US");
        if (e && e[0]) {
            const unsigned long long us = strtoull(e, nullptr, 10);
            c->owner_wait_ticks = us * c->ticks_per_s / 1000000ull;
        }
    }
    if (hipMalloc(&c->d_queue, 16 * sizeof(unsigned long long)) != hipSuccess) { ctx_free(c); return RIMPHONY_ENOMEM; }
    if (hipEventCreate(&c->ev_start) != hipSuccess || hipEventCreate(&c->ev_stop) != hipSuccess ||
        hipEventCreate(&c->ev_fstart) != hipSuccess || hipEventCreate(&c->ev_fstop) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_batch, hipEventDisableTiming) != hipSuccess) {
        ctx_free(c);
        return RIMPHONY_EHIP;
    }
    if (hipDeviceSynchronize() != hipSuccess) { ctx_free(c); return RIMPHONY_EHIP; }
    *out = c;
    return RIMPHONY_OK;
}

extern "C" void rimphony_ctx_destroy(rimphony_ctx *c)
{
    if (!c) return;
    (void) hipSetDevice(c->device);
    (void) hipDeviceSynchronize();
    ctx_free(c);
}

extern "C" int rimphony_ctx_shared_mode(const rimphony_ctx *c) { return c ? c->shared_mode : RIMPHONY_EINVAL; }
extern "C" int rimphony_ctx_device(const rimphony_ctx *c, int *device)
{
    if (!c || !device) return RIMPHONY_EINVAL;
    *device = c->device;
    return RIMPHONY_OK;
}

int rim_ctx_enter(rimphony_ctx *c, hipStream_t st)
{
    rim_clear_last_error();
    HIP_TRY(hipSetDevice(c->device));
    if (c->ev_batch_valid) HIP_TRY(hipStreamWaitEvent(st, c->ev_batch, 0));
    return RIMPHONY_OK;
}
void rim_ctx_leave(rimphony_ctx *c, hipStream_t st)
{
    // also after a failure part-way: whatever was enqueued must finish before the workspace is reused
    if (hipEventRecord(c->ev_batch, st) == hipSuccess) c->ev_batch_valid = 1;
}

static int ensure_spill(rimphony_ctx *c, size_t waves)
{
    if (c->spill_waves >= waves) return RIMPHONY_OK;
    if (c->d_spill) (void) hipFree(c->d_spill);
    c->d_spill = nullptr;
    c->spill_waves = 0;
    if (hipMalloc(&c->d_spill, waves * SPILL_DOUBLES_PER_WAVE * sizeof(double)) != hipSuccess) return RIMPHONY_ENOMEM;
    c->spill_waves = waves;
    return RIMPHONY_OK;
}

static int ensure_norm(rimphony_ctx *c, size_t n)
{
    if (c->norm_cap >= n) return RIMPHONY_OK;
    if (c->d_norm) (void) hipFree(c->d_norm);
    if (c->d_perm) (void) hipFree(c->d_perm);
    c->d_norm = nullptr;
    c->d_perm = nullptr;
    c->norm_cap = 0;
    if (hipMalloc(&c->d_norm, n * sizeof(double)) != hipSuccess) { c->d_norm = nullptr; return RIMPHONY_ENOMEM; }
    if (hipMalloc(&c->d_perm, (n + ORDER_BUCKETS) * sizeof(unsigned)) != hipSuccess) {
        (void) hipFree(c->d_norm);
        c->d_norm = nullptr;
        c->d_perm = nullptr;
        return RIMPHONY_ENOMEM;
    }
    c->norm_cap = n;
    return RIMPHONY_OK;
}

// persistent grid: enough single-wave workgroups to fill every SIMD several times over
static unsigned persistent_grid(const rimphony_ctx *c, unsigned long long ntasks, int waves_per_cu)
{
    unsigned long long g = (unsigned long long) c->n_cu * (unsigned) waves_per_cu;
    if (g > ntasks) g = ntasks;
    if (g < 1) g = 1;
    return (unsigned) g;
}

template <int KIND>
static int launch_norm(rimphony_ctx *c, size_t n, const ParamPtrs &pp, double *d_norm, hipStream_t st)
{
    HIP_TRY(hipMemsetAsync(c->d_queue, 0, sizeof(unsigned long long), st));
    const unsigned grid = persistent_grid(c, n, 16);
    int rc = ensure_spill(c, grid);
    if (rc) return rc;
    hipLaunchKernelGGL(norm_kernel<KIND>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pp, n, d_norm, c->d_queue, c->d_spill);
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}

static int make_param_ptrs(int kind, const double *const *d_params, ParamPtrs &pp)
{
    if (kind < 0 || kind > 3 || !d_params) return RIMPHONY_EINVAL;
    for (int k = 0; k < 5; k++) pp.p[k] = nullptr;
    for (int k = 0; k < NPARAMS[kind]; k++) {
        if (!d_params[k]) return RIMPHONY_EINVAL;
        pp.p[k] = d_params[k];
    }
    return RIMPHONY_OK;
}

extern "C" int rimphony_batch_norm_device(rimphony_ctx *c, int kind, size_t n, const double *const *d_params,
                                          double *d_norm, void *stream)
{
    if (!c || kind < 0 || kind > 3) return RIMPHONY_EINVAL;
    if (n == 0) return RIMPHONY_OK;
    if (!d_norm) return RIMPHONY_EINVAL;
    ParamPtrs pp;
    int rc = make_param_ptrs(kind, d_params, pp);
    if (rc) return rc;
    hipStream_t st = (hipStream_t) stream;
    RimCtxScope scope(c, st);
    rc = scope.enter();
    if (rc) return rc;
    switch (kind) {
    case 0: return launch_norm<0>(c, n, pp, d_norm, st);
    case 1: return launch_norm<1>(c, n, pp, d_norm, st);
    case 2: return launch_norm<2>(c, n, pp, d_norm, st);
    default: return launch_norm<3>(c, n, pp, d_norm, st);
    }
}

template <class P>
static int launch_coop(rimphony_ctx *c, const SymArgs &a, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, int *resident)
{
    const unsigned long long ntasks = (unsigned long long) a.n * (unsigned) a.nslots;
    // more waves than tasks on small batches: the surplus waves start as helpers right away
    const unsigned long long want_waves = (ntasks > (1ull << 40) || c->no_assist) ? ntasks : ntasks * 64ull;
    // Every wave of the grid should be resident: an idle wave waits for the waves that still own a task.  Ask the
    // runtime how many of these workgroups a CU really holds (registers, LDS) instead of trusting the launch bounds;
    // cached per context (= per device) and kernel instantiation.  The query knows nothing of other work on the
    // device: should part of the grid not be resident after all, helpers leave after 2 s without work and an owner
    // that waits in vain recomputes its batch itself (coop_kernel), so every wait ends and no result changes.
    if (*resident == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, coop_kernel<P>, 64, RIM_DYN_LDS) != hipSuccess || nb < 1)
            nb = 4;       // conservative: one wave per SIMD
        *resident = nb < 4 * (int) P::WAVES ? nb : 4 * (int) P::WAVES;
    }
    int resident_per_cu = *resident;
    {   // The occupancy query has been seen to count one workgroup too many when the grid is LDS-bound to the last
        // granule (a grid with non-resident waves does not fail, it stalls: every helper's 2 s idle bound).  Re-derive
        // the LDS limit here with the 512-byte allocation granule and one granule of slack.
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(coop_kernel<P>)) == hipSuccess) {
            const size_t lds = ((fa.sharedSizeBytes + RIM_DYN_LDS + 511) / 512) * 512;
            if (lds > 0) {
                const int by_lds = (int) ((160 * 1024 - 512) / lds);
                if (by_lds >= 1 && by_lds < resident_per_cu) resident_per_cu = by_lds;
            }
        }
    }
    if (c->shared_mode) resident_per_cu = resident_per_cu >= 8 ? resident_per_cu / 4 : 2;   // leave room for the other tenant
    const unsigned grid = persistent_grid(c, want_waves, resident_per_cu);
    int rc = ensure_spill(c, grid);
    if (rc) return rc;
    if (c->board_slots < grid) {
        if (c->d_board) (void) hipFree(c->d_board);
        c->d_board = nullptr;
        c->board_slots = 0;
        if (hipMalloc(&c->d_board, (size_t) grid * sizeof(AssistSlot) + BOARD_FLAG_WORDS * sizeof(unsigned)) != hipSuccess)
            return RIMPHONY_ENOMEM;
        c->board_slots = grid;
    }
    SymArgs b = a;
    b.spill = c->d_spill;
    b.board = c->no_assist ? nullptr : c->d_board;
    b.board_flags = (unsigned *) (c->d_board + c->board_slots);
    b.idle_ticks = 2ull * c->ticks_per_s;
    b.owner_ticks = c->owner_wait_ticks;
    // the squad that serves the longest chain from the start (coop_common.h): only where there is a bulk to overlap
    // with (many more tasks than waves) and the GPU is this context's own; blocks k * stride with an odd stride, so
    // that the squad is spread over the XCDs (blocks go to them round-robin) and their CUs
    b.early_squad = 0;
    b.early_stride = 1;
    b.early_classes = 1;
    b.early_min = c->early_min;
    b.turbo = c->rounds;
    const unsigned want_squad = c->early_squad >= 0 ? (unsigned) c->early_squad : (unsigned) P::EARLY_SQUAD;
    if (P::EARLY_HELP && b.board && !c->shared_mode && want_squad && ntasks >= 4ull * grid && grid >= 16u * want_squad) {
        b.early_stride = (grid / want_squad) | 1u;
        const unsigned fit = (grid - 1u) / b.early_stride + 1u;
        b.early_squad = want_squad < fit ? want_squad : fit;
        // one title per 64 waves of the squad (a batch has up to 62 requests), at most four
        b.early_classes = b.early_squad >= 256u ? 4u : b.early_squad >= 128u ? 2u : 1u;
    }
    // every claim word starts closed (count 0); flags: not exhausted, `grid` waves that may fetch a task (the squad
    // never does), nobody idle
    HIP_TRY(hipMemsetAsync(c->d_board, 0, (size_t) grid * sizeof(AssistSlot), st));
    hipLaunchKernelGGL(board_init_kernel, dim3(1), dim3(128), RIM_DYN_LDS, st, b.board_flags, grid - b.early_squad);
    HIP_TRY(hipEventRecord(ev_start, st));
    hipLaunchKernelGGL(coop_kernel<P>, dim3(grid), dim3(64), RIM_DYN_LDS, st, b);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev_stop, st));
    return RIMPHONY_OK;
}

template <int KIND, int PREC = 0>
static int launch_symphony(rimphony_ctx *c, const SymArgs &a, hipStream_t st)
{
    const int rc = launch_coop<SymphonyProblem<KIND, PREC>>(c, a, st, c->ev_start, c->ev_stop,
                                                           PREC ? &c->resident_f32[KIND] : &c->resident[0][KIND]);
    if (rc == RIMPHONY_OK) c->ev_valid = 1;
    return rc;
}

template <int KIND>
static int launch_heyvaerts(rimphony_ctx *c, const SymArgs &a, hipStream_t st)
{
    const int rc = launch_coop<HeyvaertsProblem<KIND>>(c, a, st, c->ev_fstart, c->ev_fstop, &c->resident[1][KIND]);
    if (rc == RIMPHONY_OK) c->evf_valid = 1;
    return rc;
}

// The slots of a batch as (point, group) tasks (rimphony_group.hip).  Symphony: group 0 = the selected ones of
// {j_I, alpha_I, j_Q, alpha_Q}, group 1 = those of {j_V, alpha_V} (symphony_group.h); Faraday: the one group
// {rho_Q, rho_V} (heyvaerts_group.h).
static int launch_group(rimphony_ctx *c, int kind, const SymArgs &a, uint32_t coeff_mask, int faraday, hipStream_t st)
{
    GroupArgs ga;
    ga.base = a;
    ga.ngroups = 0;
    ga.gslots[0] = ga.gslots[1] = 0;
    ga.gnmem[0] = ga.gnmem[1] = 0;
    for (int grp = 0; grp < (faraday ? 1 : 2); grp++) {
        unsigned packed = 0;
        int nm = 0;
        const int lo = faraday ? 6 : (grp ? 4 : 0), hi = faraday ? 8 : (grp ? 6 : 4);
        for (int k = lo; k < hi; k++)
            if (coeff_mask & (1u << k)) { packed |= (unsigned) k << (4 * nm); nm++; }
        if (nm) { ga.gslots[ga.ngroups] = packed; ga.gnmem[ga.ngroups] = nm; ga.ngroups++; }
    }
    if (ga.ngroups == 0) return RIMPHONY_OK;
    const unsigned long long ntasks = (unsigned long long) a.n * (unsigned) ga.ngroups;
    const unsigned long long want_waves = (ntasks > (1ull << 40) || c->no_assist) ? ntasks : ntasks * 64ull;
    int *resident = &c->resident_group[faraday ? 1 : 0][kind];
    const void *kfn = rim_group_kernel(kind, faraday);
    const int waves = rim_group_waves(faraday);
    if (*resident == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, 64, RIM_DYN_LDS) != hipSuccess || nb < 1) nb = 4;
        *resident = nb < 4 * waves ? nb : 4 * waves;
    }
    int resident_per_cu = *resident;
    {   // the LDS limit re-derived with the 512-byte allocation granule and one granule of slack (see launch_coop)
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, kfn) == hipSuccess) {
            const size_t lds = ((fa.sharedSizeBytes + RIM_DYN_LDS + 511) / 512) * 512;
            if (lds > 0) {
                const int by_lds = (int) ((160 * 1024 - 512) / lds);
                if (by_lds >= 1 && by_lds < resident_per_cu) resident_per_cu = by_lds;
            }
        }
    }
    if (c->shared_mode) resident_per_cu = resident_per_cu >= 8 ? resident_per_cu / 4 : 2;
    const unsigned grid = persistent_grid(c, want_waves, resident_per_cu);
    if (c->gspill_waves < grid || (faraday && !c->gspill_faraday)) {
        if (c->d_gspill) (void) hipFree(c->d_gspill);
        c->d_gspill = nullptr;
        c->gspill_waves = 0;
        // per wave: four members' lists at the GSL limits of the path -- 0.86 MB for the Symphony groups (5000 + 1000
        // entries), 1.3 MB for the Faraday pair in lock-step (4096 + 4096; off by default): 4.4 GB of the 288 for a full
        // persistent grid of 5120 waves, held until the context is destroyed
        const size_t per_wave = faraday ? SPILL_HEYGROUP_DOUBLES_PER_WAVE : SPILL_GROUP_DOUBLES_PER_WAVE;
        const size_t bytes = (size_t) grid * (per_wave > SPILL_GROUP_DOUBLES_PER_WAVE ? per_wave : SPILL_GROUP_DOUBLES_PER_WAVE) * sizeof(double);
        if (hipMalloc(&c->d_gspill, bytes) != hipSuccess) {
            (void) hipGetLastError();
            char msg[160];
            snprintf(msg, sizeof msg, "%zu MB for the subinterval spill region of %u waves (lists beyond the LDS part)", bytes >> 20, grid);
            rim_set_last_error("hipMalloc", msg);
            return RIMPHONY_ENOMEM;
        }
        c->gspill_waves = grid;
        c->gspill_faraday = faraday;
    }
    if (c->gboard_slots < grid) {
        if (c->d_gboard) (void) hipFree(c->d_gboard);
        c->d_gboard = nullptr;
        c->gboard_slots = 0;
        if (hipMalloc(&c->d_gboard, (size_t) grid * sizeof(GroupSlot) + BOARD_FLAG_WORDS * sizeof(unsigned)) != hipSuccess)
            return RIMPHONY_ENOMEM;
        c->gboard_slots = grid;
    }
    ga.gspill = c->d_gspill;
    ga.gboard = c->d_gboard;
    ga.coop = c->no_assist ? 0 : 1;
    ga.prof = nullptr;
#if defined(RIM_PROF)
    if (grid <= RIM_PROF_ROWS) HIP_TRY(hipGetSymbolAddress((void **) &ga.prof, HIP_SYMBOL(g_rim_prof)));
#endif
    ga.base.board = nullptr;
    ga.base.board_flags = (unsigned *) (c->d_gboard + c->gboard_slots);
    ga.base.idle_ticks = 2ull * c->ticks_per_s;
    ga.base.owner_ticks = c->owner_wait_ticks;
    // every claim word starts closed; flags: not exhausted, `grid` active waves, nobody idle (the payload arrays of the
    // slots need no reset: an entry is written before its round is opened)
    HIP_TRY(hipMemset2DAsync(c->d_gboard, sizeof(GroupSlot), 0, 32, grid, st));
    hipLaunchKernelGGL(board_init_kernel, dim3(1), dim3(128), RIM_DYN_LDS, st, ga.base.board_flags, grid);
    HIP_TRY(hipEventRecord(faraday ? c->ev_fstart : c->ev_start, st));
    const int rc = rim_group_launch(kind, faraday, grid, st, ga);       // sets the thread's last error itself
    if (rc) return rc;
    HIP_TRY(hipEventRecord(faraday ? c->ev_fstop : c->ev_stop, st));
    if (faraday) c->evf_valid = 1; else c->ev_valid = 1;
    return RIMPHONY_OK;
}

// `precision` of the batch entries.  F64 is the product.  F32_INTEGRAND (BASELINE configs[4]'s fp32-core integrand) is
// NOT offered: on the anisotropic distributions -- pitchy kappa is the one configs[4] names -- its 1e-7 non-smooth noise
// sits above GSL's round-off detectors on integrals that cancel (1.53 x slower than fp64, 1.9 % new NaNs, round 2); on the
// power-law and thermal distributions it ran 0.90 / 0.96 of the fp64 time of the ROUND-2 kernel it is built on, and the
// fp64 default has since moved to the group kernel at 0.66 of that time: the variant is 1.36-1.43 x SLOWER than fp64 and
// lossier.  A precision that is slower and lossier is not a mode, it is a trap: RIMPHONY_ENOTSUP for
// every kind, unless the context was created with RIMPHONY_F32_VARIANT=1 (measurement hook: tools/f32_variant.py, the
// error-envelope test), and even then not for the anisotropic kinds.
static int rim_precision_check(const rimphony_ctx *c, int kind, int precision)
{
    if (precision == RIMPHONY_PRECISION_F64) return RIMPHONY_OK;
    if (precision != RIMPHONY_PRECISION_F32_INTEGRAND || !c || !c->f32_variant) return RIMPHONY_ENOTSUP;
#if !defined(RIM_F32_SMOOTH)      // (the measurement build of the "smooth" variant serves all four kinds)
    if (kind == RIMPHONY_PITCHY_PL || kind == RIMPHONY_PITCHY_KAPPA) return RIMPHONY_ENOTSUP;
#endif
    return RIMPHONY_OK;
}

// The batch entry points share the context's workspace (task queue, norms, visiting order, spill regions, assist
// board, timing events).  Calls on one context are therefore serialised on the host (c->mu) and ordered on the device:
// a call's stream first waits for the event the previous call recorded after its last kernel, whatever stream that was.
static int batch_compute_locked(rimphony_ctx *c, int kind, size_t n, const double *d_s, const double *d_theta,
                                const double *const *d_params, uint32_t coeff_mask, int precision,
                                double *d_out, int32_t *d_status, uint64_t *d_work, hipStream_t st);

extern "C" int rimphony_batch_compute_device_ex(rimphony_ctx *c, int kind, size_t n,
                                                const double *d_s, const double *d_theta,
                                                const double *const *d_params, uint32_t coeff_mask, int precision,
                                                double *d_out, int32_t *d_status, uint64_t *d_work, void *stream)
{
    if (!c || kind < 0 || kind > 3) return RIMPHONY_EINVAL;
    { const int prc = rim_precision_check(c, kind, precision); if (prc) return prc; }
    if (n == 0) return RIMPHONY_OK;          // empty batch: nothing to read or write
    if (!d_out || !d_s || !d_theta) return RIMPHONY_EINVAL;
    hipStream_t st = (hipStream_t) stream;
    RimCtxScope scope(c, st);
    int rc = scope.enter();
    if (rc) return rc;
    return batch_compute_locked(c, kind, n, d_s, d_theta, d_params, coeff_mask, precision, d_out, d_status, d_work, st);
}

extern "C" int rimphony_batch_compute_device(rimphony_ctx *c, int kind, size_t n,
                                             const double *d_s, const double *d_theta,
                                             const double *const *d_params, uint32_t coeff_mask,
                                             double *d_out, int32_t *d_status, void *stream)
{
    return rimphony_batch_compute_device_ex(c, kind, n, d_s, d_theta, d_params, coeff_mask, RIMPHONY_PRECISION_F64,
                                            d_out, d_status, nullptr, stream);
}

static int batch_compute_locked(rimphony_ctx *c, int kind, size_t n, const double *d_s, const double *d_theta,
                                const double *const *d_params, uint32_t coeff_mask, int precision,
                                double *d_out, int32_t *d_status, uint64_t *d_work, hipStream_t st)
{
    ParamPtrs pp;
    int rc = make_param_ptrs(kind, d_params, pp);
    if (rc) return rc;

    rc = ensure_norm(c, n);
    if (rc) return rc;

    // full_calculation()
    switch (kind) {
    case 0: rc = launch_norm<0>(c, n, pp, c->d_norm, st); break;
    case 1: rc = launch_norm<1>(c, n, pp, c->d_norm, st); break;
    case 2: rc = launch_norm<2>(c, n, pp, c->d_norm, st); break;
    default: rc = launch_norm<3>(c, n, pp, c->d_norm, st); break;
    }
    if (rc) return rc;

    SymArgs a;
    a.pp = pp;
    a.s = d_s;
    a.theta = d_theta;
    a.norm = c->d_norm;
    a.out = d_out;
    a.status = d_status;
    a.n = n;
    a.queue = c->d_queue;
    a.perm = nullptr;
    a.spill = nullptr;
    a.board = nullptr;
    a.board_flags = nullptr;
    const bool ordered = n > 4096 && n < 0xffffffffull;
    const double *const order_gmin = (kind == RIMPHONY_POWER_LAW) ? pp.p[1] : (kind == RIMPHONY_PITCHY_PL) ? pp.p[2] : nullptr;
    const double *const order_temp = (kind == RIMPHONY_THERMAL_JUETTNER) ? pp.p[0] : nullptr;
    // the visiting order of the Symphony launch now; the Faraday launch gets its own (same array, rebuilt on the stream
    // after the Symphony kernel: see below)
    auto build_order = [&](int faraday) -> int {
        unsigned *hist = c->d_perm + c->norm_cap;
        const unsigned nb = (unsigned) ((n + 255) / 256);
        HIP_TRY(hipMemsetAsync(hist, 0, ORDER_BUCKETS * sizeof(unsigned), st));
        hipLaunchKernelGGL(order_hist_kernel, dim3(nb), dim3(256), RIM_DYN_LDS, st, d_s, d_theta, order_gmin, order_temp, n, hist, faraday);
        hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(64), RIM_DYN_LDS, st, hist);
        hipLaunchKernelGGL(order_scatter_kernel, dim3(nb), dim3(256), RIM_DYN_LDS, st, d_s, d_theta, order_gmin, order_temp, n, hist,
                           c->d_perm, faraday);
        HIP_TRY(hipGetLastError());
        return RIMPHONY_OK;
    };
    if (ordered) {
        rc = build_order(0);
        if (rc) return rc;
        a.perm = c->d_perm;
    }
    a.heartbeat = c->hb_dev;
    a.hb_task = c->hb_task;
    a.work = (unsigned long long *) d_work;
    a.idle_ticks = 0;
    a.owner_ticks = 0;
    a.early_squad = 0;          // (set per launch by launch_coop)
    a.early_stride = 1;
    a.early_classes = 1;
    a.early_min = 0;
    a.turbo = 0;
    if (d_work) HIP_TRY(hipMemsetAsync(d_work, 0, n * 8 * sizeof(uint64_t), st));
    a.nslots = 0;
    uint32_t computed = 0;
    for (int k = 0; k < 6; k++)     // symphony slots: j/alpha x I,Q,V
        if (coeff_mask & (1u << k)) { a.slot[a.nslots++] = k; computed |= 1u << k; }
    for (int k = a.nslots; k < 8; k++) a.slot[k] = 0;
    SymArgs fa = a;                 // Faraday slots: rho_Q, rho_V
    fa.nslots = 0;
    for (int k = 6; k < 8; k++)
        if (coeff_mask & (1u << k)) { fa.slot[fa.nslots++] = k; computed |= 1u << k; }
    for (int k = fa.nslots; k < 8; k++) fa.slot[k] = 0;

    HIP_TRY(hipMemsetAsync(c->d_queue, 0, 16 * sizeof(unsigned long long), st));
    {
        const size_t total = n * 8;
        hipLaunchKernelGGL(fill_unselected_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), RIM_DYN_LDS, st,
                           d_out, d_status, n, computed);
        HIP_TRY(hipGetLastError());
    }
    if (a.nslots > 0) {
        if (precision == RIMPHONY_PRECISION_F32_INTEGRAND) {
            // the six Symphony coefficients with the fp32-core integrand (detmath.h); the Faraday pair below stays fp64
            // (kinds 2 and 3 were refused at the entry: RIMPHONY_ENOTSUP)
            if (kind == 0) rc = launch_symphony<0, 1>(c, a, st);
#if defined(RIM_F32_SMOOTH)
            else if (kind == 2) rc = launch_symphony<2, 1>(c, a, st);
            else if (kind == 3) rc = launch_symphony<3, 1>(c, a, st);
#endif
            else rc = launch_symphony<1, 1>(c, a, st);
        } else if (!c->sym_solo) {
            // the coefficients of a point that share their samples advance in lock-step on one wave
            rc = launch_group(c, kind, a, coeff_mask, 0, st);
        } else {
            switch (kind) {
            case 0: rc = launch_symphony<0>(c, a, st); break;
            case 1: rc = launch_symphony<1>(c, a, st); break;
            case 2: rc = launch_symphony<2>(c, a, st); break;
            default: rc = launch_symphony<3>(c, a, st); break;
            }
        }
        if (rc) return rc;
    }
    if (fa.nslots > 0 && ordered && !c->faraday_symphony_order) {
        rc = build_order(1);
        if (rc) return rc;
    }
    if (fa.nslots > 0 && c->faraday_group) {
        // rho_Q and rho_V of a point in lock-step (heyvaerts_group.h): they share their samples' coordinates, distribution
        // derivatives and Bessel functions.  Bit-identical, and measured SLOWER than one wave per coefficient (an executed
        // pass serves 1.38 coefficients and costs 1.6 solo passes: DESIGN.md section 5) -- off unless asked for.
        rc = launch_group(c, kind, fa, coeff_mask, 1, st);
        if (rc) return rc;
    } else if (fa.nslots > 0) {
        switch (kind) {
        case 0: rc = launch_heyvaerts<0>(c, fa, st); break;
        case 1: rc = launch_heyvaerts<1>(c, fa, st); break;
        case 2: rc = launch_heyvaerts<2>(c, fa, st); break;
        default: rc = launch_heyvaerts<3>(c, fa, st); break;
        }
        if (rc) return rc;
    }
    return RIMPHONY_OK;
}

extern "C" int rimphony_debug_heartbeat(rimphony_ctx *c, uint64_t task, uint64_t **host_words)
{
    if (!c || !host_words) return RIMPHONY_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->hb_host) {
        void *h = nullptr, *d = nullptr;
        HIP_TRY(hipHostMalloc(&h, 16 * sizeof(unsigned long long), hipHostMallocMapped));
        memset(h, 0, 16 * sizeof(unsigned long long));
        HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
        c->hb_host = (unsigned long long *) h;
        c->hb_dev = (unsigned long long *) d;
    }
    // the group kernel (the default for the Symphony slots) writes no heartbeat: a watchdog polling word [10] would
    // wait for ever
    if (!(task & (1ull << 62)) && !c->sym_solo) {
        rim_set_last_error("rimphony_debug_heartbeat", "Symphony tasks are only watched with RIMPHONY_SYM_SOLO=1");
        return RIMPHONY_ENOTSUP;
    }
    c->hb_task = task;
    *host_words = (uint64_t *) c->hb_host;
    return RIMPHONY_OK;
}

extern "C" int rimphony_debug_counters(rimphony_ctx *c, uint64_t out[32])
{
    if (!c || !out) return RIMPHONY_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
#if defined(RIM_PROF)
    {   // diagnostic build: 32 region timers summed over waves (then reset); out must hold 32 words
        static unsigned long long hostbuf[RIM_PROF_ROWS * 32];
        HIP_TRY(hipMemcpyFromSymbol(hostbuf, HIP_SYMBOL(g_rim_prof), sizeof hostbuf));
        for (int k = 0; k < 32; k++) out[k] = 0;
        for (size_t r = 0; r < RIM_PROF_ROWS; r++) for (int k = 0; k < 32; k++) out[k] += hostbuf[r * 32 + k];
        memset(hostbuf, 0, sizeof hostbuf);
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_rim_prof), hostbuf, sizeof hostbuf));
        return RIMPHONY_OK;
    }
#endif
    for (int k = 16; k < 32; k++) out[k] = 0;
    if (c->ev_batch_valid) HIP_TRY(hipEventSynchronize(c->ev_batch));     // the launch may be on a non-blocking stream
    HIP_TRY(hipMemcpy(out, c->d_queue, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RIMPHONY_OK;
}

extern "C" int rimphony_last_symphony_ms(rimphony_ctx *c, float *ms)
{
    if (!c || !ms) return RIMPHONY_EINVAL;
    if (!c->ev_valid) return RIMPHONY_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev_stop));
    HIP_TRY(hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
    return RIMPHONY_OK;
}

extern "C" int rimphony_last_faraday_ms(rimphony_ctx *c, float *ms)
{
    if (!c || !ms) return RIMPHONY_EINVAL;
    if (!c->evf_valid) return RIMPHONY_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev_fstop));
    HIP_TRY(hipEventElapsedTime(ms, c->ev_fstart, c->ev_fstop));
    return RIMPHONY_OK;
}

extern "C" int rimphony_last_work(rimphony_ctx *c, rimphony_work *out)
{
    if (!c || !out) return RIMPHONY_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long h[8];
    if (c->ev_batch_valid) HIP_TRY(hipEventSynchronize(c->ev_batch));     // the launch may be on a non-blocking stream
    HIP_TRY(hipMemcpy(h, c->d_queue, sizeof h, hipMemcpyDeviceToHost));
    out->samples = h[1];
    out->passes = h[2];
    out->inner_qags = h[3];
    out->faraday_samples = h[5];
    out->faraday_passes = h[6];
    out->faraday_inner_qags = h[7];
    return RIMPHONY_OK;
}

extern "C" int rimphony_last_tail(rimphony_ctx *c, uint64_t out[8])
{
    if (!c || !out) return RIMPHONY_EINVAL;
    // the queue words belong to the context's workspace: read them under its lock, after the last call's kernels
    // (a concurrent call on the same context resets them)
    std::lock_guard<std::recursive_mutex> lock(*c->mu);
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long h[16];
    if (c->ev_batch_valid) HIP_TRY(hipEventSynchronize(c->ev_batch));
    HIP_TRY(hipMemcpy(h, c->d_queue, sizeof h, hipMemcpyDeviceToHost));
    out[0] = h[14] >> 40; out[1] = h[14] & 0xffffffffffull;        // (batches << 40) | row: rows to 1.1e12
    out[2] = h[15] >> 40; out[3] = h[15] & 0xffffffffffull;
    out[4] = h[8]; out[5] = h[9];
    out[6] = h[10]; out[7] = h[11];
    return RIMPHONY_OK;
}

extern "C" int rimphony_batch_compute_ex(rimphony_ctx *c, int kind, size_t n,
                                         const double *s, const double *theta, const double *const *params,
                                         uint32_t coeff_mask, int precision, double *out, int32_t *status, uint64_t *work)
{
    if (!c || !out || kind < 0 || kind > 3 || !params) return RIMPHONY_EINVAL;
    { const int prc = rim_precision_check(c, kind, precision); if (prc) return prc; }
    if (n == 0) return RIMPHONY_OK;
    if (!s || !theta) return RIMPHONY_EINVAL;
    const int np = NPARAMS[kind];
    for (int k = 0; k < np; k++) if (!params[k]) return RIMPHONY_EINVAL;
    // ONE lock across staging, launch, synchronisation and copy-out: the staging buffers belong to the context, and a
    // second thread on the same context must not overwrite or re-allocate them before this call has copied its
    // table out (include/rimphony_hip.h: "Calls on one context are serialised").  The device entry below re-locks
    // the same (recursive) mutex.
    RimCtxScope scope(c, nullptr);
    rim_clear_last_error();
    HIP_TRY(hipSetDevice(c->device));
    const double *dp[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    uint64_t *d_work = nullptr;
    {
        if (c->ev_batch_valid) HIP_TRY(hipEventSynchronize(c->ev_batch));   // the staging buffers may still be in use
        const size_t need_in = n * (size_t) (2 + np);
        if (c->in_cap < need_in) {
            if (c->d_in) (void) hipFree(c->d_in);
            c->d_in = nullptr; c->in_cap = 0;
            if (hipMalloc(&c->d_in, need_in * sizeof(double)) != hipSuccess) { c->d_in = nullptr; return RIMPHONY_ENOMEM; }
            c->in_cap = need_in;
        }
        if (c->out_cap < n) {
            if (c->d_out) (void) hipFree(c->d_out);
            if (c->d_status) (void) hipFree(c->d_status);
            c->d_out = nullptr; c->d_status = nullptr; c->out_cap = 0;
            // status and the optional work counters share one allocation: [n][8] int32, then [n][8] uint64
            if (hipMalloc(&c->d_out, n * 8 * sizeof(double)) != hipSuccess) { c->d_out = nullptr; return RIMPHONY_ENOMEM; }
            if (hipMalloc(&c->d_status, n * 8 * (sizeof(int32_t) + sizeof(uint64_t))) != hipSuccess) {
                (void) hipFree(c->d_out);
                c->d_out = nullptr; c->d_status = nullptr;
                return RIMPHONY_ENOMEM;
            }
            c->out_cap = n;
        }
        HIP_TRY(hipMemcpy(c->d_in, s, n * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_in + n, theta, n * sizeof(double), hipMemcpyHostToDevice));
        for (int k = 0; k < np; k++) {
            HIP_TRY(hipMemcpy(c->d_in + (size_t) (2 + k) * n, params[k], n * sizeof(double), hipMemcpyHostToDevice));
            dp[k] = c->d_in + (size_t) (2 + k) * n;
        }
        if (work) d_work = (uint64_t *) (c->d_status + c->out_cap * 8);
    }
    int rc = rimphony_batch_compute_device_ex(c, kind, n, c->d_in, c->d_in + n, dp, coeff_mask, precision, c->d_out,
                                              c->d_status, d_work, nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipMemcpy(out, c->d_out, n * 8 * sizeof(double), hipMemcpyDeviceToHost));
    if (status) HIP_TRY(hipMemcpy(status, c->d_status, n * 8 * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (work) HIP_TRY(hipMemcpy(work, d_work, n * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RIMPHONY_OK;
}

extern "C" int rimphony_batch_compute(rimphony_ctx *c, int kind, size_t n,
                                      const double *s, const double *theta, const double *const *params,
                                      uint32_t coeff_mask, double *out, int32_t *status)
{
    return rimphony_batch_compute_ex(c, kind, n, s, theta, params, coeff_mask, RIMPHONY_PRECISION_F64, out, status, nullptr);
}

// ---- multi-device batch (SURVEY 8b `n_devices`, 8e) --------------------------------------------
// Row i of the caller's table goes to context i mod n_ctx (interleaved: per-point cost correlates with s and theta,
// so contiguous blocks of a sorted table would load one GPU with all the expensive points).  One host thread per
// context gathers its shard, runs the ordinary single-device call and scatters its rows back into the caller's
// [n][8] table; there is no exchange between devices while they compute, and a point's result does not depend on
// which device evaluated it, so the table is bit-identical for every n_ctx.  Contexts that sit on the SAME device
// (a 1-GPU box rehearsing the path) run one after the other: two persistent grids on one device time-slice badly.
static std::mutex g_device_mu[64];

extern "C" int rimphony_batch_compute_multi(rimphony_ctx *const *ctxs, int n_ctx, int kind, size_t n,
                                            const double *s, const double *theta, const double *const *params,
                                            uint32_t coeff_mask, int precision, double *out, int32_t *status, uint64_t *work)
{
    if (!ctxs || n_ctx < 1 || n_ctx > 64 || kind < 0 || kind > 3 || !params) return RIMPHONY_EINVAL;
    for (int r = 0; r < n_ctx; r++) if (!ctxs[r]) return RIMPHONY_EINVAL;
    { const int prc = rim_precision_check(ctxs[0], kind, precision); if (prc) return prc; }
    if (n == 0) return RIMPHONY_OK;
    if (!s || !theta || !out) return RIMPHONY_EINVAL;
    const int np = NPARAMS[kind];
    for (int k = 0; k < np; k++) if (!params[k]) return RIMPHONY_EINVAL;
    if (n_ctx == 1) return rimphony_batch_compute_ex(ctxs[0], kind, n, s, theta, params, coeff_mask, precision, out, status, work);

    std::vector<int> rcs((size_t) n_ctx, RIMPHONY_OK);
    std::vector<std::string> errs((size_t) n_ctx);
    auto run = [&](int r) {
        const size_t m = (n - (size_t) r + (size_t) n_ctx - 1) / (size_t) n_ctx;      // rows r, r + n_ctx, ...
        if (m == 0) return;
        try {
            std::vector<double> in((size_t) (2 + np) * m), o(m * 8);
            std::vector<int32_t> stt(status ? m * 8 : 0);
            std::vector<uint64_t> wk(work ? m * 8 : 0);
            const double *pp[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
            for (size_t j = 0; j < m; j++) {
                const size_t i = (size_t) r + j * (size_t) n_ctx;
                in[j] = s[i];
                in[m + j] = theta[i];
                for (int k = 0; k < np; k++) in[(size_t) (2 + k) * m + j] = params[k][i];
            }
            for (int k = 0; k < np; k++) pp[k] = in.data() + (size_t) (2 + k) * m;
            int rc;
            {
                const int dev = ctxs[r]->device;
                std::lock_guard<std::mutex> same_device(g_device_mu[dev & 63]);
                rc = rimphony_batch_compute_ex(ctxs[r], kind, m, in.data(), in.data() + m, pp, coeff_mask, precision,
                                               o.data(), status ? stt.data() : nullptr, work ? wk.data() : nullptr);
            }
            if (rc) { rcs[(size_t) r] = rc; errs[(size_t) r] = rimphony_last_error(); return; }
            for (size_t j = 0; j < m; j++) {
                const size_t i = (size_t) r + j * (size_t) n_ctx;
                memcpy(out + i * 8, o.data() + j * 8, 8 * sizeof(double));
                if (status) memcpy(status + i * 8, stt.data() + j * 8, 8 * sizeof(int32_t));
                if (work) memcpy(work + i * 8, wk.data() + j * 8, 8 * sizeof(uint64_t));
            }
        } catch (const std::bad_alloc &) {
            rcs[(size_t) r] = RIMPHONY_ENOMEM;
        }
    };
    std::vector<std::thread> threads;
    try {
        for (int r = 1; r < n_ctx; r++) threads.emplace_back(run, r);
    } catch (...) {
        for (auto &t : threads) t.join();
        return RIMPHONY_ENOMEM;
    }
    run(0);
    for (auto &t : threads) t.join();
    for (int r = 0; r < n_ctx; r++)
        if (rcs[(size_t) r]) {
            if (!errs[(size_t) r].empty()) rim_set_last_error("rimphony_batch_compute_multi", errs[(size_t) r].c_str());
            return rcs[(size_t) r];
        }
    return RIMPHONY_OK;
}

// Where did the coefficients of a table end?  hist[slot][b] = number of rows whose status word of that slot has bit b
// set (b = 0..6: the RIMPHONY_ST_* bits), hist[slot][7] = rows with status 0.
__global__ void status_hist_kernel(const int32_t *status, size_t n, unsigned long long *hist)
{
    __shared__ unsigned h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < n * 8; idx += (size_t) gridDim.x * blockDim.x) {
        const int slot = (int) (idx & 7);
        const int v = status[idx];
        if (v == 0) atomicAdd(&h[slot * 8 + 7], 1u);
        for (int b = 0; b < 7; b++) if (v & (1 << b)) atomicAdd(&h[slot * 8 + b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long) h[threadIdx.x]);
}

extern "C" int rimphony_status_histogram_device(rimphony_ctx *c, size_t n, const int32_t *d_status, uint64_t hist[64], void *stream)
{
    if (!c || !hist) return RIMPHONY_EINVAL;
    for (int k = 0; k < 64; k++) hist[k] = 0;
    if (n == 0) return RIMPHONY_OK;
    if (!d_status) return RIMPHONY_EINVAL;
    std::lock_guard<std::recursive_mutex> lock(*c->mu);
    rim_clear_last_error();
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t) stream;
    // the histogram words live behind the counter words of the queue allocation? no: they are read back here, so a
    // private scratch allocation keeps this call independent of a batch in flight
    unsigned long long *d_hist = nullptr;
    if (hipMalloc(&d_hist, 64 * sizeof(unsigned long long)) != hipSuccess) return RIMPHONY_ENOMEM;
    int rc = RIMPHONY_OK;
    if (hipMemsetAsync(d_hist, 0, 64 * sizeof(unsigned long long), st) != hipSuccess) rc = RIMPHONY_EHIP;
    if (!rc) {
        const size_t total = n * 8;
        const unsigned blocks = (unsigned) ((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        hipLaunchKernelGGL(status_hist_kernel, dim3(blocks), dim3(256), 0, st, d_status, n, d_hist);
        if (hipGetLastError() != hipSuccess) rc = RIMPHONY_EHIP;
    }
    if (!rc && hipMemcpyAsync(hist, d_hist, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) != hipSuccess) rc = RIMPHONY_EHIP;
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = RIMPHONY_EHIP;
    (void) hipFree(d_hist);
    if (rc == RIMPHONY_EHIP) rim_set_last_error("rimphony_status_histogram_device", "HIP call failed");
    return rc;
}

extern "C" int rimphony_highfreq_batch_device(rimphony_ctx *c, int kind, size_t n, const double *d_s, const double *d_theta,
                                              const double *const *d_params, double *d_out, void *stream);

// host-buffer form of rimphony_highfreq_batch_device (out: [n][2] = rho_Q, rho_V)
extern "C" int rimphony_highfreq_batch(rimphony_ctx *c, int kind, size_t n, const double *s, const double *theta,
                                       const double *const *params, double *out)
{
    if (!c || !params) return RIMPHONY_EINVAL;
    if (kind != RIMPHONY_POWER_LAW && kind != RIMPHONY_THERMAL_JUETTNER) return RIMPHONY_EINVAL;
    if (n == 0) return RIMPHONY_OK;
    if (!s || !theta || !out) return RIMPHONY_EINVAL;
    RimCtxScope scope(c, nullptr);
    {
        const int rc0 = scope.enter();
        if (rc0) return rc0;
        if (c->ev_batch_valid) HIP_TRY(hipEventSynchronize(c->ev_batch));   // the staging buffer may still be in use
    }
    const int np = kind == RIMPHONY_POWER_LAW ? 2 : 1;       // only p and gamma_min (or T) enter the closed forms
    for (int k = 0; k < np; k++) if (!params[k]) return RIMPHONY_EINVAL;
    const size_t need_in = n * (size_t) (2 + np + 2);
    if (c->in_cap < need_in) {
        if (c->d_in) (void) hipFree(c->d_in);
        c->d_in = nullptr; c->in_cap = 0;
        if (hipMalloc(&c->d_in, need_in * sizeof(double)) != hipSuccess) return RIMPHONY_ENOMEM;
        c->in_cap = need_in;
    }
    HIP_TRY(hipMemcpy(c->d_in, s, n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_in + n, theta, n * sizeof(double), hipMemcpyHostToDevice));
    const double *dp[2] = { nullptr, nullptr };
    for (int k = 0; k < np; k++) {
        HIP_TRY(hipMemcpy(c->d_in + (size_t) (2 + k) * n, params[k], n * sizeof(double), hipMemcpyHostToDevice));
        dp[k] = c->d_in + (size_t) (2 + k) * n;
    }
    double *d_res = c->d_in + (size_t) (2 + np) * n;
    const int rc = rimphony_highfreq_batch_device(c, kind, n, c->d_in, c->d_in + n, dp, d_res, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d_res, n * 2 * sizeof(double), hipMemcpyDeviceToHost));
    return RIMPHONY_OK;
}

extern "C" int rimphony_bessel_batch_device(rimphony_ctx *c, size_t count, const double *d_n, const double *d_x,
                                            double *d_j, double *d_dj, void *stream)
{
    if (!c || (count && (!d_n || !d_x))) return RIMPHONY_EINVAL;
    if (count == 0) return RIMPHONY_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(bessel_kernel, dim3((unsigned) ((count + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       count, d_n, d_x, d_j, d_dj);
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}

// ---- the reference's scalar FFI seam (leung-bessel/src/lib.rs:36-42, 56-75) -------------------
// `extern { fn pkgw_bessel_j(n: c_double, x: c_double) -> c_double; fn pkgw_bessel_dj(..) }`: the same two C symbols,
// so the reference's leung-bessel crate can link this library instead of bessel.c.  They are the HOST build of the
// very functions the kernels inline (dev_bessel.h is __host__ __device__): no HIP call, no GPU needed, reentrant,
// a few hundred nanoseconds per call like the C code they replace, and -- the rounding contract of detmath.h --
// the same bits as rimphony_bessel_batch_device returns for the same arguments.  Errors are NaN, as in bessel.c.
extern "C" double pkgw_bessel_j(double n, double x) { return rim::bessel_j(n, x); }
extern "C" double pkgw_bessel_dj(double n, double x) { return rim::bessel_dj(n, x); }

static int fill_point_args(int kind, const double *params, int coeff, int stokes, int negative_lobe,
                           double s, double theta, PointArgs &pa)
{
    if (kind < 0 || kind > 3 || !params) return RIMPHONY_EINVAL;
    if (coeff < 0 || coeff > 1 || stokes < 0 || stokes > 2) return RIMPHONY_EINVAL;
    for (int k = 0; k < 5; k++) pa.par[k] = (k < NPARAMS[kind]) ? params[k] : 0.;
    pa.s = s; pa.theta = theta; pa.coeff = coeff; pa.stokes = stokes; pa.negative_lobe = negative_lobe;
    return RIMPHONY_OK;
}

// normalisation of a single host-described point -> c->d_norm[0]
static int single_point_norm(rimphony_ctx *c, int kind, const double *params, hipStream_t st)
{
    int rc = ensure_norm(c, 8);
    if (rc) return rc;
    // params live in d_norm[1..5] as five 1-element SoA arrays
    double h[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int k = 0; k < NPARAMS[kind]; k++) h[1 + k] = params[k];
    HIP_TRY(hipMemcpyAsync(c->d_norm, h, sizeof h, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // h is a stack buffer
    ParamPtrs pp;
    for (int k = 0; k < 5; k++) pp.p[k] = c->d_norm + 1 + k;
    switch (kind) {
    case 0: return launch_norm<0>(c, 1, pp, c->d_norm, st);
    case 1: return launch_norm<1>(c, 1, pp, c->d_norm, st);
    case 2: return launch_norm<2>(c, 1, pp, c->d_norm, st);
    default: return launch_norm<3>(c, 1, pp, c->d_norm, st);
    }
}

template <int KIND>
__global__ void integrand_kernel_n(PointArgs pa, const double *norm_ptr, size_t count, const double *n,
                                   const double *gamma, double *out)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    SymPoint pt;
    pt.s = pa.s;
    rim_sincos(pa.theta, &pt.sin_th, &pt.cos_th);
    pt.coeff = pa.coeff;
    pt.stokes = pa.stokes;
    DistParams d;
    for (int k = 0; k < 5; k++) d.par[k] = pa.par[k];
    dist_prepare<KIND>(d, norm_ptr[0]);
    LeungOrder ord[2];
    const SymOrder so = sym_order(n[i], ord);
    out[i] = gamma_integrand<KIND>(pt, d, so, gamma[i]);
}

extern "C" int rimphony_gamma_integrand_batch_device(rimphony_ctx *c, int kind, const double *params,
                                                     int coeff, int stokes, double s, double theta,
                                                     size_t count, const double *d_n, const double *d_gamma,
                                                     double *d_out, void *stream)
{
    if (!c || (count && (!d_n || !d_gamma || !d_out))) return RIMPHONY_EINVAL;
    PointArgs pa;
    int rc = fill_point_args(kind, params, coeff, stokes, 0, s, theta, pa);
    if (rc) return rc;
    if (count == 0) return RIMPHONY_OK;
    hipStream_t st = (hipStream_t) stream;
    RimCtxScope scope(c, st);
    rc = scope.enter();
    if (rc) return rc;
    rc = single_point_norm(c, kind, params, st);
    if (rc) return rc;
    const dim3 grid((unsigned) ((count + 63) / 64)), block(64);
    switch (kind) {
    case 0: hipLaunchKernelGGL(integrand_kernel_n<0>, grid, block, RIM_DYN_LDS, st, pa, c->d_norm, count, d_n, d_gamma, d_out); break;
    case 1: hipLaunchKernelGGL(integrand_kernel_n<1>, grid, block, RIM_DYN_LDS, st, pa, c->d_norm, count, d_n, d_gamma, d_out); break;
    case 2: hipLaunchKernelGGL(integrand_kernel_n<2>, grid, block, RIM_DYN_LDS, st, pa, c->d_norm, count, d_n, d_gamma, d_out); break;
    default: hipLaunchKernelGGL(integrand_kernel_n<3>, grid, block, RIM_DYN_LDS, st, pa, c->d_norm, count, d_n, d_gamma, d_out); break;
    }
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}

// helpers shared with rimphony_diag.hip (rimphony_internal.h)
int rim_point_setup(rimphony_ctx *c, int kind, const double *params, int coeff, int stokes, int negative_lobe,
                    double s, double theta, hipStream_t st, PointArgs &pa)
{
    int rc = fill_point_args(kind, params, coeff, stokes, negative_lobe, s, theta, pa);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    return single_point_norm(c, kind, params, st);
}
int rim_wave_grid(rimphony_ctx *c, size_t count, int waves_per_cu, unsigned *grid)
{
    *grid = persistent_grid(c, count, waves_per_cu);
    return ensure_spill(c, *grid);
}
const double *rim_ctx_norm(const rimphony_ctx *c) { return c->d_norm; }
double *rim_ctx_spill(const rimphony_ctx *c) { return c->d_spill; }

extern "C" int rimphony_gamma_integral_batch_device(rimphony_ctx *c, int kind, const double *params,
                                                    int coeff, int stokes, int negative_lobe, double s, double theta,
                                                    size_t count, const double *d_n, double *d_out, void *stream)
{
    if (!c || (count && (!d_n || !d_out))) return RIMPHONY_EINVAL;
    PointArgs pa;
    int rc = fill_point_args(kind, params, coeff, stokes, negative_lobe, s, theta, pa);
    if (rc) return rc;
    if (count == 0) return RIMPHONY_OK;
    hipStream_t st = (hipStream_t) stream;
    RimCtxScope scope(c, st);
    rc = scope.enter();
    if (rc) return rc;
    rc = single_point_norm(c, kind, params, st);
    if (rc) return rc;
    const unsigned grid = persistent_grid(c, count, 16);
    rc = ensure_spill(c, grid);
    if (rc) return rc;
    switch (kind) {
    case 0: hipLaunchKernelGGL(gamma_integral_kernel<0>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, c->d_norm, count, d_n, d_out, c->d_spill); break;
    case 1: hipLaunchKernelGGL(gamma_integral_kernel<1>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, c->d_norm, count, d_n, d_out, c->d_spill); break;
    case 2: hipLaunchKernelGGL(gamma_integral_kernel<2>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, c->d_norm, count, d_n, d_out, c->d_spill); break;
    default: hipLaunchKernelGGL(gamma_integral_kernel<3>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, c->d_norm, count, d_n, d_out, c->d_spill); break;
    }
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}

// ---- unit seam for the leaf functions of detmath.h ------------------------------------------
__global__ void detmath_kernel(int op, size_t n, const double *x, const double *y, double *out)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i], b = y ? y[i] : 0.;
    double r, t;
    switch (op) {
    case 0: r = rim_exp(a); break;
    case 1: r = rim_log(a); break;
    case 2: r = rim_log10(a); break;
    case 3: r = rim_pow(a, b); break;
    case 4: r = rim_sqrt(a); break;
    case 5: r = rim_log10_region(a); break;
    case 6: r = rim_lgamma_pos(a); break;
    case 7: rim_sincos(a, &r, &t); break;
    case 8: rim_sincos(a, &t, &r); break;
    case 9: r = rim_div_by(a, b, 1. / b); break;
    case 10: r = rim_cbrt_normal(a); break;
    case 11: r = rim_rgamma_near(a); break;
    case 12: { double o4[4]; rim_third_powers(a, o4); const int j = (int) b; r = j == 0 ? o4[0] : j == 1 ? o4[1] : j == 2 ? o4[2] : o4[3]; break; }
    case 13: r = rim_rqrt4_normal(a); break;
    case 14: r = rim_powexp_normal(a, b, -a * 1e-3); break;
    default: r = RIM_NAN; break;
    }
    out[i] = r;
}

extern "C" int rimphony_detmath_batch_device(rimphony_ctx *c, int op, size_t n, const double *d_x, const double *d_y,
                                             double *d_out, void *stream)
{
    if (!c || op < 0 || op > 14) return RIMPHONY_EINVAL;
    if (n == 0) return RIMPHONY_OK;
    if (!d_x || !d_out || ((op == 3 || op == 9 || op == 12 || op == 14) && !d_y)) return RIMPHONY_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(detmath_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream, op, n, d_x, d_y, d_out);
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}

// ---- high-frequency closed forms ---------------------------------------------
__global__ void highfreq_kernel(int kind, size_t n, const double *s, const double *theta, const double *p0,
                                const double *p1, double *out)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double sn, cs;
    rim_sincos(theta[i], &sn, &cs);
    double q, v;
    if (kind == RIMPHONY_POWER_LAW) {
        q = rim_hf_powerlaw_faraday_q(p0[i], p1[i], s[i], sn);
        v = rim_hf_powerlaw_faraday_v(p0[i], p1[i], s[i], sn);
    } else {
        rim_hf_thermal_faraday(p0[i], s[i], sn, cs, &q, &v);
    }
    out[2 * i] = q;
    out[2 * i + 1] = v;
}

extern "C" int rimphony_highfreq_batch_device(rimphony_ctx *c, int kind, size_t n, const double *d_s, const double *d_theta,
                                              const double *const *d_params, double *d_out, void *stream)
{
    if (!c) return RIMPHONY_EINVAL;
    if (kind != RIMPHONY_POWER_LAW && kind != RIMPHONY_THERMAL_JUETTNER) return RIMPHONY_EINVAL;
    if (n == 0) return RIMPHONY_OK;
    if (!d_s || !d_theta || !d_params || !d_out || !d_params[0]) return RIMPHONY_EINVAL;
    if (kind == RIMPHONY_POWER_LAW && !d_params[1]) return RIMPHONY_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(highfreq_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream, kind, n,
                       d_s, d_theta, d_params[0], kind == RIMPHONY_POWER_LAW ? d_params[1] : nullptr, d_out);
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}

extern "C" int rimphony_qag_selftest_device(rimphony_ctx *c, size_t count, const int32_t *d_family,
                                            const double *d_p0, const double *d_p1, const double *d_a, const double *d_b,
                                            double epsabs, double epsrel, int32_t limit,
                                            double *d_result, double *d_abserr, int32_t *d_qstatus, int32_t *d_size,
                                            void *stream)
{
    if (!c || (count && (!d_family || !d_p0 || !d_p1 || !d_a || !d_b || !d_result || !d_abserr || !d_qstatus || !d_size)))
        return RIMPHONY_EINVAL;
    if (limit < 1) return RIMPHONY_EINVAL;
    if (count == 0) return RIMPHONY_OK;
    RimCtxScope scope(c, (hipStream_t) stream);
    int rc = scope.enter();
    if (rc) return rc;
    const unsigned grid = persistent_grid(c, count, 16);
    rc = ensure_spill(c, grid);
    if (rc) return rc;
    hipLaunchKernelGGL(qag_selftest_kernel, dim3(grid), dim3(64), 0, (hipStream_t) stream, count, d_family, d_p0, d_p1,
                       d_a, d_b, epsabs, epsrel, (int) limit, d_result, d_abserr, d_qstatus, d_size, c->d_spill);
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}
