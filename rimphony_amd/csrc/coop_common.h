// coop_common.h -- what the persistent, cooperative kernels of librimphony_hip.so share: the kernel arguments, the task
// queue, the assist board (cooperative tail) and the per-task context loads.  Included by rimphony_hip.hip
// (coop_kernel<P>: the Heyvaerts kernel, the solo Symphony kernels) and rimphony_group.hip (group_kernel: the
// Symphony coefficients of a point in lock-step); they are separate translation units because hipcc's code
// generation for one big kernel depends on what else is in the unit (rimphony_internal.h).
#ifndef RIM_COOP_COMMON_H
#define RIM_COOP_COMMON_H

#include <hip/hip_runtime.h>
#include "symphony_wave.h"

using namespace rim;

struct ParamPtrs { const double *p[5]; };
struct AssistSlot;

template <int KIND>
__device__ inline void load_params(const ParamPtrs &pp, size_t i, DistParams &d)
{
    constexpr int NP = (KIND == DIST_POWER_LAW) ? 4 : (KIND == DIST_THERMAL_JUETTNER) ? 1 : (KIND == DIST_PITCHY_PL) ? 5 : 4;
#pragma unroll
    for (int k = 0; k < 5; k++) d.par[k] = (k < NP) ? pp.p[k][i] : 0.;
}

// Fetch the next task index for the whole wave.  The two barriers are not for
// memory ordering: they are convergent operations that pin the reconvergence
// point of the surrounding `if (lane == 0)` regions.  Without them LLVM threads
// the lane-0-only epilogue of one loop iteration straight into the lane-0-only
// atomic of the next, the structurizer then builds a separate inner cycle for
// lanes 1..63, and those lanes run ahead of lane 0, read t from their own first
// lane (always 0) and re-execute task 0 forever (observed on gfx950, ROCm 7.2).
__device__ __forceinline__ unsigned long long wave_next_task(unsigned long long *queue, int lane)
{
    __syncthreads();
    unsigned long long t = 0;
    if (lane == 0) t = atomicAdd(queue, 1ull);
    __syncthreads();
    return ((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (t >> 32)) << 32) |
           (unsigned) __builtin_amdgcn_readfirstlane((int) (t & 0xffffffffull));
}

struct SymArgs {
    ParamPtrs pp;
    const double *s, *theta, *norm;
    double *out;
    int32_t *status;
    size_t n;
    int nslots;
    int slot[8];
    unsigned long long *queue;      // [0] task head, [1] samples, [2] passes, [3] inner qags
    const unsigned *perm;           // task order (expensive points first) or null
    double *spill;                  // [gridDim.x][SPILL_DOUBLES_PER_WAVE] wave-private global store overflow
    struct AssistSlot *board;       // [gridDim.x] cooperative-tail board (null: cooperation off)
    unsigned *board_flags;          // [0] task queue exhausted, [1] waves that still own / may fetch a task
    unsigned long long *heartbeat;  // diagnostics: host-mapped words written by the wave that owns hb_task
    unsigned long long hb_task;
    unsigned long long *work;       // optional [n][8]: integrand samples spent on each coefficient (null: not counted)
    unsigned long long idle_ticks;  // wall_clock64 ticks after which a helper that has found nothing leaves (2 s)
    unsigned long long owner_ticks; // ... after which an owner stops waiting for helpers and recomputes its batch (120 s)
    // Early help for the launch's longest chain (Faraday kernel; early_squad = 0: off).  A task's batches are sequential
    // (an outer quadrature on its way to GSL's iteration limit: thousands of them), and while the queue is full its owner
    // evaluates every batch alone -- ~10 ms each -- so the chain is still young when the queue runs dry and the launch
    // then waits for it.  `early_squad` waves of the grid (blocks k * early_stride) never fetch a task: they are helpers
    // from the start.  An owner whose outer quadrature has grown past `early_min` subintervals enters that size in
    // flags[CHAMP] (atomic max); the one task that holds the maximum -- the CHAMPION -- publishes its batches to the squad
    // while the queue is still full, and gives the title back when its quadrature ends.  A quadrature that converges
    // holds the title for a few batches at most (the squad serves it and it is soon over); the one that is on its way to
    // GSL's limit outgrows every other, keeps the title and advances at the pace of the cooperative tail (~0.25 ms a
    // batch) alongside the bulk instead of after it.  Who evaluates a request never changes its value: tables are bit-identical.
    int turbo;                  // rounds of the long outer quadratures (heyvaerts_wave.h; RIMPHONY_ROUNDS=0 turns them off)
    unsigned early_squad, early_stride;
    unsigned early_classes;     // 1, 2 or 4 titles (64 waves of the squad each): that many long quadratures are served side by side
    int early_min;
};

__constant__ int c_slot_coeff[8] = { 0, 1, 0, 1, 0, 1, 2, 2 };
__constant__ int c_slot_stokes[8] = { 0, 0, 1, 1, 2, 2, 1, 2 };

// ---- cooperative tail: the assist board ---------------------------------------------------
//
// Per-task cost has a heavy tail (rare points are 100-1000x the mean; the reference prints
// "SLOW" for them).  With one wave per task, the end of a launch would be a few waves grinding
// through such tasks while the rest of the chip idles.  Once the task queue has run dry and some
// wave is idle, a wave that still owns a task therefore PUBLISHES each batch of (n, lobe) requests
// of its state machine on its slot of a board in global memory, and waves without a task claim
// individual requests, evaluate the gamma-integral and write the value back.  Every request is a
// pure function of (point, coefficient, n, lobe), so who evaluates it cannot change a bit of the
// result.
//
// Protocol (placement-independent, agent-scope release/acquire, every spin bounded):
//   slot.claim = (seq << 32) | (count << 8) | next        open iff next < count
//   owner:  write the compacted requests (sc1 stores) -> s_waitcnt vmcnt(0) ->
//           claim = (seq << 32) | (count << 8); advertise the slot in one of 64 hint words
//   anyone: c = load(claim); if next(c) < count(c): CAS(claim, c, c + 1) claims request next(c).
//           A successful CAS proves batch `seq` was still open, so the request data (read with sc1
//           loads) belonged to it and stays valid until this claimer reports `done`.
//   result: res[k], status[k] (sc1 stores) -> s_waitcnt vmcnt(0) -> atomicAdd(done)
//   owner:  poll done == count -> read res[] (sc1 loads) -> claim = (seq << 32) (closed)
// Discovery: there are 64 hint lines; an idle wave polls ONE of them (its index mod 64: one sc1 load
// per poll, with exponential back-off) and looks at a slot's claim word only when the hint it
// sees has changed.  An owner advertises a batch on every hint line of one "channel"
// (line & (span - 1) == channel), where span ~ the number of waves that still own a task: with
// 100 owners each batch is seen by 1/64 of the idle waves, with 2 owners by half of them, yet the
// polls stay spread over 64 lines.  (All waves polling the same few lines -- and a per-poll look at
// the claim word -- was measured to slow the computing waves several-fold.)
// Idle waves count themselves in flags[IDLE] so that owners publish only when somebody can help.
// A batch normally has up to 62 requests (the Kronrod nodes of an interval's two children).  A long outer quadrature of
// the Faraday kernel publishes the children of up to RIM_TURBO_MAX intervals in one batch (heyvaerts_wave.h, "rounds"):
// 62 requests per interval, 248 in all -- the count and cursor fields of the claim word are 8 bits wide.
#define RIM_SLOT_REQS 256           // >= 62 * RIM_TURBO_MAX (symphony_wave.h)
struct AssistSlot {
    unsigned long long claim;
    unsigned long long point;
    unsigned done;
    int slot;
    unsigned long long req_n[RIM_SLOT_REQS];     // bit patterns of doubles: every access is an agent-scope atomic
    unsigned long long res[RIM_SLOT_REQS];
    int req_lobe[RIM_SLOT_REQS];
    int res_status[RIM_SLOT_REQS];
    unsigned res_samples[RIM_SLOT_REQS];         // integrand samples the request took (booked by the owner when it reads the value)
};

// Board payload accessors: relaxed agent-scope atomics = sc1 (write-through / L1-bypassing) stores
// and loads.  Thousands of waves use the board at the same time at the end of a launch; agent-scope
// release/acquire FENCES there (an L2 write-back + L1 invalidate each) serialise on the XCD's L2 and
// cost more than the work being shared, so the hand-off uses write-through granules instead:
// sc1 stores -> s_waitcnt vmcnt(0) -> counter/flag atomic on the producer, sc1 loads after the
// poll on the consumer (MI355X_MICROARCH.md, "Valid forms").
__device__ __forceinline__ void bput(unsigned long long *p, unsigned long long v)
{ __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bput(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bput(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long bget(const unsigned long long *p)
{ return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int bget(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned bget(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Flag words; each lives in its own 128-byte line, because thousands of waves poll them.
#define BOARD_FLAG_EXHAUSTED 0
#define BOARD_FLAG_ACTIVE 32
#define BOARD_FLAG_IDLE 64
#define BOARD_FLAG_CHAMP 96      // early help: (size << 16) | (block + 1) of the longest outer quadrature in progress
#define BOARD_HINTS 224         // (four title lines: 96, 128, 160, 192) 64 hint lines (stride 32 words): (seq << 16) | (slot + 1) of a published batch
#define BOARD_HINT_STRIDE 32
#define BOARD_FLAG_WORDS (BOARD_HINTS + 64 * BOARD_HINT_STRIDE)

__device__ __forceinline__ unsigned long long bcast_u64(unsigned long long v)
{
    return ((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (v >> 32)) << 32) |
           (unsigned) __builtin_amdgcn_readfirstlane((int) (v & 0xffffffffull));
}

// Number of hint words in use (power of two <= 64): about one per wave that still owns a task, so
// that each published batch is seen by (idle waves / owners) helpers.
__device__ __forceinline__ unsigned hint_span(unsigned active)
{
    unsigned h = 1;
    while (h < 64u && h * 2u <= active) h *= 2u;
    return h;
}

__device__ __forceinline__ bool claim_open(unsigned long long c) { return (unsigned) (c & 0xff) < (unsigned) ((c >> 8) & 0xff); }

// Claim the next unevaluated request of batch `seq` on `slot`; -1 if there is none (or the batch is over).
__device__ __forceinline__ int assist_claim(AssistSlot *slot, unsigned seq, int lane)
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

__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// lane index of the k-th set bit of mask (k < popcount(mask))
__device__ __forceinline__ int kth_set_bit(unsigned long long mask, int k)
{
    for (int i = 0; i < k; i++) mask &= mask - 1;
    return __builtin_ffsll((long long) mask) - 1;
}

template <int KIND>
__device__ __forceinline__ void load_context(const SymArgs &a, size_t i, int slot, SymPoint &pt, DistParams &d, double &norm)
{
    pt.s = uni(a.s[i]);
    rim_sincos(a.theta[i], &pt.sin_th, &pt.cos_th);
    pt.sin_th = uni(pt.sin_th);
    pt.cos_th = uni(pt.cos_th);
    pt.coeff = uni(c_slot_coeff[slot]);
    pt.stokes = uni(c_slot_stokes[slot]);
    load_params<KIND>(a.pp, i, d);
    norm = uni(a.norm[i]);
    dist_prepare<KIND>(d, norm);
#pragma unroll
    for (int k = 0; k < 5; k++) d.par[k] = uni(d.par[k]);
    d.inv_gamma_cutoff = uni(d.inv_gamma_cutoff);
    d.inv_kappa_width = uni(d.inv_kappa_width);
    d.neg_inverse_t = uni(d.neg_inverse_t);
}

#endif
