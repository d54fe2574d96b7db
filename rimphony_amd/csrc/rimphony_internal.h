// rimphony_internal.h -- what the translation units of librimphony_hip.so share (not installed).
//
// The diagnostic kernels live in their own translation unit (rimphony_diag.hip): with them in the same unit as
// coop_kernel, hipcc's code generation for coop_kernel changes (87 -> 113 spilled VGPRs, +1.4 % run time,
// measured), although they share no non-inlined code.
#ifndef RIMPHONY_INTERNAL_H
#define RIMPHONY_INTERNAL_H

#include <hip/hip_runtime.h>
#include "../../include/rimphony_hip.h"
#include "symphony_wave.h"

// LDS subinterval-store capacities.  Largest counts seen on the reference's golden file and
// on the bench tables: 31 (gamma integrals), 41 (n chunks), 48 (normalisation).
#ifndef CAP_INNER
#define CAP_INNER 64
#endif
#ifndef CAP_OUTER
#define CAP_OUTER 64
#endif
#define CAP_NORM 256
// per-wave global spill behind the LDS stores: the GSL limits of the path (5000 inner for
// Symphony, 4096 for Heyvaerts, 1000 / 4096 outer)
#define SPILL_INNER 5000
#define SPILL_OUTER 4096
#define SPILL_DOUBLES_PER_WAVE (RIM_ISTORE_DOUBLES(SPILL_INNER) + RIM_ISTORE_DOUBLES(SPILL_OUTER))
// minimum waves per SIMD the register allocator must leave room for (symphony kernel)
// (measured on MI355X, 65536-point launches: 4 -> 25.8k, 5 -> 27.0k, 6 -> 27.8k points/s; at 6 the
// allocator spills 47 VGPRs to scratch and still wins)
#ifndef RIM_SYM_WAVES
#define RIM_SYM_WAVES 6
#define RIM_HEY_WAVES 5          // heyvaerts: 96 VGPRs (2..6 measured: 869, 708, 653, 626, 646 ms on the 8192-point power-law batch;
                                 // final build, 4 / 5 / 6: 782 / 741 / 718 ms on 16384 power-law points but 993 / 967 / 996 ms on 4096
                                 // pitchy-kappa points, whose tail is sequential -- 5 kept)
#endif

// waves of the Faraday kernel's grid that serve the launch's longest outer quadrature from the start (coop_common.h,
// SymArgs::early_squad), for the kinds that have such chains (HeyvaertsProblem::EARLY_SQUAD); RIMPHONY_EARLY_SQUAD
// overrides for every kind, 0 = off
#ifndef RIM_EARLY_SQUAD_DEFAULT
#define RIM_EARLY_SQUAD_DEFAULT 64
#endif

#if defined(RIM_PROF)
#define RIM_DYN_LDS 256             // the region timers accumulate in dynamic LDS
#else
#define RIM_DYN_LDS 0
#endif
// A failed HIP call: remember what failed where rimphony_last_error() can find it (thread-local; a library does
// not print) and return RIMPHONY_EHIP.
void rim_set_last_error(const char *what, const char *detail);
#define HIP_TRY(expr)                                                            \
    do {                                                                         \
        hipError_t e_ = (expr);                                                  \
        if (e_ != hipSuccess) {                                                  \
            rim_set_last_error(#expr, hipGetErrorString(e_));                    \
            return RIMPHONY_EHIP;                                                \
        }                                                                        \
    } while (0)


struct PointArgs {
    double par[5];
    double s, theta;
    int coeff, stokes, negative_lobe;
};

// validate + normalise one host-described parameter point (norm -> rim_ctx_norm(c)[0]) on `st`
int rim_point_setup(rimphony_ctx *c, int kind, const double *params, int coeff, int stokes, int negative_lobe,
                    double s, double theta, hipStream_t st, PointArgs &pa);
// persistent grid of single-wave workgroups for `count` work items + its per-wave spill region
int rim_wave_grid(rimphony_ctx *c, size_t count, int waves_per_cu, unsigned *grid);
// Every entry point that touches the context's workspace (norms, spill regions, staging buffers, queue words) runs
// inside one of these: the context's (recursive) host lock for the whole call -- staging, launches, synchronisation and
// copy-out included --, the device selected, `st` ordered behind the previous call's work on the workspace, and that
// call's own last kernel recorded for the next one when the scope ends.  enter() also forgets the thread's previous
// error text, so that rimphony_last_error() never describes an older call.
void rim_ctx_lock(rimphony_ctx *c);
void rim_ctx_unlock(rimphony_ctx *c);
int rim_ctx_enter(rimphony_ctx *c, hipStream_t st);
void rim_ctx_leave(rimphony_ctx *c, hipStream_t st);
void rim_clear_last_error();
struct RimCtxScope {
    rimphony_ctx *c;
    hipStream_t st;
    bool entered;
    RimCtxScope(rimphony_ctx *ctx, hipStream_t stream) : c(ctx), st(stream), entered(false) { rim_ctx_lock(c); }
    int enter() { const int rc = rim_ctx_enter(c, st); entered = (rc == 0); return rc; }
    ~RimCtxScope() { if (entered) rim_ctx_leave(c, st); rim_ctx_unlock(c); }
    RimCtxScope(const RimCtxScope &) = delete;
    RimCtxScope &operator=(const RimCtxScope &) = delete;
};
const double *rim_ctx_norm(const rimphony_ctx *c);
double *rim_ctx_spill(const rimphony_ctx *c);

#endif
