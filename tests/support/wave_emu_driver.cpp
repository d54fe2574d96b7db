// CPU emulation driver: one Symphony coefficient computed by the wave-level code
// of rimphony_amd/csrc on 64 host threads (tests/debugging only).
#define RIM_WAVE_EMU 1
#include <cstdio>
#include <vector>
#include "../../rimphony_amd/csrc/symphony_wave.h"
#include "../../rimphony_amd/csrc/heyvaerts_wave.h"

using namespace rim;

#define CAP_INNER 64
#define CAP_OUTER 64

struct EmuTask {
    int faraday;
    int kind, coeff, stokes;
    double s, theta, par[5], norm;
    double result[64];
    int status[64];
    unsigned long long samples, passes, inner_qags;
};

template <int KIND>
static void lane_body(EmuTask *t)
{
    __shared__ double s_tab[96];
    __shared__ double s_inner[RIM_ISTORE_DOUBLES(CAP_INNER)];
    __shared__ double s_outer[RIM_ISTORE_DOUBLES(CAP_OUTER)];
    __shared__ TaskState s_park;
    __shared__ QagPark s_qpark;
    const GKLane g = gk_lane_init(s_tab);
    static double s_spill[RIM_ISTORE_DOUBLES(5000) + RIM_ISTORE_DOUBLES(4096)];
    const IStore inner = istore_carve(s_inner, CAP_INNER, s_spill, 5000);
    const IStore outer = istore_carve(s_outer, CAP_OUTER, s_spill + RIM_ISTORE_DOUBLES(5000), 4096);
    if (g.lane == 0) s_qpark.ctr = WaveCounters{0, 0, 0};
    wv_sync();

    SymPoint pt;
    pt.s = uni(t->s);
    rim_sincos(t->theta, &pt.sin_th, &pt.cos_th);
    pt.coeff = t->coeff;
    pt.stokes = t->stokes;
    DistParams d;
    for (int k = 0; k < 5; k++) d.par[k] = t->par[k];
    dist_prepare<KIND>(d, t->norm);
    int st = 0;
    double val;
    if (t->faraday) {
        __shared__ HeyTask s_hpark;
        HeyPoint hp;
        hp.s = pt.s; hp.sin_th = pt.sin_th; hp.cos_th = pt.cos_th;
        hp.sigma0 = hp.s * hp.sin_th; hp.sigma0_sq = hp.sigma0 * hp.sigma0; hp.stokes = t->stokes;
        const HeyConsts hc = hey_consts(hey_series_table_host(), hey_series_lds_table_host());
        val = heyvaerts_coefficient<KIND>(hp, d, hc, g, inner, outer, &s_hpark, &s_qpark, st);
    } else {
        val = symphony_coefficient<KIND>(pt, d, g, inner, outer, &s_park, &s_qpark, st);
    }
    t->result[g.lane] = val;
    t->status[g.lane] = st;
    wv_sync();
    if (g.lane == 0) {
        t->samples = s_qpark.ctr.samples;
        t->passes = s_qpark.ctr.steps;
        t->inner_qags = s_qpark.ctr.inner_qags;
    }
}

struct ThreadArg { EmuTask *t; int lane; };

static void *thread_main(void *p)
{
    ThreadArg *a = (ThreadArg *) p;
    emu_lane_ref() = a->lane;
    switch (a->t->kind) {
    case 0: lane_body<0>(a->t); break;
    case 1: lane_body<1>(a->t); break;
    case 2: lane_body<2>(a->t); break;
    default: lane_body<3>(a->t); break;
    }
    return nullptr;
}

extern "C" int emu_symphony(int kind, int coeff, int stokes, double s, double theta, const double *par, double norm,
                            double *result, int *status, unsigned long long *work3)
{
    EmuTask t;
    t.faraday = (coeff == 2);
    t.kind = kind; t.coeff = coeff; t.stokes = stokes; t.s = s; t.theta = theta; t.norm = norm;
    for (int k = 0; k < 5; k++) t.par[k] = par[k];
    pthread_barrier_init(&emu_wave().bar, nullptr, 64);
    pthread_t th[64];
    ThreadArg args[64];
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 1 << 20);
    for (int i = 0; i < 64; i++) {
        args[i].t = &t; args[i].lane = i;
        pthread_create(&th[i], &attr, thread_main, &args[i]);
    }
    for (int i = 0; i < 64; i++) pthread_join(th[i], nullptr);
    pthread_barrier_destroy(&emu_wave().bar);
    int uniform = 1;
    for (int i = 1; i < 64; i++) {
        if (std::memcmp(&t.result[i], &t.result[0], 8) != 0 || t.status[i] != t.status[0]) uniform = 0;
    }
    *result = t.result[0];
    *status = t.status[0];
    work3[0] = t.samples; work3[1] = t.passes; work3[2] = t.inner_qags;
    return uniform;
}
