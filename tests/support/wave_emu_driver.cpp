// CPU emulation driver: one Symphony coefficient computed by the wave-level code
// of rimphony_amd/csrc on 64 host threads (tests/debugging only).
#define RIM_WAVE_EMU 1
#include <cstdio>
#include <vector>
#include "../../rimphony_amd/csrc/symphony_wave.h"
#include "../../rimphony_amd/csrc/heyvaerts_wave.h"
namespace rim { unsigned long long g_emu_hist[64]; }
#include "../../rimphony_amd/csrc/symphony_group.h"
#include "../../rimphony_amd/csrc/heyvaerts_group.h"

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
        hey_point_derive(hp); hey_point_endless<KIND>(hp, d); hp.stokes = t->stokes;
        const HeyConsts hc = hey_consts();
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


// ---- a whole group (symphony_group.h) on the emulated wave -----------------------------------------------------
struct EmuGroupTask {
    int kind, nmem;
    unsigned slots;
    double s, theta, par[5], norm;
    double vals[64][RIM_GROUP];
    int stats[64][RIM_GROUP];
    unsigned long long samples, passes, inner_qags, member_passes, stash_filed;
};

template <int KIND>
static void group_lane_body(EmuGroupTask *t)
{
    __shared__ double s_tab[96];
    __shared__ double s_ginner[RIM_GROUP * RIM_ISTORE_DOUBLES(CAP_GINNER)];
    __shared__ double s_gouter[RIM_GROUP * RIM_ISTORE_DOUBLES(CAP_GOUTER)];
    __shared__ TaskState s_park[RIM_GROUP];
    __shared__ GroupPark s_gp;
    static double s_spill[SPILL_GROUP_DOUBLES_PER_WAVE];
    const GKLane g = gk_lane_init(s_tab);
    if (g.lane == 0) { s_gp.ctr = WaveCounters{0, 0, 0}; s_gp.member_passes = 0; s_gp.stash_filed = 0; s_gp.stash_used = 0; s_gp.hb = nullptr; }
    wv_sync();
    SymPoint pt;
    pt.s = uni(t->s);
    rim_sincos(t->theta, &pt.sin_th, &pt.cos_th);
    pt.coeff = 0;
    pt.stokes = group_slot(t->slots, 0) >> 1;
    DistParams d;
    for (int k = 0; k < 5; k++) d.par[k] = t->par[k];
    dist_prepare<KIND>(d, t->norm);
    double vals[RIM_GROUP] = { 0, 0, 0, 0 };
    int stats[RIM_GROUP] = { 0, 0, 0, 0 };
    // the members' outer (n) lists live entirely in the spill region (no LDS part), as in the kernel
    (void) s_gouter;
    symphony_group<KIND>(pt, d, t->slots, t->nmem, g, s_ginner, s_spill, nullptr,
                         s_spill + RIM_GROUP * RIM_ISTORE_DOUBLES(SPILL_GINNER), s_park, &s_gp, vals, stats);
    for (int m = 0; m < RIM_GROUP; m++) { t->vals[g.lane][m] = vals[m]; t->stats[g.lane][m] = stats[m]; }
    wv_sync();
    if (g.lane == 0) {
        t->samples = s_gp.ctr.samples; t->passes = s_gp.ctr.steps; t->inner_qags = s_gp.ctr.inner_qags;
        t->member_passes = s_gp.member_passes; t->stash_filed = s_gp.stash_filed;
    }
}

static void *group_thread_main(void *p)
{
    ThreadArg *a = (ThreadArg *) p;
    EmuGroupTask *t = (EmuGroupTask *) (void *) a->t;
    emu_lane_ref() = a->lane;
    switch (t->kind) {
    case 0: group_lane_body<0>(t); break;
    case 1: group_lane_body<1>(t); break;
    case 2: group_lane_body<2>(t); break;
    default: group_lane_body<3>(t); break;
    }
    return nullptr;
}

// slots: the members' output slots (0..5), 4 bits each; returns 1 if every lane ended with the same values
extern "C" int emu_symphony_group(int kind, unsigned slots, int nmem, double s, double theta, const double *par, double norm,
                                  double *vals, int *stats, unsigned long long *work4)
{
    static EmuGroupTask t;
    t.kind = kind; t.nmem = nmem; t.slots = slots; t.s = s; t.theta = theta; t.norm = norm;
    for (int k = 0; k < 5; k++) t.par[k] = par[k];
    pthread_barrier_init(&emu_wave().bar, nullptr, 64);
    pthread_t th[64];
    ThreadArg args[64];
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 1 << 20);
    for (int i = 0; i < 64; i++) {
        args[i].t = (EmuTask *) (void *) &t; args[i].lane = i;
        pthread_create(&th[i], &attr, group_thread_main, &args[i]);
    }
    for (int i = 0; i < 64; i++) pthread_join(th[i], nullptr);
    pthread_barrier_destroy(&emu_wave().bar);
    int uniform = 1;
    for (int i = 1; i < 64; i++)
        for (int m = 0; m < nmem; m++)
            if (std::memcmp(&t.vals[i][m], &t.vals[0][m], 8) != 0 || t.stats[i][m] != t.stats[0][m]) uniform = 0;
    for (int m = 0; m < nmem; m++) { vals[m] = t.vals[0][m]; stats[m] = t.stats[0][m]; }
    work4[0] = t.samples; work4[1] = t.passes; work4[2] = t.inner_qags; work4[3] = t.member_passes; work4[4] = t.stash_filed;
    for (int k = 0; k < 40; k++) { work4[5 + k] = rim::g_emu_hist[k]; rim::g_emu_hist[k] = 0; }
    return uniform;
}

// ---- unit seam: one application of the 31-point rule (wave_gk31) to given node values --------------------------
struct EmuRuleTask { const double *fv_nodes; double hl; double out[4]; };
static void *rule_thread_main(void *p)
{
    ThreadArg *a = (ThreadArg *) p;
    EmuRuleTask *t = (EmuRuleTask *) (void *) a->t;
    emu_lane_ref() = a->lane;
    __shared__ double s_tab[96];
    const GKLane g = gk_lane_init(s_tab);
    const int node = gk_node_of_lane(g.j);
    const double fv = (g.half == 0 && g.node) ? t->fv_nodes[node] : 0.;
    const GKRes r = wave_gk31(fv, t->hl, g);
    if (g.lane == 0) { t->out[0] = r.result; t->out[1] = r.abserr; t->out[2] = r.resabs; t->out[3] = r.resasc; }
    return nullptr;
}
// fv_nodes[31]: integrand values at the nodes in ASCENDING order of the abscissa (the order of gk31_table.h)
extern "C" void emu_gk31(const double *fv_nodes, double hl, double *out4)
{
    static EmuRuleTask t;
    t.fv_nodes = fv_nodes; t.hl = hl;
    pthread_barrier_init(&emu_wave().bar, nullptr, 64);
    pthread_t th[64];
    ThreadArg args[64];
    for (int i = 0; i < 64; i++) {
        args[i].t = (EmuTask *) (void *) &t; args[i].lane = i;
        pthread_create(&th[i], nullptr, rule_thread_main, &args[i]);
    }
    for (int i = 0; i < 64; i++) pthread_join(th[i], nullptr);
    pthread_barrier_destroy(&emu_wave().bar);
    for (int k = 0; k < 4; k++) out4[k] = t.out[k];
}


// ---- the Faraday pair (heyvaerts_group.h) on the emulated wave ---------------------------------------------------
template <int KIND>
static void hey_group_lane_body(EmuGroupTask *t)
{
    __shared__ double s_tab[96];
    __shared__ double s_ginner[RIM_GROUP * RIM_ISTORE_DOUBLES(CAP_GINNER)];
    __shared__ HeyTask s_park[RIM_GROUP];
    __shared__ GroupParkBase s_gp;
    static double s_spill[SPILL_HEYGROUP_DOUBLES_PER_WAVE];
    const GKLane g = gk_lane_init(s_tab);
    if (g.lane == 0) { s_gp.ctr = WaveCounters{0, 0, 0}; s_gp.member_passes = 0; s_gp.stash_filed = 0; s_gp.stash_used = 0; s_gp.hb = nullptr; }
    wv_sync();
    HeyPoint hp;
    hp.s = uni(t->s);
    rim_sincos(t->theta, &hp.sin_th, &hp.cos_th);
    hey_point_derive(hp); hp.stokes = STOKES_Q;
    DistParams d;
    for (int k = 0; k < 5; k++) d.par[k] = t->par[k];
    dist_prepare<KIND>(d, t->norm);
    hey_point_endless<KIND>(hp, d);
    const HeyConsts hc = hey_consts();
    double vals[RIM_GROUP] = { 0, 0, 0, 0 };
    int stats[RIM_GROUP] = { 0, 0, 0, 0 };
    heyvaerts_group<KIND>(hp, d, hc, t->slots, t->nmem, g, s_ginner, s_spill,
                          s_spill + RIM_GROUP * RIM_ISTORE_DOUBLES(SPILL_GINNER), s_park, &s_gp, vals, stats);
    for (int m = 0; m < RIM_GROUP; m++) { t->vals[g.lane][m] = vals[m]; t->stats[g.lane][m] = stats[m]; }
    wv_sync();
    if (g.lane == 0) {
        t->samples = s_gp.ctr.samples; t->passes = s_gp.ctr.steps; t->inner_qags = s_gp.ctr.inner_qags;
        t->member_passes = s_gp.member_passes; t->stash_filed = s_gp.stash_filed;
    }
}

static void *hey_group_thread_main(void *p)
{
    ThreadArg *a = (ThreadArg *) p;
    EmuGroupTask *t = (EmuGroupTask *) (void *) a->t;
    emu_lane_ref() = a->lane;
    switch (t->kind) {
    case 0: hey_group_lane_body<0>(t); break;
    case 1: hey_group_lane_body<1>(t); break;
    case 2: hey_group_lane_body<2>(t); break;
    default: hey_group_lane_body<3>(t); break;
    }
    return nullptr;
}

extern "C" int emu_heyvaerts_group(int kind, unsigned slots, int nmem, double s, double theta, const double *par, double norm,
                                   double *vals, int *stats, unsigned long long *work4)
{
    static EmuGroupTask t;
    t.kind = kind; t.nmem = nmem; t.slots = slots; t.s = s; t.theta = theta; t.norm = norm;
    for (int k = 0; k < 5; k++) t.par[k] = par[k];
    pthread_barrier_init(&emu_wave().bar, nullptr, 64);
    pthread_t th[64];
    ThreadArg args[64];
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 1 << 20);
    for (int i = 0; i < 64; i++) {
        args[i].t = (EmuTask *) (void *) &t; args[i].lane = i;
        pthread_create(&th[i], &attr, hey_group_thread_main, &args[i]);
    }
    for (int i = 0; i < 64; i++) pthread_join(th[i], nullptr);
    pthread_barrier_destroy(&emu_wave().bar);
    int uniform = 1;
    for (int i = 1; i < 64; i++)
        for (int m = 0; m < nmem; m++)
            if (std::memcmp(&t.vals[i][m], &t.vals[0][m], 8) != 0 || t.stats[i][m] != t.stats[0][m]) uniform = 0;
    for (int m = 0; m < nmem; m++) { vals[m] = t.vals[0][m]; stats[m] = t.stats[0][m]; }
    work4[0] = t.samples; work4[1] = t.passes; work4[2] = t.inner_qags; work4[3] = t.member_passes; work4[4] = t.stash_filed;
    return uniform;
}
