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
enum HeyPhase { HP_DERIV1 = 0, HP_DERIV2, HP_QAG_FIRST, HP_QAG_BISECT };

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

template <int KIND>
__device__ __forceinline__ double heyvaerts_coefficient(const HeyPoint &pt, const DistParams &dist, const HeyConsts &hc,
                                                        const GKLane &g, const IStore &inner, const IStore &outer,
                                                        HeyTask *park, QagPark *qpark, int &status_out)
{
    const int lane = g.lane;
    const double TOL = 1e-5, DELTA_SCALE_FACTOR = 5.;

    HeyTask T;
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

    // start the outer QAG of the current stage's next chunk
    auto start_chunk = [&](double a, double b) {
        T.qa = a; T.qb = b;
        qag_begin(T.oq, 0., 1e-3, 4096);
        T.phase = HP_QAG_FIRST;
    };
    auto start_deriv = [&](double x) {
        T.dr_x = x;
        T.dr_h = 1e-6;
        T.phase = HP_DERIV1;
    };
    auto fail = [&]() { T.failed = 1; T.stage = HS_DONE; };

    // head of the marching loop of the current stage (the `while keep_going` tests of heyvaerts.rs:102,134,161)
    auto loop_head = [&]() {
        if (++T.steps > RIM_HEY_MAX_STEPS) { T.status |= ST_CHUNK_CAP; fail(); return; }
        if (T.stage == HS_NR_RIGHT) {
            if (T.nr_val != 0.) start_deriv(T.pomega_right);
            else start_chunk(T.pomega_right, T.pomega_right + T.delta_right);
        } else if (T.stage == HS_NR_LEFT) {
            start_deriv(T.pomega_left);
        } else {   // HS_QR
            if (T.qr_val != 0.) start_deriv(T.sigma_low);
            else start_chunk(T.sigma_low, T.sigma_low + T.delta_sigma);
        }
    };

    while (T.stage != HS_DONE) {
        if (++T.batches > RIM_MAX_BATCHES) { T.status |= ST_CHUNK_CAP; fail(); break; }

        // ---- 1. post a batch of outer abscissae, one per lane -------------------
        double req_u = 0.;
        bool req_active = false;
        const int phase = T.phase;
        const bool qr = T.stage == HS_QR;

        if (phase == HP_DERIV1 || phase == HP_DERIV2) {
            if (lane < 4) {
                const double h = T.dr_h, x = T.dr_x;
                req_u = (lane == 0) ? x - h : (lane == 1) ? x + h : (lane == 2) ? x - h / 2 : x + h / 2;
                req_active = true;
            }
        } else if (phase == HP_QAG_FIRST) {
            const double center = 0.5 * (T.qa + T.qb);
            const double hl = 0.5 * (T.qb - T.qa);
            req_u = center + hl * g.t;
            req_active = g.node && g.half == 0;
        } else {
            qag_pick(T.oq, outer, lane);
            const double la = g.half ? T.oq.a2 : T.oq.a1;
            const double lb = g.half ? T.oq.b2 : T.oq.b1;
            const double center = 0.5 * (la + lb);
            const double hl = 0.5 * (lb - la);
            req_u = center + hl * g.t;
            req_active = g.node;
        }

        if (lane == 0) {
            *park = T;
            if (qpark->hb) {
                hb_store(qpark->hb + 1, (unsigned long long) T.batches);
                hb_store(qpark->hb + 2, (unsigned long long) (100 * T.stage + phase));
                hb_store(qpark->hb + 5, (unsigned long long) T.steps);
            }
        }
        int batch_status = 0;

        // ---- 2. evaluate the requests: nr_outer_integrand / qr_outer_integrand -------
        double gval = 0.;
        {
            unsigned long long mask = wv_ballot(req_active);
            while (mask) {
                const int k = __builtin_ffsll((long long) mask) - 1;
                mask &= mask - 1;
                const double u = readlane_d(req_u, k);
                double lo, hi;
                bool empty = false;
                if (!qr) {
                    const double sigma_min = rim_sqrt(u * u + pt.sigma0_sq);
                    const double sigma_max = RIM_INVERSE_SQRT_3 * rim_pow(sigma_min, 1.5);
                    empty = sigma_max <= sigma_min;
                    lo = sigma_min; hi = sigma_max;
                } else {
                    const double pomega_max_phys = rim_sqrt(RIM_THREE_TWO_THIRDS * rim_pow(u, 4. / 3.) - pt.sigma0_sq);
                    const double pomega_max_qr = rim_sqrt(u * u - pt.sigma0_sq);
                    const double pomega_max = rust_min(pomega_max_phys, pomega_max_qr);
                    lo = -pomega_max; hi = pomega_max;
                }
                lo = uni(lo); hi = uni(hi);
                double val = 0.;
                if (!uni(empty)) {
                    auto f = [&](double v, bool active) -> double {
                        return active ? hey_element<KIND>(pt, dist, hc, qr, u, v) : 0.;
                    };
                    QagState iq;
                    wave_qag(f, g, inner, lo, hi, 0., 1e-3, 4096, iq, qpark);
                    if (lane == 0) qpark->ctr.inner_qags += 1;
                    val = uni(iq.result);
                    if (uni(iq.status) != QAG_SUCCESS) {
                        val = RIM_NAN;
                        batch_status |= ST_INNER_FAIL;
                        if (iq.status == QAG_ESTORE) batch_status |= ST_STORE_FULL;
                    }
                }
                if (lane == k) gval = val;
            }
        }

        wv_sync();
        T = *park;
        hey_uniformize(T);
        T.status |= uni(batch_status);

        // ---- 3. continuation ---------------------------------------------------------
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
                    continue;
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
                if (rel_deriv == 0. || rim_fabs(1. / (rel_deriv * T.delta_right)) > DELTA_SCALE_FACTOR)
                    T.delta_right *= DELTA_SCALE_FACTOR;
                start_chunk(T.pomega_right, T.pomega_right + T.delta_right);
            } else if (T.stage == HS_NR_LEFT) {
                if (rel_deriv == 0. || rim_fabs(1. / (rel_deriv * T.delta_left)) > DELTA_SCALE_FACTOR)
                    T.delta_left *= DELTA_SCALE_FACTOR;
                start_chunk(T.pomega_left - T.delta_left, T.pomega_left);
            } else {
                if (rel_deriv == 0. || rim_fabs(1. / (rel_deriv * T.delta_sigma)) > DELTA_SCALE_FACTOR) {
                    if (T.delta_sigma < 1e6 * pt.sigma0) T.delta_sigma *= DELTA_SCALE_FACTOR;
                }
                start_chunk(T.sigma_low, T.sigma_low + T.delta_sigma);
            }
            continue;
        }

        bool chunk_done;
        if (phase == HP_QAG_FIRST) {
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
        if (!chunk_done) { T.phase = HP_QAG_BISECT; continue; }

        // .unwrap_or(NAN) of the outer integral (heyvaerts.rs:204-211, 253-260)
        double contrib = T.oq.result;
        if (T.oq.status != QAG_SUCCESS) {
            contrib = RIM_NAN;
            T.status |= ST_OUTER_FAIL;
            if (T.oq.status == QAG_ESTORE) T.status |= ST_STORE_FULL;
        }
        if (rim_isnan(contrib)) { fail(); break; }

        if (T.stage == HS_NR_FIRST) {
            T.nr_val = contrib;
            T.stage = HS_NR_RIGHT;
            T.steps = 0;
            loop_head();
            continue;
        }
        bool keep_going = true;
        if (T.stage == HS_NR_RIGHT) {
            if (T.nr_val != 0.) keep_going = rim_fabs(contrib / T.nr_val) > TOL;
            T.nr_val += contrib;
            T.pomega_right += T.delta_right;
            if (!keep_going) { T.stage = HS_NR_LEFT; T.steps = 0; }
            loop_head();
        } else if (T.stage == HS_NR_LEFT) {
            keep_going = rim_fabs(contrib / T.nr_val) > TOL;
            T.nr_val += contrib;
            T.pomega_left -= T.delta_left;
            if (!keep_going) {
                T.stage = HS_QR;
                T.steps = 0;
                T.qr_val = 0.;
                T.sigma_low = rust_max(pt.sigma0, RIM_INVERSE_SQRT_3 * rim_pow(pt.sigma0, 1.5));
                T.delta_sigma = pt.sigma0;
            }
            loop_head();
        } else {   // HS_QR
            if (T.qr_val != 0.) keep_going = rim_fabs(contrib / T.qr_val) > TOL;
            T.qr_val += contrib;
            T.sigma_low += T.delta_sigma;
            if (!keep_going) {
                const double ssin = pt.s * pt.sin_th;
                T.result = 2. * (RIM_ELECTRON_CHARGE * RIM_ELECTRON_CHARGE) * (T.nr_val + T.qr_val)
                    / (RIM_MASS_ELECTRON * (ssin * ssin));
                T.stage = HS_DONE;
            } else {
                loop_head();
            }
        }
    }

    status_out = T.status;
    if (T.failed) {
        status_out |= ST_NONFINITE;
        return RIM_NAN;
    }
    if (rim_isnan(T.result)) status_out |= ST_NONFINITE;
    return T.result;
}

}  // namespace rim
#endif
