// rimphony_diag.hip -- the two diagnostic slices of lib.rs:254-298 that need kernels of their own
// (diagnostic_symphony_n_integral, diagnostic_symphony_gamma_contribution) and the DistributionFunction seam
// (calc_f / calc_f_derivatives over arrays).  See rimphony_internal.h for why
// they are not in rimphony_hip.hip.
#include "rimphony_internal.h"
#include "heyvaerts_wave.h"

using namespace rim;

// diagnostic_symphony_n_integral (lib.rs:254-260): one outer QAG over n in [n_lo, n_hi] of the gamma-integral,
// run through the same post / evaluate / consume steps as a chunk of n_integration.
template <int KIND>
__global__ __launch_bounds__(64) void n_integral_kernel(PointArgs pa, const double *norm_ptr, size_t count,
                                                        const double *n_lo, const double *n_hi, double *out, double *spill_base)
{
    __shared__ double s_tab[96];
    __shared__ double s_inner[RIM_ISTORE_DOUBLES(CAP_INNER)];
    __shared__ double s_outer[RIM_ISTORE_DOUBLES(CAP_OUTER)];
    __shared__ TaskState s_park;
    const GKLane g = gk_lane_init(s_tab);
    double *spill = spill_base + (size_t) blockIdx.x * SPILL_DOUBLES_PER_WAVE;
    const IStore inner = istore_carve(s_inner, CAP_INNER, spill, SPILL_INNER);
    const IStore outer = istore_carve(s_outer, CAP_OUTER, spill + RIM_ISTORE_DOUBLES(SPILL_INNER), SPILL_OUTER);
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
        TaskState T;
        sym_begin(pt, T);
        T.phase = PH_QAG_FIRST;
        T.lobe = pa.negative_lobe;
        T.qa = uni(n_lo[i]);
        T.qb = uni(n_hi[i]);
        T.ni_failed = 0;
        qag_begin(T.oq, 0., 1e-3, 1000);
        while (T.phase == PH_QAG_FIRST || T.phase == PH_QAG_BISECT) {
            SymBatch B;
            if (!sym_post(pt, g, outer, T, B)) break;
            __syncthreads();
            if (g.lane == 0) s_park = T;
            int batch_status = 0;
            double gval = 0.;
            unsigned long long mask = wv_ballot(B.req_active);
            while (mask) {
                const int k = __builtin_ffsll((long long) mask) - 1;
                mask &= mask - 1;
                const double n = readlane_d(B.req_n, k);
                const int lb = wv_readlane(B.req_lobe, k);
                const double val = sym_eval_request<KIND>(pt, d, g, inner, &s_qpark, n, lb, batch_status);
                if (g.lane == k) gval = val;
            }
            __syncthreads();
            T = s_park;
            task_uniformize(T);
            sym_consume(pt, g, outer, T, B, gval, uni(batch_status));
        }
        // sym_consume has run the chunk's epilogue: contrib holds the QAG value, ni_failed an Err
        if (g.lane == 0) out[i] = (T.ni_failed || (T.status & ST_OUTER_FAIL)) ? RIM_NAN : T.contrib;
    }
}

// Unit seam for gsl::deriv_central (gsl.rs:233-257 -> gsl_deriv_central) as n_integration drives it
// (symphony.rs:238-240): the derivative of the gamma-integral with respect to the harmonic number at n_start[i], step
// h = 1e-10 n_start.  One wave per abscissa runs the PH_DERIV1 (and, if the first estimate asks for it, PH_DERIV2) phases
// of the coefficient's state machine -- the code the product kernels run -- and reports dr_r0, the accepted estimate.
template <int KIND>
__global__ __launch_bounds__(64) void deriv_probe_kernel(PointArgs pa, const double *norm_ptr, size_t count,
                                                         const double *n_start, double *out, double *spill_base)
{
    __shared__ double s_tab[96];
    __shared__ double s_inner[RIM_ISTORE_DOUBLES(CAP_INNER)];
    __shared__ double s_outer[RIM_ISTORE_DOUBLES(CAP_OUTER)];
    __shared__ TaskState s_park;
    const GKLane g = gk_lane_init(s_tab);
    double *spill = spill_base + (size_t) blockIdx.x * SPILL_DOUBLES_PER_WAVE;
    const IStore inner = istore_carve(s_inner, CAP_INNER, spill, SPILL_INNER);
    const IStore outer = istore_carve(s_outer, CAP_OUTER, spill + RIM_ISTORE_DOUBLES(SPILL_INNER), SPILL_OUTER);
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
        TaskState T;
        sym_begin(pt, T);
        sym_start_lobe(pt, T, pa.negative_lobe);
        T.n_start = uni(n_start[i]);
        T.dr_h = 1e-10 * T.n_start;
        T.phase = PH_DERIV1;
        while (T.phase == PH_DERIV1 || T.phase == PH_DERIV2) {
            SymBatch B;
            if (!sym_post(pt, g, outer, T, B)) break;
            __syncthreads();
            if (g.lane == 0) s_park = T;
            int batch_status = 0;
            double gval = 0.;
            unsigned long long mask = wv_ballot(B.req_active);
            while (mask) {
                const int k = __builtin_ffsll((long long) mask) - 1;
                mask &= mask - 1;
                const double n = readlane_d(B.req_n, k);
                const int lb = wv_readlane(B.req_lobe, k);
                const double val = sym_eval_request<KIND>(pt, d, g, inner, &s_qpark, n, lb, batch_status);
                if (g.lane == k) gval = val;
            }
            __syncthreads();
            T = s_park;
            task_uniformize(T);
            sym_consume(pt, g, outer, T, B, gval, uni(batch_status));
        }
        if (g.lane == 0) out[i] = T.dr_r0;
    }
}

// diagnostic_symphony_gamma_contribution (lib.rs:288-296 -> symphony.rs:491-567): all harmonics n at fixed gamma.
// One wave per gamma; the integrand is sampled in n, so every lane carries its own order data (registers/scratch,
// as in integrand_kernel_n).  Discrete sums are accumulated by lane 0 in the reference's order.
template <int KIND>
__global__ __launch_bounds__(64) void gamma_contribution_kernel(PointArgs pa, const double *norm_ptr, size_t count,
                                                                const double *gammas, double *out, double *spill_base)
{
    __shared__ double s_tab[96];
    __shared__ double s_store[RIM_ISTORE_DOUBLES(CAP_INNER)];
    __shared__ double s_buf[64];
    const GKLane g = gk_lane_init(s_tab);
    const IStore st = istore_carve(s_store, CAP_INNER, spill_base + (size_t) blockIdx.x * SPILL_DOUBLES_PER_WAVE, SPILL_INNER);
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
        const double gamma = uni(gammas[i]);
        const double delta = rim_fabs(pt.cos_th) * rim_sqrt(gamma * gamma - 1.);
        const long long n_minus = sat_i64(pt.s * (gamma - delta) + 1.);
        const long long n_plus = sat_i64(pt.s * (gamma + delta));
        const bool fully_discrete = n_plus - n_minus < 1000;
        const long long disc_end = fully_discrete ? n_plus + 1 : n_minus + 31;     // exclusive
        double ans = 0.;
        for (long long base = n_minus; base < disc_end; base += 64) {
            const long long nn = base + g.lane;
            double v = 0.;
            if (nn < disc_end) {
                LeungOrder ord[2];
                const SymOrder so = sym_order((double) nn, ord);
                v = gamma_integrand<KIND>(pt, d, so, gamma);
            }
            __syncthreads();
            s_buf[g.lane] = v;
            __syncthreads();
            const long long left = disc_end - base;
            const int cnt = left < 64 ? (int) left : 64;
            for (int k = 0; k < cnt; k++) ans += s_buf[k];
        }
        if (!fully_discrete) {
            auto f = [&](double n, bool active) -> double {
                if (!active) return 0.;
                LeungOrder ord[2];
                const SymOrder so = sym_order(n, ord);
                return gamma_integrand<KIND>(pt, d, so, gamma);
            };
            QagState q;
            wave_qag(f, g, st, (double) (n_minus + 31), (double) n_plus, 0., 1e-3, 5000, q, &s_qpark);
            ans += (q.status == QAG_SUCCESS) ? q.result : RIM_NAN;
        }
        double res = RIM_NAN;
        if (rim_isfinite(ans)) {
            const double tpe = RIM_TWO_PI * RIM_ELECTRON_CHARGE;
            const double acos_th = rim_fabs(pt.cos_th);
            res = (pt.coeff == COEFF_EMISSION) ? ans * ((tpe * tpe) / (RIM_SPEED_LIGHT * acos_th))
                                               : ans * (-1. * (tpe * tpe) / (2. * RIM_MASS_ELECTRON * RIM_SPEED_LIGHT * acos_th));
        }
        if (g.lane == 0) out[i] = res;
    }
}

extern "C" int rimphony_n_integral_batch_device(rimphony_ctx *c, int kind, const double *params,
                                                int coeff, int stokes, int negative_lobe, double s, double theta,
                                                size_t count, const double *d_n_lo, const double *d_n_hi, double *d_out,
                                                void *stream)
{
    if (!c || (count && (!d_n_lo || !d_n_hi || !d_out))) return RIMPHONY_EINVAL;
    hipStream_t st = (hipStream_t) stream;
    RimCtxScope scope(c, st);          // the context's lock + ordering behind earlier work on its workspace
    { const int rc0 = scope.enter(); if (rc0) return rc0; }
    PointArgs pa;
    int rc = rim_point_setup(c, kind, params, coeff, stokes, negative_lobe, s, theta, st, pa);
    if (rc) return rc;
    if (count == 0) return RIMPHONY_OK;
    unsigned grid = 0;
    rc = rim_wave_grid(c, count, 16, &grid);
    if (rc) return rc;
    const double *norm = rim_ctx_norm(c);
    double *spill = rim_ctx_spill(c);
    switch (kind) {
    case 0: hipLaunchKernelGGL(n_integral_kernel<0>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, count, d_n_lo, d_n_hi, d_out, spill); break;
    case 1: hipLaunchKernelGGL(n_integral_kernel<1>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, count, d_n_lo, d_n_hi, d_out, spill); break;
    case 2: hipLaunchKernelGGL(n_integral_kernel<2>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, count, d_n_lo, d_n_hi, d_out, spill); break;
    default: hipLaunchKernelGGL(n_integral_kernel<3>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, count, d_n_lo, d_n_hi, d_out, spill); break;
    }
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}

extern "C" int rimphony_deriv_probe_batch_device(rimphony_ctx *c, int kind, const double *params,
                                                 int coeff, int stokes, int negative_lobe, double s, double theta,
                                                 size_t count, const double *d_n_start, double *d_out, void *stream)
{
    if (!c || (count && (!d_n_start || !d_out))) return RIMPHONY_EINVAL;
    hipStream_t st = (hipStream_t) stream;
    RimCtxScope scope(c, st);          // the context's lock + ordering behind earlier work on its workspace
    { const int rc0 = scope.enter(); if (rc0) return rc0; }
    PointArgs pa;
    int rc = rim_point_setup(c, kind, params, coeff, stokes, negative_lobe, s, theta, st, pa);
    if (rc) return rc;
    if (count == 0) return RIMPHONY_OK;
    unsigned grid = 0;
    rc = rim_wave_grid(c, count, 16, &grid);
    if (rc) return rc;
    const double *norm = rim_ctx_norm(c);
    double *spill = rim_ctx_spill(c);
    switch (kind) {
    case 0: hipLaunchKernelGGL(deriv_probe_kernel<0>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, count, d_n_start, d_out, spill); break;
    case 1: hipLaunchKernelGGL(deriv_probe_kernel<1>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, count, d_n_start, d_out, spill); break;
    case 2: hipLaunchKernelGGL(deriv_probe_kernel<2>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, count, d_n_start, d_out, spill); break;
    default: hipLaunchKernelGGL(deriv_probe_kernel<3>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, count, d_n_start, d_out, spill); break;
    }
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}

extern "C" int rimphony_gamma_contribution_batch_device(rimphony_ctx *c, int kind, const double *params, int coeff, int stokes,
                                                        double s, double theta, size_t count, const double *d_gamma,
                                                        double *d_out, void *stream)
{
    if (!c || (count && (!d_gamma || !d_out))) return RIMPHONY_EINVAL;
    hipStream_t st = (hipStream_t) stream;
    RimCtxScope scope(c, st);          // the context's lock + ordering behind earlier work on its workspace
    { const int rc0 = scope.enter(); if (rc0) return rc0; }
    PointArgs pa;
    int rc = rim_point_setup(c, kind, params, coeff, stokes, 0, s, theta, st, pa);
    if (rc) return rc;
    if (count == 0) return RIMPHONY_OK;
    unsigned grid = 0;
    rc = rim_wave_grid(c, count, 8, &grid);
    if (rc) return rc;
    const double *norm = rim_ctx_norm(c);
    double *spill = rim_ctx_spill(c);
    switch (kind) {
    case 0: hipLaunchKernelGGL(gamma_contribution_kernel<0>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, count, d_gamma, d_out, spill); break;
    case 1: hipLaunchKernelGGL(gamma_contribution_kernel<1>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, count, d_gamma, d_out, spill); break;
    case 2: hipLaunchKernelGGL(gamma_contribution_kernel<2>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, count, d_gamma, d_out, spill); break;
    default: hipLaunchKernelGGL(gamma_contribution_kernel<3>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, count, d_gamma, d_out, spill); break;
    }
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}

// ---- DistributionFunction::calc_f / calc_f_derivatives (lib.rs:111-146) over arrays -----------------------
// One thread per (gamma, cos_xi) pair of ONE distribution.  norm_override: NaN = the distribution's own
// normalisation (full_calculation), anything else = that value (the reference's derivative tests set norm = 1,
// pitchy_pl.rs:216-217, pitchy_kappa.rs:149-150).
template <int KIND>
__global__ void calc_f_kernel(PointArgs pa, const double *norm_ptr, double norm_override, size_t count,
                              const double *gamma, const double *cos_xi, double *f, double *dfdg, double *dfdcx)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    DistParams d;
    for (int k = 0; k < 5; k++) d.par[k] = pa.par[k];
    dist_prepare<KIND>(d, rim_isnan(norm_override) ? norm_ptr[0] : norm_override);
    if (f) f[i] = calc_f<KIND>(d, gamma[i], cos_xi[i]);
    if (dfdg || dfdcx) {
        double a, b;
        calc_f_derivatives<KIND>(d, gamma[i], cos_xi[i], a, b);
        if (dfdg) dfdg[i] = a;
        if (dfdcx) dfdcx[i] = b;
    }
}

extern "C" int rimphony_calc_f_batch_device(rimphony_ctx *c, int kind, const double *params, double norm_override,
                                            size_t count, const double *d_gamma, const double *d_cos_xi,
                                            double *d_f, double *d_dfdg, double *d_dfdcx, void *stream)
{
    if (!c || (count && (!d_gamma || !d_cos_xi))) return RIMPHONY_EINVAL;
    hipStream_t st = (hipStream_t) stream;
    RimCtxScope scope(c, st);          // the context's lock + ordering behind earlier work on its workspace
    { const int rc0 = scope.enter(); if (rc0) return rc0; }
    PointArgs pa;
    // coefficient / stokes / s / theta play no part in f; any valid values satisfy the argument checks
    int rc = rim_point_setup(c, kind, params, 0, 0, 0, 1., 1., st, pa);
    if (rc) return rc;
    if (count == 0) return RIMPHONY_OK;
    const double *norm = rim_ctx_norm(c);
    const dim3 grid((unsigned) ((count + 63) / 64)), block(64);
    switch (kind) {
    case 0: hipLaunchKernelGGL(calc_f_kernel<0>, grid, block, RIM_DYN_LDS, st, pa, norm, norm_override, count, d_gamma, d_cos_xi, d_f, d_dfdg, d_dfdcx); break;
    case 1: hipLaunchKernelGGL(calc_f_kernel<1>, grid, block, RIM_DYN_LDS, st, pa, norm, norm_override, count, d_gamma, d_cos_xi, d_f, d_dfdg, d_dfdcx); break;
    case 2: hipLaunchKernelGGL(calc_f_kernel<2>, grid, block, RIM_DYN_LDS, st, pa, norm, norm_override, count, d_gamma, d_cos_xi, d_f, d_dfdg, d_dfdcx); break;
    default: hipLaunchKernelGGL(calc_f_kernel<3>, grid, block, RIM_DYN_LDS, st, pa, norm, norm_override, count, d_gamma, d_cos_xi, d_f, d_dfdg, d_dfdcx); break;
    }
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}

// host-buffer form of the same (what a host-language mirror of the trait calls)
extern "C" int rimphony_calc_f_batch(rimphony_ctx *c, int kind, const double *params, double norm_override, size_t count,
                                     const double *gamma, const double *cos_xi, double *f, double *dfdg, double *dfdcx)
{
    if (!c || (count && (!gamma || !cos_xi))) return RIMPHONY_EINVAL;
    if (count == 0) return rimphony_calc_f_batch_device(c, kind, params, norm_override, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    double *buf = nullptr;
    if (hipMalloc(&buf, 5 * count * sizeof(double)) != hipSuccess) return RIMPHONY_ENOMEM;
    int rc = RIMPHONY_EHIP;
    do {
        if (hipMemcpy(buf, gamma, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(buf + count, cos_xi, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) break;
        rc = rimphony_calc_f_batch_device(c, kind, params, norm_override, count, buf, buf + count, buf + 2 * count,
                                          buf + 3 * count, buf + 4 * count, nullptr);
        if (rc) break;
        rc = RIMPHONY_EHIP;
        if (hipDeviceSynchronize() != hipSuccess) break;
        if (f && hipMemcpy(f, buf + 2 * count, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) break;
        if (dfdg && hipMemcpy(dfdg, buf + 3 * count, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) break;
        if (dfdcx && hipMemcpy(dfdcx, buf + 4 * count, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = RIMPHONY_OK;
    } while (0);
    (void) hipFree(buf);
    return rc;
}


// ---- Heyvaerts unit seams (heyvaerts.rs:213-296 outer integrands, 302-468 elements) ---------------------------
// The Faraday coefficients are compared with the oracle end to end; these two seams let a divergence be localised:
// the element functions (one thread per (fixed, v) pair) and the outer integrands (one wave per abscissa: the inner
// QAG of hey_eval_request).
static __device__ HeyPoint hey_point_of(const PointArgs &pa)
{
    HeyPoint pt;
    pt.s = pa.s;
    rim_sincos(pa.theta, &pt.sin_th, &pt.cos_th);
    hey_point_derive(pt);
    pt.stokes = pa.stokes;
    return pt;
}

template <int KIND>
__global__ void hey_element_kernel(PointArgs pa, const double *norm_ptr, int qr, size_t count,
                                   const double *fixed, const double *v, double *out)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const HeyPoint pt = hey_point_of(pa);
    DistParams d;
    for (int k = 0; k < 5; k++) d.par[k] = pa.par[k];
    dist_prepare<KIND>(d, norm_ptr[0]);
    const HeyConsts hc = hey_consts();
    out[i] = hey_element<KIND>(pt, d, hc, qr != 0, fixed[i], v[i]);
}

template <int KIND>
__global__ __launch_bounds__(64) void hey_outer_kernel(PointArgs pa, const double *norm_ptr, int qr,
                                                       size_t count, const double *u, double *out, double *spill_base)
{
    __shared__ double s_tab[96];
    __shared__ double s_inner[RIM_ISTORE_DOUBLES(CAP_INNER)];
    __shared__ QagParkBase s_qpark;
    const GKLane g = gk_lane_init(s_tab);
    const IStore inner = istore_carve(s_inner, CAP_INNER, spill_base + (size_t) blockIdx.x * SPILL_DOUBLES_PER_WAVE, SPILL_INNER);
    if (threadIdx.x == 0) { s_qpark.ctr = WaveCounters{0, 0, 0}; s_qpark.hb = nullptr; }
    __syncthreads();
    HeyPoint pt = hey_point_of(pa);
    pt.s = uni(pt.s); pt.sin_th = uni(pt.sin_th); pt.cos_th = uni(pt.cos_th);
    pt.sigma0 = uni(pt.sigma0); pt.sigma0_sq = uni(pt.sigma0_sq); pt.dinv = uni(pt.dinv);
    DistParams d;
    for (int k = 0; k < 5; k++) d.par[k] = pa.par[k];
    dist_prepare<KIND>(d, norm_ptr[0]);
    const HeyConsts hc = hey_consts();
    for (size_t i = blockIdx.x; i < count; i += gridDim.x) {
        int st = 0;
        const double val = hey_eval_request<KIND>(pt, d, hc, g, inner, &s_qpark, uni(u[i]), qr, st);
        if (g.lane == 0) out[i] = val;
    }
}


extern "C" int rimphony_hey_element_batch_device(rimphony_ctx *c, int kind, const double *params, int stokes, double s, double theta,
                                                 int qr, size_t count, const double *d_fixed, const double *d_v, double *d_out,
                                                 void *stream)
{
    if (!c || (count && (!d_fixed || !d_v || !d_out)) || (stokes != RIMPHONY_STOKES_Q && stokes != RIMPHONY_STOKES_V)) return RIMPHONY_EINVAL;
    hipStream_t st = (hipStream_t) stream;
    RimCtxScope scope(c, st);          // the context's lock + ordering behind earlier work on its workspace
    { const int rc0 = scope.enter(); if (rc0) return rc0; }
    PointArgs pa;
    int rc = rim_point_setup(c, kind, params, 0, stokes, 0, s, theta, st, pa);
    if (rc) return rc;
    if (count == 0) return RIMPHONY_OK;
    const double *norm = rim_ctx_norm(c);
    const dim3 grid((unsigned) ((count + 63) / 64)), block(64);
    switch (kind) {
    case 0: hipLaunchKernelGGL(hey_element_kernel<0>, grid, block, RIM_DYN_LDS, st, pa, norm, qr, count, d_fixed, d_v, d_out); break;
    case 1: hipLaunchKernelGGL(hey_element_kernel<1>, grid, block, RIM_DYN_LDS, st, pa, norm, qr, count, d_fixed, d_v, d_out); break;
    case 2: hipLaunchKernelGGL(hey_element_kernel<2>, grid, block, RIM_DYN_LDS, st, pa, norm, qr, count, d_fixed, d_v, d_out); break;
    default: hipLaunchKernelGGL(hey_element_kernel<3>, grid, block, RIM_DYN_LDS, st, pa, norm, qr, count, d_fixed, d_v, d_out); break;
    }
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}

extern "C" int rimphony_hey_outer_batch_device(rimphony_ctx *c, int kind, const double *params, int stokes, double s, double theta,
                                               int qr, size_t count, const double *d_u, double *d_out, void *stream)
{
    if (!c || (count && (!d_u || !d_out)) || (stokes != RIMPHONY_STOKES_Q && stokes != RIMPHONY_STOKES_V)) return RIMPHONY_EINVAL;
    hipStream_t st = (hipStream_t) stream;
    RimCtxScope scope(c, st);          // the context's lock + ordering behind earlier work on its workspace
    { const int rc0 = scope.enter(); if (rc0) return rc0; }
    PointArgs pa;
    int rc = rim_point_setup(c, kind, params, 0, stokes, 0, s, theta, st, pa);
    if (rc) return rc;
    if (count == 0) return RIMPHONY_OK;
    unsigned grid = 0;
    rc = rim_wave_grid(c, count, 16, &grid);
    if (rc) return rc;
    const double *norm = rim_ctx_norm(c);
    switch (kind) {
    case 0: hipLaunchKernelGGL(hey_outer_kernel<0>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, qr, count, d_u, d_out, rim_ctx_spill(c)); break;
    case 1: hipLaunchKernelGGL(hey_outer_kernel<1>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, qr, count, d_u, d_out, rim_ctx_spill(c)); break;
    case 2: hipLaunchKernelGGL(hey_outer_kernel<2>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, qr, count, d_u, d_out, rim_ctx_spill(c)); break;
    default: hipLaunchKernelGGL(hey_outer_kernel<3>, dim3(grid), dim3(64), RIM_DYN_LDS, st, pa, norm, qr, count, d_u, d_out, rim_ctx_spill(c)); break;
    }
    HIP_TRY(hipGetLastError());
    return RIMPHONY_OK;
}
