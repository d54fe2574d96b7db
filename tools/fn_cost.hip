// Static cost of the leaf functions of the Symphony integrand: one kernel per function, compiled to ISA by
// tools/fn_cost.sh and counted by tools/fn_cost.py (vector instructions, fp64 arithmetic among them).  Not part of
// the product; a measuring aid for DESIGN.md section 5's per-pass budget.
#include <hip/hip_runtime.h>
#include "dev_symphony.h"
using namespace rim;

#define K(name, expr) extern "C" __global__ void k_##name(const double *in, double *out) { \
    const double a = in[threadIdx.x], b = in[64 + threadIdx.x], c = in[128 + threadIdx.x]; (void) b; (void) c; out[threadIdx.x] = (expr); }

K(empty, a)
K(pow_third, rim_pow_normal(a, 1. / 3.))
K(log, rim_log(a))
K(log_normal, rim_log_normal(a))
K(log10_region, rim_log10_region(a))
K(exp_bounded, rim_exp_bounded(a))
K(exp, rim_exp(a))
K(pow, rim_pow(a, b))
K(pow_normal, rim_pow_normal(a, b))
K(sqrt, rim_sqrt(a))
K(div, a / b)
K(div_moderate, rim_div_moderate(a, b))
K(acos01, rim_acos01(a))
K(atan_pos, rim_atan_pos(a))
K(exp_factor, exp_factor<0>(a, b))
extern "C" __global__ void k_meissel_first(const double *in, double *out)
{
    const LeungOrder o = leung_order(in[200]);     // wave-uniform order
    out[threadIdx.x] = meissel_first<0>(o, in[threadIdx.x]);
}
extern "C" __global__ void k_debye_pair(const double *in, double *out)
{
    double r0, r1;
    debye_eps_pair<0>(in[200], in[201], in[threadIdx.x], &r0, &r1);
    out[threadIdx.x] = r0; out[64 + threadIdx.x] = r1;
}
extern "C" __global__ void k_select_code(const double *in, double *out)
{
    const LeungOrder o = leung_order(in[200]);
    double pos;
    const int code = leung_select_code(o, in[threadIdx.x], pos);
    out[threadIdx.x] = pos + code;
}
extern "C" __global__ void k_bessel_pair(const double *in, double *out)
{
    __shared__ LeungOrder store[2];
    const SymOrder so = sym_order(in[200], store);
    double jn, djn;
    sym_bessel_pair<0>(so, in[threadIdx.x], jn, djn);
    out[threadIdx.x] = jn; out[64 + threadIdx.x] = djn;
}
template <int KIND> __device__ void dist_of(const double *in, DistParams &d)
{
    for (int k = 0; k < 5; k++) d.par[k] = in[192 + k];
    dist_prepare<KIND>(d, in[199]);
}
#define KF(kind) \
extern "C" __global__ void k_calc_f_##kind(const double *in, double *out) { DistParams d; dist_of<kind>(in, d); \
    out[threadIdx.x] = calc_f<kind, 0>(d, in[threadIdx.x], in[64 + threadIdx.x]); } \
extern "C" __global__ void k_calc_fd_##kind(const double *in, double *out) { DistParams d; dist_of<kind>(in, d); double g, c; \
    calc_f_derivatives<kind, 0>(d, in[threadIdx.x], in[64 + threadIdx.x], g, c); out[threadIdx.x] = g; out[64 + threadIdx.x] = c; }
KF(0) KF(1) KF(2) KF(3)

// ---- Faraday leaf functions ----
#include "dev_heyvaerts.h"
__device__ HeyPoint pt_of(const double *in)
{
    HeyPoint pt;
    pt.s = in[210]; pt.sin_th = in[211]; pt.cos_th = in[212]; pt.sigma0 = in[213]; pt.sigma0_sq = in[214]; pt.stokes = (int) in[215];
    return pt;
}
extern "C" __global__ void k_fill_coord(const double *in, double *out)
{
    const HeyCoord c = fill_coord_vars(pt_of(in), in[threadIdx.x], in[64 + threadIdx.x]);
    out[threadIdx.x] = c.x + c.gamma + c.mu;
}
#define KH(kind) \
extern "C" __global__ void k_h_nr_##kind(const double *in, double *out) { DistParams d; dist_of<kind>(in, d); const HeyPoint pt = pt_of(in); \
    HeyCoord c; c.sigma = in[threadIdx.x]; c.pomega = in[64 + threadIdx.x]; c.x = in[128 + threadIdx.x]; c.gamma = in[256 + threadIdx.x]; c.mu = in[320 + threadIdx.x]; \
    out[threadIdx.x] = h_nr_element<kind>(pt, d, c); } \
extern "C" __global__ void k_f_nr_##kind(const double *in, double *out) { DistParams d; dist_of<kind>(in, d); const HeyPoint pt = pt_of(in); \
    HeyCoord c; c.sigma = in[threadIdx.x]; c.pomega = in[64 + threadIdx.x]; c.x = in[128 + threadIdx.x]; c.gamma = in[256 + threadIdx.x]; c.mu = in[320 + threadIdx.x]; \
    out[threadIdx.x] = f_nr_element<kind>(pt, d, c); } \
extern "C" __global__ void k_dfdsigma_##kind(const double *in, double *out) { DistParams d; dist_of<kind>(in, d); const HeyPoint pt = pt_of(in); \
    HeyCoord c; c.sigma = in[threadIdx.x]; c.pomega = in[64 + threadIdx.x]; c.x = in[128 + threadIdx.x]; c.gamma = in[256 + threadIdx.x]; c.mu = in[320 + threadIdx.x]; \
    out[threadIdx.x] = dfdsigma<kind>(pt, d, c); }
KH(0) KH(1) KH(2)
extern "C" __global__ void k_nr_common(const double *in, double *out)
{
    HeyCoord c; c.sigma = in[threadIdx.x]; c.pomega = in[64 + threadIdx.x]; c.x = in[128 + threadIdx.x]; c.gamma = 0; c.mu = 0;
    double a1, a2, xa1p, ssq, xsq;
    nr_common(c, a1, a2, xa1p, ssq, xsq);
    out[threadIdx.x] = a1 + a2 + xa1p + ssq + xsq;
}
K(pow15, rim_pow15(a))
K(pow25, rim_pow25(a))
K(full_div_same_divisor3, a / c + b / c + (a * b) / c)
