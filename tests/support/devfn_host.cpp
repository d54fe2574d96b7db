// Host build of the scalar __host__ __device__ functions of the HIP path, for
// the CPU ("not gpu") tests only: it lets the op-order parity between the
// device functions and the oracle be checked bit for bit without a GPU.
// Not part of the product library.
#include "../../rimphony_amd/csrc/dev_symphony.h"
#include "../../rimphony_amd/csrc/dev_heyvaerts.h"
using namespace rim;
extern "C" {
double devh_bessel_j(double n, double x) { return bessel_j(n, x); }
double devh_bessel_dj(double n, double x) { return bessel_dj(n, x); }
double devh_gamma_integrand(int kind, int coeff, int stokes, double s, double cos_th, double sin_th,
                            const double *par, double norm, double n, double gamma)
{
    SymPoint pt{s, cos_th, sin_th, coeff, stokes};
    DistParams d;
    for (int i = 0; i < 5; i++) d.par[i] = par[i];
    LeungOrder ord[2];
    SymOrder so = sym_order(n, ord);
    switch (kind) {
    case 0: dist_prepare<0>(d, norm); return gamma_integrand<0>(pt, d, so, gamma);
    case 1: dist_prepare<1>(d, norm); return gamma_integrand<1>(pt, d, so, gamma);
    case 2: dist_prepare<2>(d, norm); return gamma_integrand<2>(pt, d, so, gamma);
    default: dist_prepare<3>(d, norm); return gamma_integrand<3>(pt, d, so, gamma);
    }
}
void devh_sincos(double x, double *s, double *c) { rim_sincos(x, s, c); }
double devh_hey_element(int kind, int stokes, double s, double cos_th, double sin_th, const double *par, double norm,
                        int qr, double fixed, double v)
{
    HeyPoint pt;
    pt.s = s; pt.cos_th = cos_th; pt.sin_th = sin_th; hey_point_derive(pt);
    pt.stokes = stokes;
    DistParams d;
    for (int i = 0; i < 5; i++) d.par[i] = par[i];
    const HeyConsts hc = hey_consts();
    switch (kind) {
    case 0: dist_prepare<0>(d, norm); return hey_element<0>(pt, d, hc, qr != 0, fixed, v);
    case 1: dist_prepare<1>(d, norm); return hey_element<1>(pt, d, hc, qr != 0, fixed, v);
    case 2: dist_prepare<2>(d, norm); return hey_element<2>(pt, d, hc, qr != 0, fixed, v);
    default: dist_prepare<3>(d, norm); return hey_element<3>(pt, d, hc, qr != 0, fixed, v);
    }
}
}
