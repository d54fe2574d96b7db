/* rimo_heyvaerts.c -- placeholder until the Heyvaerts Faraday integrator is restated. */
#include "rimo.h"
#include "rimo_math.h"
double rimo_heyvaerts(const rimo_dist *d, int coeff, int stokes, double s, double theta, rimo_counters *c)
{
    (void) d; (void) coeff; (void) stokes; (void) s; (void) theta; (void) c;
    return RIM_NAN;
}
