/* rimo_highfreq.c -- CPU oracle (test infrastructure) for the closed-form high-frequency Faraday
 * approximations: power_law.rs:133-170 and thermal_juettner.rs:94-142 of the reference
 * (`high_freq_approximation()`; (Faraday, Q) and (Faraday, V) only, everything else NaN).
 * The formulas live in rimphony_amd/csrc/highfreq.h, shared with the HIP kernel; tests/test_highfreq.py
 * pins them against an independent numpy/scipy transcription. */
#include "rimo.h"
#include "rimo_math.h"
#include "../rimphony_amd/csrc/highfreq.h"

/* out[0] = rho_Q, out[1] = rho_V (dimensionless); kind 0 = power law {p, gamma_min, ..}, 1 = thermal {T} */
int rimo_highfreq(int kind, const double *params, double s, double theta, double out[2])
{
    double sn, cs;
    m_sincos(theta, &sn, &cs);
    if (kind == 0) {
        out[0] = rim_hf_powerlaw_faraday_q(params[0], params[1], s, sn);
        out[1] = rim_hf_powerlaw_faraday_v(params[0], params[1], s, sn);
        return 0;
    }
    if (kind == 1) {
        rim_hf_thermal_faraday(params[0], s, sn, cs, &out[0], &out[1]);
        return 0;
    }
    out[0] = out[1] = RIM_NAN;
    return -1;
}

void rimo_bessel_k012(double x, double k[3]) { rim_bessel_k012(x, &k[0], &k[1], &k[2]); }
