// The reference's examples/one-powerlaw-direct.rs against the C++ mirror:
//   g++ -std=c++17 one_powerlaw_direct.cpp -L.. -lrimphony_hip -Wl,-rpath,'$ORIGIN/..' -o one_powerlaw_direct
#include <cstdio>
#include "rimphony.hpp"
int main()
{
    using namespace rimphony;
    auto ctx = std::make_shared<Context>(0);
    const double ji = PowerLawDistribution(2.5).gamma_limits(1., 1e12, 1e10).full_calculation(ctx)
                          .compute_cgs(Coefficient::Emission, Stokes::I, 1e9, 1e3, 1., 0.9);
    std::printf("Symphony j_I: %e   Ours: %e\n", 2.64399749412774e-21, ji);
    // the high-frequency closed form of power_law.rs:146-149 ("for gamma_min = 10, I get 1.8e-9")
    const double rho_q = PowerLawDistribution(2.5).gamma_limits(10., 1e12, 1e10).high_freq_approximation(ctx)
                             .compute_dimensionless(Coefficient::Faraday, Stokes::Q, 1e4, 0.78539816339744831);
    std::printf("HF rho_Q: %.17g\n", rho_q);
    return std::fabs(ji / 2.64399749412774e-21 - 1.) < 1e-3 && std::fabs(rho_q - 1.8e-9) < 0.05e-9 ? 0 : 1;
}
