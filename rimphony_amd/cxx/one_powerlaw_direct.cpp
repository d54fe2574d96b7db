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
    // the DistributionFunction trait as pitchy_pl.rs:203-238 uses it: norm = 1, analytic vs forward difference
    PitchyPowerLawDistribution ppd(2.7, 1.3);
    ppd.norm = 1.;
    const double g = 5.5, cx = 0.4, eps = 1e-6;
    const auto d = ppd.calc_f_derivatives(*ctx, g, cx);
    const double f0 = ppd.calc_f(*ctx, g, cx);
    const double ndg = (ppd.calc_f(*ctx, g + eps, cx) - f0) / eps, ndc = (ppd.calc_f(*ctx, g, cx + eps) - f0) / eps;
    std::printf("dfdg %.10e (numeric %.10e)  dfdcx %.10e (numeric %.10e)\n", d[0], ndg, d[1], ndc);
    const bool deriv_ok = std::fabs((d[0] - ndg) / ndg) < 1e-4 && std::fabs((d[1] - ndc) / ndc) < 1e-4;
    // the batched entries: one context, the multi-device entry over {that context} (degenerate sharding), work counters
    std::vector<double> s, th;
    std::vector<std::vector<double>> par(4);
    for (int i = 0; i < 24; i++) {
        s.push_back(3. + 7. * i); th.push_back(0.2 + 0.05 * i);
        par[0].push_back(2.2 + 0.05 * i); par[1].push_back(1.); par[2].push_back(1e12); par[3].push_back(1e10);
    }
    std::vector<int32_t> st;
    std::vector<uint64_t> work;
    const auto one = BatchCalculator(ctx, RIMPHONY_POWER_LAW).compute(s, th, par, 0x0f, &st, &work);
    const auto multi = BatchCalculator::compute_multi({ctx}, RIMPHONY_POWER_LAW, s, th, par, 0x0f);
    bool batch_ok = true;
    for (size_t i = 0; i < one.size(); i++) {
        const bool selected = (i % 8) < 4;
        batch_ok = batch_ok && (selected ? (one[i] == multi[i] && std::isfinite(one[i]) && work[i] > 0) : (std::isnan(one[i]) && work[i] == 0));
    }
    bool threw = false;
    try { BatchCalculator(ctx, RIMPHONY_POWER_LAW).compute(s, th, {par[0]}, 0x01); } catch (const std::runtime_error &) { threw = true; }
    std::printf("batch entries: %s, misuse throws: %s, device shared with another context: %s\n", batch_ok ? "ok" : "MISMATCH",
                threw ? "yes" : "NO", ctx->shared_mode() ? "yes" : "no");
    if (!batch_ok || !threw) return 1;
    return std::fabs(ji / 2.64399749412774e-21 - 1.) < 1e-3 && std::fabs(rho_q - 1.8e-9) < 0.05e-9 && deriv_ok ? 0 : 1;
}
