// rimphony.hpp -- C++ host-side mirror of rimphony's public API over the C ABI of
// include/rimphony_hip.h (header-only).  The reference's host language is Rust,
// which this image lacks; this header keeps the same names, argument meaning and
// error behaviour so a user of the crate finds what they expect:
//
//   enum Stokes / Coefficient                                   lib.rs:74-107
//   trait SynchrotronCalculator { compute_dimensionless, compute_cgs,
//       compute_all_dimensionless, compute_all_cgs }           lib.rs:150-210
//   PowerLawDistribution::new(p).gamma_limits(..).full_calculation(..)   power_law.rs:71-111
//   ThermalJuettnerDistribution::new(t).full_calculation(..)             thermal_juettner.rs:45-72
//   PitchyPowerLawDistribution::new(p,k).gamma_limits(..)...             pitchy_pl.rs:73-115
//   PitchyKappaDistribution::new(kappa,width,k).gamma_cutoff(..)...      pitchy_kappa.rs:70-125
//   trait DistributionFunction { calc_f, calc_f_derivatives }            lib.rs:111-146
//
// plus the batched entry the GPU path exists for: BatchCalculator::compute().
// Numerical failure is NaN, never an exception (symphony.rs:115-117); API misuse
// and HIP errors throw std::runtime_error.  No CPU fallback.
#ifndef RIMPHONY_HPP
#define RIMPHONY_HPP

#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rimphony_hip.h"

namespace rimphony {

constexpr double PI = 3.14159265358979323846;
constexpr double TWO_PI = 2. * PI;
constexpr double MASS_ELECTRON = 9.1093826e-28;
constexpr double SPEED_LIGHT = 2.99792458e10;
constexpr double ELECTRON_CHARGE = 4.80320680e-10;

enum class Stokes { I = 0, Q = 1, V = 2 };
enum class Coefficient { Emission = 0, Absorption = 1, Faraday = 2 };

enum class Precision { F64 = RIMPHONY_PRECISION_F64, F32Integrand = RIMPHONY_PRECISION_F32_INTEGRAND };

inline void check(int rc, const char *what)
{
    if (rc == RIMPHONY_OK) return;
    std::string msg = std::string(what) + ": " + rimphony_strerror(rc);
    // the HIP call that failed (thread-local text): only a RIMPHONY_EHIP return is described by it
    const char *detail = rc == RIMPHONY_EHIP ? rimphony_last_error() : nullptr;
    if (detail && *detail) msg += std::string(" -- ") + detail;
    throw std::runtime_error(msg);
}

class Context {
public:
    explicit Context(int device = 0) { check(rimphony_ctx_create(device, &ctx_), "rimphony_ctx_create"); }
    ~Context() { rimphony_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    rimphony_ctx *get() const { return ctx_; }
    // true when another context (this process or another) already owned the device at creation: the persistent grids are
    // then a quarter of the device and the cooperative tail is off (include/rimphony_hip.h, "context")
    bool shared_mode() const { return rimphony_ctx_shared_mode(ctx_) != 0; }
private:
    rimphony_ctx *ctx_ = nullptr;
};

// slot order of compute_all_dimensionless (lib.rs:176-177)
inline int slot_of(Coefficient c, Stokes s)
{
    if (c == Coefficient::Faraday) return s == Stokes::Q ? 6 : s == Stokes::V ? 7 : -1;
    return 2 * static_cast<int>(s) + static_cast<int>(c);
}

// N x (full_calculation + compute_all_dimensionless): host SoA arrays in, [n][8] out.
class BatchCalculator {
public:
    BatchCalculator(std::shared_ptr<Context> ctx, int dist_kind) : ctx_(std::move(ctx)), kind_(dist_kind)
    {
        if (rimphony_dist_nparams(kind_) < 0) throw std::runtime_error("unknown distribution kind");
    }
    // status: optional [n][8] RIMPHONY_ST_* words; work: optional [n][8] integrand samples spent per coefficient
    std::vector<double> compute(const std::vector<double> &s, const std::vector<double> &theta,
                                const std::vector<std::vector<double>> &params, uint32_t mask = RIMPHONY_SLOTS_ALL,
                                std::vector<int32_t> *status = nullptr, std::vector<uint64_t> *work = nullptr,
                                Precision precision = Precision::F64) const
    {
        const size_t n = s.size();
        const std::vector<const double *> pp = columns(kind_, n, theta, params);
        std::vector<double> out(n * 8, std::numeric_limits<double>::quiet_NaN());
        if (status) status->assign(n * 8, 0);
        if (work) work->assign(n * 8, 0);
        check(rimphony_batch_compute_ex(ctx_->get(), kind_, n, s.data(), theta.data(), pp.data(), mask,
                                        static_cast<int>(precision), out.data(), status ? status->data() : nullptr,
                                        work ? work->data() : nullptr), "rimphony_batch_compute_ex");
        return out;
    }
    // The same table from several devices: row i is evaluated by contexts[i mod N] (one host thread per context) and
    // lands in row i -- the table does not depend on N.
    static std::vector<double> compute_multi(const std::vector<std::shared_ptr<Context>> &contexts, int dist_kind,
                                             const std::vector<double> &s, const std::vector<double> &theta,
                                             const std::vector<std::vector<double>> &params,
                                             uint32_t mask = RIMPHONY_SLOTS_ALL, std::vector<int32_t> *status = nullptr,
                                             std::vector<uint64_t> *work = nullptr, Precision precision = Precision::F64)
    {
        const size_t n = s.size();
        const std::vector<const double *> pp = columns(dist_kind, n, theta, params);
        std::vector<rimphony_ctx *> raw;
        for (const auto &c : contexts) raw.push_back(c->get());
        std::vector<double> out(n * 8, std::numeric_limits<double>::quiet_NaN());
        if (status) status->assign(n * 8, 0);
        if (work) work->assign(n * 8, 0);
        check(rimphony_batch_compute_multi(raw.data(), static_cast<int>(raw.size()), dist_kind, n, s.data(), theta.data(),
                                           pp.data(), mask, static_cast<int>(precision), out.data(),
                                           status ? status->data() : nullptr, work ? work->data() : nullptr),
              "rimphony_batch_compute_multi");
        return out;
    }
private:
    static std::vector<const double *> columns(int kind, size_t n, const std::vector<double> &theta,
                                               const std::vector<std::vector<double>> &params)
    {
        if (rimphony_dist_nparams(kind) < 0) throw std::runtime_error("unknown distribution kind");
        if (theta.size() != n || static_cast<int>(params.size()) != rimphony_dist_nparams(kind))
            throw std::runtime_error("compute: array shapes do not match");
        std::vector<const double *> pp;
        for (const auto &p : params) {
            if (p.size() != n) throw std::runtime_error("compute: parameter array length");
            pp.push_back(p.data());
        }
        return pp;
    }
    std::shared_ptr<Context> ctx_;
    int kind_;
};

// trait SynchrotronCalculator for one parameter point (each call is a 1-point batch)
class FullSynchrotronCalculator {
public:
    FullSynchrotronCalculator(std::shared_ptr<Context> ctx, int kind, std::vector<double> params)
        : batch_(std::move(ctx), kind), params_(std::move(params)) {}

    double compute_dimensionless(Coefficient coeff, Stokes stokes, double s, double theta) const
    {
        const int k = slot_of(coeff, stokes);
        if (k < 0) return std::numeric_limits<double>::quiet_NaN();    // (Faraday, I): lib.rs:239-240
        return run(s, theta, 1u << k)[k];
    }
    double compute_cgs(Coefficient coeff, Stokes stokes, double nu, double b, double n_e, double theta) const
    {
        const double nu_c = ELECTRON_CHARGE * b / (TWO_PI * MASS_ELECTRON * SPEED_LIGHT);
        const double val = compute_dimensionless(coeff, stokes, nu / nu_c, theta);
        return coeff == Coefficient::Emission ? val * n_e * nu : val * n_e / nu;
    }
    std::array<double, 8> compute_all_dimensionless(double s, double theta) const
    {
        const auto v = run(s, theta, RIMPHONY_SLOTS_ALL);
        std::array<double, 8> rv;
        for (int k = 0; k < 8; k++) rv[k] = v[k];
        return rv;
    }
    std::array<double, 8> compute_all_cgs(double nu, double b, double n_e, double theta) const
    {
        const double nu_c = ELECTRON_CHARGE * b / (TWO_PI * MASS_ELECTRON * SPEED_LIGHT);
        auto rv = compute_all_dimensionless(nu / nu_c, theta);
        for (int k = 0; k < 8; k++) rv[k] = (k == 0 || k == 2 || k == 4) ? rv[k] * n_e * nu : rv[k] * n_e / nu;
        return rv;
    }
private:
    std::vector<double> run(double s, double theta, uint32_t mask) const
    {
        std::vector<std::vector<double>> pp;
        for (double p : params_) pp.push_back({p});
        return batch_.compute({s}, {theta}, pp, mask);
    }
    BatchCalculator batch_;
    std::vector<double> params_;
};

// `high_freq_approximation()` of the reference (power_law.rs:117-170, thermal_juettner.rs:78-142): the closed-form
// (Faraday, Q) and (Faraday, V); every other coefficient is NaN.
class HighFrequencyApproximation {
public:
    HighFrequencyApproximation(std::shared_ptr<Context> ctx, int kind, std::vector<double> params)
        : ctx_(std::move(ctx)), kind_(kind), params_(std::move(params)) {}
    double compute_dimensionless(Coefficient coeff, Stokes stokes, double s, double theta) const
    {
        if (coeff != Coefficient::Faraday || stokes == Stokes::I) return std::numeric_limits<double>::quiet_NaN();
        std::vector<const double *> pp;
        for (const double &p : params_) pp.push_back(&p);
        double out[2];
        check(rimphony_highfreq_batch(ctx_->get(), kind_, 1, &s, &theta, pp.data(), out), "rimphony_highfreq_batch");
        return stokes == Stokes::Q ? out[0] : out[1];
    }
private:
    std::shared_ptr<Context> ctx_;
    int kind_;
    std::vector<double> params_;
};

// The DistributionFunction trait (lib.rs:111-146): calc_f and calc_f_derivatives, evaluated by the HIP library.
// `norm` mirrors the public field of the reference's structs (its tests overwrite it, pitchy_pl.rs:216-217): NaN
// (the default) means "the normalisation full_calculation computes".
class DistributionFunction {
public:
    virtual ~DistributionFunction() = default;
    double norm = std::numeric_limits<double>::quiet_NaN();
    double calc_f(const Context &ctx, double gamma, double cos_xi) const
    {
        double f;
        const std::vector<double> p = abi_params();
        check(rimphony_calc_f_batch(ctx.get(), abi_kind(), p.data(), norm, 1, &gamma, &cos_xi, &f, nullptr, nullptr),
              "rimphony_calc_f_batch");
        return f;
    }
    std::array<double, 2> calc_f_derivatives(const Context &ctx, double gamma, double cos_xi) const
    {
        std::array<double, 2> d;
        const std::vector<double> p = abi_params();
        check(rimphony_calc_f_batch(ctx.get(), abi_kind(), p.data(), norm, 1, &gamma, &cos_xi, nullptr, &d[0], &d[1]),
              "rimphony_calc_f_batch");
        return d;
    }
protected:
    virtual int abi_kind() const = 0;
    virtual std::vector<double> abi_params() const = 0;
};

class PowerLawDistribution : public DistributionFunction {
protected:
    int abi_kind() const override { return RIMPHONY_POWER_LAW; }
    std::vector<double> abi_params() const override { return {p_, gmin_, gmax_, gcut_}; }
public:
    explicit PowerLawDistribution(double p) : p_(p) {}
    PowerLawDistribution &gamma_limits(double gmin, double gmax, double gcut) { gmin_ = gmin; gmax_ = gmax; gcut_ = gcut; return *this; }
    FullSynchrotronCalculator full_calculation(std::shared_ptr<Context> ctx) const
    { return FullSynchrotronCalculator(std::move(ctx), RIMPHONY_POWER_LAW, {p_, gmin_, gmax_, gcut_}); }
    HighFrequencyApproximation high_freq_approximation(std::shared_ptr<Context> ctx) const
    { return HighFrequencyApproximation(std::move(ctx), RIMPHONY_POWER_LAW, {p_, gmin_}); }
private:
    double p_, gmin_ = 1., gmax_ = 1e12, gcut_ = 1e10;     // defaults: power_law.rs:71-79
};

class ThermalJuettnerDistribution : public DistributionFunction {
protected:
    int abi_kind() const override { return RIMPHONY_THERMAL_JUETTNER; }
    std::vector<double> abi_params() const override { return {t_}; }
public:
    explicit ThermalJuettnerDistribution(double t) : t_(t) {}
    FullSynchrotronCalculator full_calculation(std::shared_ptr<Context> ctx) const
    { return FullSynchrotronCalculator(std::move(ctx), RIMPHONY_THERMAL_JUETTNER, {t_}); }
    HighFrequencyApproximation high_freq_approximation(std::shared_ptr<Context> ctx) const
    { return HighFrequencyApproximation(std::move(ctx), RIMPHONY_THERMAL_JUETTNER, {t_}); }
private:
    double t_;
};

class PitchyPowerLawDistribution : public DistributionFunction {
protected:
    int abi_kind() const override { return RIMPHONY_PITCHY_PL; }
    std::vector<double> abi_params() const override { return {p_, k_, gmin_, gmax_, gcut_}; }
public:
    PitchyPowerLawDistribution(double p, double k) : p_(p), k_(k) {}
    PitchyPowerLawDistribution &gamma_limits(double gmin, double gmax, double gcut) { gmin_ = gmin; gmax_ = gmax; gcut_ = gcut; return *this; }
    FullSynchrotronCalculator full_calculation(std::shared_ptr<Context> ctx) const
    { return FullSynchrotronCalculator(std::move(ctx), RIMPHONY_PITCHY_PL, {p_, k_, gmin_, gmax_, gcut_}); }
private:
    double p_, k_, gmin_ = 1., gmax_ = 1e12, gcut_ = 1e10;
};

class PitchyKappaDistribution : public DistributionFunction {
protected:
    int abi_kind() const override { return RIMPHONY_PITCHY_KAPPA; }
    std::vector<double> abi_params() const override { return {kappa_, width_, k_, gcut_}; }
public:
    PitchyKappaDistribution(double kappa, double width, double k) : kappa_(kappa), width_(width), k_(k) {}
    PitchyKappaDistribution &gamma_cutoff(double gcut) { gcut_ = gcut; return *this; }
    FullSynchrotronCalculator full_calculation(std::shared_ptr<Context> ctx) const
    { return FullSynchrotronCalculator(std::move(ctx), RIMPHONY_PITCHY_KAPPA, {kappa_, width_, k_, gcut_}); }
private:
    double kappa_, width_, k_, gcut_ = 1e10;
};

}  // namespace rimphony
#endif
