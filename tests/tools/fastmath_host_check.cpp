// Host check of csrc/qenv_fastmath.hpp (the lean fp64 handle's elementary functions) against libm, with seeds of the
// accuracy the hardware delivers (v_rcp_f64 / v_rsq_f64: 2^-23).  Prints the worst errors as one JSON line;
// tests/test_host_cpu.py::test_lean_fastmath_accuracy asserts them.    g++ -O2 -std=c++17 -ffp-contract=off
#include <cmath>
#include <cstdio>
#include <random>
#include "../../autonomous_quadrotor_environment_amd/csrc/qenv_fastmath.hpp"
struct HostSeed {
    static double rcp_seed(double x) { int e; double m = frexp(x, &e); return ldexp((double)(1.0f / (float)m), -e); }
    static double rsq_seed(double x) {
        int e; double m = frexp(x, &e);
        if (e & 1) { m *= 2; e -= 1; }
        return ldexp((double)(1.0f / sqrtf((float)m)), -e / 2);
    }
};
namespace f = qenv::fm;
int main() {
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> U(-1, 1);
    double w_at = 0, w_as = 0, w_s = 0, w_c = 0, w_t = 0, w_div = 0, w_sq = 0, w_rs = 0;
    for (int i = 0; i < 2000000; ++i) {
        double y = U(g) * pow(10, 3 * U(g)), x = U(g) * pow(10, 3 * U(g));
        if (i % 7 == 0) y = x * (0.41421356237 + 1e-6 * U(g));   // around the octant split
        if (i % 11 == 0) x = y * 1e-9 * U(g);                      // near the axes
        w_at = fmax(w_at, fabs(f::atan2<HostSeed>(y, x) - ::atan2(y, x)));
        double s = U(g);
        if (i % 5 == 0) s = copysign(1 - 1e-6 * fabs(U(g)), s);   // near +-1
        if (i % 13 == 0) s = 0.5 + 1e-9 * U(g);                    // around the branch point
        w_as = fmax(w_as, fabs(f::asin<HostSeed>(s) - ::asin(s)));
        double th = 8 * U(g), sn, cs;
        f::sincos<HostSeed>(th, sn, cs);
        w_s = fmax(w_s, fabs(sn - ::sin(th))); w_c = fmax(w_c, fabs(cs - ::cos(th)));
        double tt = 1.5 * U(g);
        w_t = fmax(w_t, fabs(f::tan<HostSeed>(tt) - ::tan(tt)) / fmax(1.0, fabs(::tan(tt))));
        double p = pow(10, 6 * U(g));
        w_div = fmax(w_div, fabs(f::div<HostSeed>(y, p) - y / p) / fabs(y / p));
        w_sq = fmax(w_sq, fabs(f::sqrt<HostSeed>(p) - ::sqrt(p)) / ::sqrt(p));
        w_rs = fmax(w_rs, fabs(f::rsqrt<HostSeed>(p) * ::sqrt(p) - 1));
    }
    const bool edges = f::atan2<HostSeed>(0.0, 0.0) == 0.0 && fabs(f::atan2<HostSeed>(0.0, -0.0) - M_PI) < 1e-15 &&
                       fabs(f::atan2<HostSeed>(-0.0, -1.0) + M_PI) < 1e-15 && fabs(f::atan2<HostSeed>(1.0, 0.0) - M_PI / 2) < 1e-15 &&
                       fabs(f::asin<HostSeed>(1.0) - M_PI / 2) < 1e-15 && std::isnan(f::asin<HostSeed>(1.0000001)) &&
                       f::sqrt<HostSeed>(0.0) == 0.0 && std::isnan(f::sqrt<HostSeed>(-1.0));
    printf("{\"atan2_abs\": %.3g, \"asin_abs\": %.3g, \"sin_abs\": %.3g, \"cos_abs\": %.3g, \"tan_rel\": %.3g, \"div_rel\": %.3g, "
           "\"sqrt_rel\": %.3g, \"rsqrt_rel\": %.3g, \"edges\": %s}\n", w_at, w_as, w_s, w_c, w_t, w_div, w_sq, w_rs, edges ? "true" : "false");
    return 0;
}
