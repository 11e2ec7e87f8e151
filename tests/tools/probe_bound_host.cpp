// Host build of csrc/qenv_probe_bound.hpp for tests/test_host_cpu.py::test_probe_bound_is_conservative:
//   g++ -O2 -std=c++17 -shared -fPIC -o probe_bound.so tests/tools/probe_bound_host.cpp
#include "../../autonomous_quadrotor_environment_amd/csrc/qenv_probe_bound.hpp"
extern "C" {
// plant[9] = F_abs, omrj_abs, cmj_xy, cmj_z, kd, inv_m, cx, cy, cz
double pb_sum(const double* y, const double* f0, const double* isc, const double* plant, double h_max) {
    qenv::ProbePlant<double> p{plant[0], plant[1], plant[2], plant[3], plant[4], plant[5], plant[6], plant[7], plant[8]};
    return qenv::probe_bound_sum<double>(y, f0, isc, p, h_max);
}
// the float instantiation (the f32 and mixed handles run it): inputs rounded to float, the arithmetic in float
static void to_f(const double* a, float* b, int n) { for (int i = 0; i < n; ++i) b[i] = (float)a[i]; }
int pb_skippable_f(const double* y, const double* f0, const double* isc, const double* plant, double h_max, double d1, double thr) {
    float yf[13], ff[13], sf[13], pf[9];
    to_f(y, yf, 13); to_f(f0, ff, 13); to_f(isc, sf, 13); to_f(plant, pf, 9);
    qenv::ProbePlant<float> p{pf[0], pf[1], pf[2], pf[3], pf[4], pf[5], pf[6], pf[7], pf[8]};
    return qenv::probe_skippable<float>(yf, ff, sf, p, (float)h_max, (float)d1, (float)thr) ? 1 : 0;
}
int pb_skippable(const double* y, const double* f0, const double* isc, const double* plant, double h_max, double d1, double thr) {
    qenv::ProbePlant<double> p{plant[0], plant[1], plant[2], plant[3], plant[4], plant[5], plant[6], plant[7], plant[8]};
    return qenv::probe_skippable<double>(y, f0, isc, p, h_max, d1, thr) ? 1 : 0;
}
}
