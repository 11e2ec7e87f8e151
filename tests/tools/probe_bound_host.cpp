// Host build of csrc/qenv_probe_bound.hpp for tests/test_host_cpu.py::test_probe_bound_is_conservative:
//   g++ -O2 -std=c++17 -shared -fPIC -o probe_bound.so tests/tools/probe_bound_host.cpp
#include "../../autonomous_quadrotor_environment_amd/csrc/qenv_probe_bound.hpp"
extern "C" {
// plant[9] = F_abs, omrj_abs, cmj_xy, cmj_z, kd, inv_m, cx, cy, cz
double pb_sum(const double* y, const double* f0, const double* isc, const double* plant, double h_max) {
    qenv::ProbePlant<double> p{plant[0], plant[1], plant[2], plant[3], plant[4], plant[5], plant[6], plant[7], plant[8]};
    return qenv::probe_bound_sum<double>(y, f0, isc, p, h_max);
}
int pb_skippable(const double* y, const double* f0, const double* isc, const double* plant, double h_max, double d1, double thr) {
    qenv::ProbePlant<double> p{plant[0], plant[1], plant[2], plant[3], plant[4], plant[5], plant[6], plant[7], plant[8]};
    return qenv::probe_skippable<double>(y, f0, isc, p, h_max, d1, thr) ? 1 : 0;
}
}
