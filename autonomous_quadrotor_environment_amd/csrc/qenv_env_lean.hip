// qenv_env_lean.hip -- the env kernels of the lean fp64 handle (QENV_COMPUTE_LEAN) (qenv_env_impl.hpp).
#define QENV_ENV_UNIT_LEAN
#include "qenv_env_impl.hpp"

namespace qenv {
QENV_ENV_INSTANCES(, double, double, true)
}  // namespace qenv

#ifdef QENV_FUSED_STAMPS
extern "C" int qenv_debug_fused_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qenv::g_fused_stamps), sizeof(unsigned long long) * (size_t)n);
}
#endif

#ifdef QENV_ROLLOUT_STAMPS
extern "C" int qenv_debug_rollout_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qenv::g_rollout_stamps), sizeof(unsigned long long) * (size_t)n);
}
extern "C" int qenv_debug_rollout_phase(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qenv::g_rollout_phase), sizeof(unsigned long long) * (size_t)n);
}
#endif
