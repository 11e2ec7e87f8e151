// qenv_env_mix.hip -- the env kernels of the mixed handle: fp64 rows, fp32 arithmetic (qenv_env_impl.hpp).
#define QENV_ENV_UNIT_MIX
#include "qenv_env_impl.hpp"

namespace qenv {
QENV_ENV_INSTANCES(, float, double, false)
}  // namespace qenv
