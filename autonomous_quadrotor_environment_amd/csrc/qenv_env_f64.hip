// qenv_env_f64.hip -- the env kernels of the fp64 handle (qenv_env_impl.hpp).
#define QENV_ENV_UNIT_F64
#include "qenv_env_impl.hpp"

namespace qenv {
QENV_ENV_INSTANCES(, double, double, false)
}  // namespace qenv
