// np_env_t0s1.hip — the instantiations of f16_env_kernel (np_f16_env_kernel.h; reference: envs/env_base.py:83-109) for task 0
// (0 heading, 1 control, 2 tracking) and solver 1 (0 euler, 1 rk4): a translation unit of its own so that the six build side by side.
#define NP_ENV_TASK 0
#define NP_ENV_SOLVER 1
#include "np_env_tu.inc"
