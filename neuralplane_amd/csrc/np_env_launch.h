// np_env_launch.h — the hand-over between the host side of np_f16_step / np_f16_reset (np_f16_kernels.hip::launch_env: argument checks,
// dispatch rule np_dispatch.h::env_choice, timing events) and the translation units that hold the instantiations of f16_env_kernel
// (np_env_t{task}s{solver}.hip <- np_env_tu.inc).  One unit per task x solver: the six compile side by side instead of one after the other.
#pragma once
#include <hip/hip_runtime.h>

#include "np_f16_kargs.h"

namespace npf16 {

struct EnvLaunch {
    KArgs a;
    unsigned grid, block;
    hipStream_t st;
    bool timed;                // start / stop attached to the dispatch itself (hipExtLaunchKernelGGL)
    hipEvent_t start, stop;
    bool step;                 // np_f16_step (true) or np_f16_reset
    bool inner;                // np_f16_io.inner_step (step only)
    bool cached;               // the cross-step coefficient cache is valid for this launch (step only)
    bool pair, pair3, latency, latency8, latency2, latency4w;   // np_dispatch.h::EnvChoice
};

// task = 0 heading, 1 control, 2 tracking; solver = 0 euler, 1 rk4 (3/8 rule).  A reset goes to the solver-0 unit of its task.
void env_dispatch_t0s0(const EnvLaunch &l);
void env_dispatch_t0s1(const EnvLaunch &l);
void env_dispatch_t1s0(const EnvLaunch &l);
void env_dispatch_t1s1(const EnvLaunch &l);
void env_dispatch_t2s0(const EnvLaunch &l);
void env_dispatch_t2s1(const EnvLaunch &l);

}  // namespace npf16
