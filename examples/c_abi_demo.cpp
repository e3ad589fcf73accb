// examples/c_abi_demo.cpp — the C ABI of include/neuralplane_amd.h used from a plain C++/HIP host: no Python, no PyTorch.
//
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I include examples/c_abi_demo.cpp -o c_abi_demo \
//       -Lneuralplane_amd/csrc -lneuralplane_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/neuralplane_amd/csrc -Wl,-rpath,/opt/rocm/lib
//   ./c_abi_demo neuralplane_amd/assets/f16_aero_mlp.bin [n] [steps] [out.bin]
//
// F-16 Heading with the constants of envs/configs/heading.yaml, reset + `steps` x np_f16_step with a fixed action pattern and
// the in-kernel counter RNG (seed 42).  Prints a checksum; with [out.bin] the final state s[12][n] is written so that a test
// can compare it with the same run through the Python surface (tests/test_gpu_step_parity.py).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "neuralplane_amd.h"

#define HIP_OK(x)                                                                  \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            return 2;                                                              \
        }                                                                          \
    } while (0)
#define NP_OK(x)                                                                   \
    do {                                                                           \
        if ((x) != 0) {                                                            \
            std::fprintf(stderr, "%s: %s\n", #x, np_last_error());                 \
            return 3;                                                              \
        }                                                                          \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s weights.bin [n] [steps] [out.bin]\n", argv[0]);
        return 1;
    }
    const int64_t n = argc > 2 ? std::atoll(argv[2]) : 4096;
    const int steps = argc > 3 ? std::atoi(argv[3]) : 50;
    std::vector<char> blob;
    {
        FILE *f = std::fopen(argv[1], "rb");
        if (!f) return 1;
        std::fseek(f, 0, SEEK_END);
        blob.resize((size_t)std::ftell(f));
        std::fseek(f, 0, SEEK_SET);
        if (std::fread(blob.data(), 1, blob.size(), f) != blob.size()) return 1;
        std::fclose(f);
    }
    if (np_abi_version() != NP_ABI_VERSION) return 1;

    np_f16_cfg cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.task = NP_TASK_HEADING;
    cfg.solver = NP_SOLVER_EULER;
    cfg.dt = 0.02; cfg.airspeed = 0; cfg.noise_scale = 0.01;
    cfg.altitude_limit = 2500.0; cfg.acceleration_limit = 300.0; cfg.max_velocity = 3; cfg.min_velocity = 0.01;
    cfg.min_alpha = -20; cfg.max_alpha = 45; cfg.min_beta = -30; cfg.max_beta = 30;
    cfg.max_check_interval = 2500; cfg.min_check_interval = 300;
    cfg.init_T = 2000; cfg.max_altitude = 20000; cfg.min_altitude = 19000; cfg.max_vt = 1200; cfg.min_vt = 1000;
    cfg.max_heading_increment = 3; cfg.max_pitch_increment = 0.3; cfg.max_velocities_u_increment = 300.0;
    cfg.max_distance = 2000; cfg.min_distance = 2000;

    np_f16_ctx *ctx = nullptr;
    NP_OK(np_f16_ctx_create(blob.data(), blob.size(), &cfg, 0, &ctx));

    float *s, *u, *tgt, *obs, *reward, *action, *cache;
    int64_t *step_count;
    uint8_t *flags[2];
    HIP_OK(hipMalloc(&s, sizeof(float) * 12 * n));
    HIP_OK(hipMalloc(&u, sizeof(float) * 5 * n));
    HIP_OK(hipMalloc(&tgt, sizeof(float) * 3 * n));
    HIP_OK(hipMalloc(&obs, sizeof(float) * 22 * n));
    HIP_OK(hipMalloc(&reward, sizeof(float) * n));
    HIP_OK(hipMalloc(&action, sizeof(float) * 4 * n));
    HIP_OK(hipMalloc(&cache, sizeof(float) * np_f16_cache_floats(n)));
    HIP_OK(hipMalloc(&step_count, sizeof(int64_t) * n));
    for (int k = 0; k < 2; k++) HIP_OK(hipMalloc(&flags[k], 3 * n));
    HIP_OK(hipMemset(s, 0, sizeof(float) * 12 * n));
    HIP_OK(hipMemset(u, 0, sizeof(float) * 5 * n));
    HIP_OK(hipMemset(tgt, 0, sizeof(float) * 3 * n));
    HIP_OK(hipMemset(step_count, 0, sizeof(int64_t) * n));
    HIP_OK(hipMemset(flags[0], 1, 3 * n));  // BaseEnv.__init__: all flags set -> the first reset initialises every row

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    np_f16_io io;
    std::memset(&io, 0, sizeof(io));
    io.s = s; io.u = u; io.tgt = tgt; io.ld = n; io.step_count = step_count;
    io.obs = obs; io.reward = reward; io.coef_cache = cache; io.seed = 42; io.row0 = 0;
    int cur = 0;
    auto bind_flags = [&]() {
        io.done_in = flags[cur]; io.bad_in = flags[cur] + n; io.timeout_in = flags[cur] + 2 * n;
        io.done_out = flags[1 - cur]; io.bad_out = flags[1 - cur] + n; io.timeout_out = flags[1 - cur] + 2 * n;
    };
    bind_flags();
    io.call_idx = 0;
    NP_OK(np_f16_reset(ctx, n, &io, stream));
    cur = 1 - cur;

    std::vector<float> a_host((size_t)4 * n);
    io.action = action; io.act_stride = 4;
    for (int t = 0; t < steps; t++) {
        for (int64_t i = 0; i < n; i++) {  // a deterministic action pattern the Python side can reproduce exactly
            a_host[4 * i + 0] = 0.5f + 0.25f * (float)((i + t) % 3);
            a_host[4 * i + 1] = 0.125f * (float)((i + 2 * t) % 5) - 0.25f;
            a_host[4 * i + 2] = 0.0625f * (float)((i * 3 + t) % 7) - 0.1875f;
            a_host[4 * i + 3] = 0.03125f * (float)((i + 5 * t) % 9) - 0.125f;
        }
        HIP_OK(hipMemcpyAsync(action, a_host.data(), sizeof(float) * 4 * n, hipMemcpyHostToDevice, stream));
        bind_flags();
        io.call_idx = (uint64_t)(t + 1);
        io.cache_valid = t > 0;
        NP_OK(np_f16_step(ctx, n, &io, stream));
        cur = 1 - cur;
        HIP_OK(hipStreamSynchronize(stream));  // a_host is reused
    }
    std::vector<float> s_host((size_t)12 * n), r_host((size_t)n);
    HIP_OK(hipMemcpy(s_host.data(), s, sizeof(float) * 12 * n, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(r_host.data(), reward, sizeof(float) * n, hipMemcpyDeviceToHost));
    double cs = 0.0, rs = 0.0;
    for (float v : s_host) cs += (double)v;
    for (float v : r_host) rs += (double)v;
    std::printf("C_ABI_DEMO n=%lld steps=%d state_checksum=%.6f reward_sum=%.6f\n", (long long)n, steps, cs, rs);
    if (argc > 4) {
        FILE *f = std::fopen(argv[4], "wb");
        if (!f || std::fwrite(s_host.data(), sizeof(float), s_host.size(), f) != s_host.size()) return 1;
        std::fclose(f);
    }
    np_f16_ctx_destroy(ctx);
    return 0;
}
