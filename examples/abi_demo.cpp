// A consumer of libmrsim's C ABI without Python or torch: device buffers from the HIP runtime, one
// MR_Env episode of N envs through mrsim_reset / mrsim_step, then the same episode through ONE
// mrsim_rollout launch, and a check that both give the same bits and the reference's episode shape
// (51 steps, reward 10 per step, done by timeout: MR_env.py:62,89,136-152).
//
// Build: make -C mr_rl_amd/csrc demo      Run: examples/abi_demo [n_envs]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mrsim.h"

#define HIP_OK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            return 2;                                                                          \
        }                                                                                      \
    } while (0)
#define SIM_OK(x)                                                                          \
    do {                                                                                   \
        int r_ = (x);                                                                      \
        if (r_ != MRSIM_OK) {                                                              \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, mrsim_strerror(r_)); \
            return 3;                                                                      \
        }                                                                                  \
    } while (0)

struct EnvBuffers {
    MrsimState st{};
    float *obs = nullptr, *rew = nullptr;
    uint8_t* done = nullptr;
};

static int alloc_env(int64_t n, EnvBuffers& b) {
    HIP_OK(hipMalloc(&b.st.pos, n * 2 * sizeof(double)));
    HIP_OK(hipMalloc(&b.st.aux, n * 4 * sizeof(float)));
    HIP_OK(hipMalloc(&b.st.ep_ret, n * sizeof(float)));
    HIP_OK(hipMalloc(&b.obs, n * 5 * sizeof(float)));
    HIP_OK(hipMalloc(&b.rew, n * sizeof(float)));
    HIP_OK(hipMalloc(&b.done, n));
    return 0;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 4096;
    const int T = 51;
    const uint64_t seed = 7;
    if (mrsim_abi_version() != MRSIM_ABI_VERSION) { std::fprintf(stderr, "ABI version mismatch\n"); return 1; }
    if (mrsim_device_count() < 1) { std::fprintf(stderr, "no HIP device (libmrsim has no CPU fallback)\n"); return 1; }
    char name[128];
    SIM_OK(mrsim_device_name(0, name, sizeof(name)));

    MrsimParams p;
    SIM_OK(mrsim_default_params(&p));  // MR_Env.reset defaults: noise_var = 1, a0 = 1, nominal law
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    // (1) the gym loop: reset, then 51 x step with the exploration policy drawn in-kernel
    EnvBuffers a;
    if (int rc = alloc_env(n, a)) return rc;
    SIM_OK(mrsim_reset(&p, n, 0, &a.st, nullptr, nullptr, nullptr, a.obs, 0, seed, 0, stream));
    MrsimStepIO io;
    std::memset(&io, 0, sizeof(io));
    io.obs = a.obs; io.rew = a.rew; io.done = a.done;
    std::vector<uint8_t> done(n);
    int64_t done_before_end = 0;
    for (int t = 1; t <= T; ++t) {
        SIM_OK(mrsim_step(&p, n, 0, &a.st, &io, seed, (uint64_t)t, stream));
        if (t == T - 1) {
            HIP_OK(hipMemcpyAsync(done.data(), a.done, n, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
            for (int64_t i = 0; i < n; ++i) done_before_end += done[i];
        }
    }
    std::vector<double> pos_a(2 * n);
    std::vector<float> ret_a(n);
    HIP_OK(hipMemcpyAsync(done.data(), a.done, n, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(pos_a.data(), a.st.pos, 2 * n * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(ret_a.data(), a.st.ep_ret, n * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    // (2) the same episode as one fused launch
    EnvBuffers b;
    if (int rc = alloc_env(n, b)) return rc;
    SIM_OK(mrsim_reset(&p, n, 0, &b.st, nullptr, nullptr, nullptr, b.obs, 0, seed, 0, stream));
    uint8_t* done_T = nullptr;
    HIP_OK(hipMalloc(&done_T, (size_t)T * n));
    MrsimRolloutIO ro;
    std::memset(&ro, 0, sizeof(ro));
    ro.T = T; ro.done_T = done_T;
    SIM_OK(mrsim_rollout(&p, n, 0, &b.st, &ro, seed, 1, stream));
    std::vector<double> pos_b(2 * n);
    std::vector<float> ret_b(n);
    std::vector<uint8_t> done_b(n);
    HIP_OK(hipMemcpyAsync(pos_b.data(), b.st.pos, 2 * n * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(ret_b.data(), b.st.ep_ret, n * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(done_b.data(), done_T + (size_t)(T - 1) * n, n, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    int64_t n_done = 0, n_ret = 0, n_same = 0;
    for (int64_t i = 0; i < n; ++i) {
        n_done += done[i] == 1 && done_b[i] == 1;
        n_ret += ret_a[i] == 510.0f && ret_b[i] == 510.0f;
        n_same += std::memcmp(&pos_a[2 * i], &pos_b[2 * i], 2 * sizeof(double)) == 0;
    }
    std::printf("%s: n=%lld  done@51=%lld  done@50=%lld  return==510: %lld  rollout==steps (bitwise): %lld\n", name,
                (long long)n, (long long)n_done, (long long)done_before_end, (long long)n_ret, (long long)n_same);
    // error behaviour of the boundary: bad arguments come back as codes, never as exceptions
    const bool errs = mrsim_step(nullptr, n, 0, &a.st, &io, seed, 0, stream) == MRSIM_EINVAL &&
                      mrsim_step(&p, -1, 0, &a.st, &io, seed, 0, stream) == MRSIM_EINVAL &&
                      mrsim_random_policy(&p, n, 0, a.rew + 1, seed, 0, stream) == MRSIM_EALIGN;
    const bool ok = n_done == n && done_before_end == 0 && n_ret == n && n_same == n && errs;
    std::printf(ok ? "ABI_DEMO_OK\n" : "ABI_DEMO_FAIL\n");
    return ok ? 0 : 1;
}
