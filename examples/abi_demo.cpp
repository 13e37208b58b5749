// A consumer of libmrsim's C ABI without Python or torch: device buffers from the HIP runtime, one
// MR_Env episode of N envs through mrsim_reset / mrsim_step, then the same episode through ONE
// mrsim_rollout launch, and a check that both give the same bits and the reference's episode shape
// (51 steps, reward 10 per step, done by timeout: MR_env.py:62,89,136-152).  Then the DDPG collection loop of
// RL/MR_ddpg.py:270-311 with the actor on the device (ABI 3): fold a batch norm, pack the network, upload the block, and
// compare one episode as 51 x (mrsim_actor_forward -> mrsim_step) with the same episode as one mrsim_rollout whose policy
// source is the in-kernel actor -- actions and positions must agree bitwise.  Then the step kernel as the writer of a replay ring
// (MrsimStepIO.replay, ABI 5) and one env driven from the host on a pinned record whose step word the host polls (done_word).
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
    // (3) RL/MR_ddpg.py:277-278 with the actor as a policy source: actor.predict(state) + actor_noise(), env.step(action)
    int64_t act_same = 0, act_moved = 0;
    {
        const int H = MRSIM_ACTOR_HIDDEN;
        std::vector<float> w1(H * 5), b1(H), g(H), be(H), mu(H), var(H), w1f(H * 5), b1f(H), w2(H * H), b2(H), w3(2 * H), b3(2);
        uint32_t lcg = 12345u;
        auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)(lcg >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f; };
        for (auto& v : w1) v = 0.4f * rnd();
        for (auto& v : w2) v = 0.12f * rnd();
        for (auto& v : w3) v = 0.3f * rnd();
        for (int i = 0; i < H; ++i) { b1[i] = 0.1f * rnd(); b2[i] = 0.1f * rnd(); g[i] = 1.0f + 0.3f * rnd(); be[i] = 0.1f * rnd();
                                      mu[i] = 0.2f * rnd(); var[i] = 1.0f + 0.5f * rnd(); }
        b3[0] = b3[1] = 0.0f;
        // tflearn batch_normalization at inference, folded into the first fully_connected layer
        SIM_OK(mrsim_actor_fold_bn_host(H, 5, w1.data(), b1.data(), g.data(), be.data(), mu.data(), var.data(), 1e-5f,
                                        w1f.data(), b1f.data()));
        MrsimActorWeights W;
        std::memset(&W, 0, sizeof(W));
        W.w1 = w1f.data(); W.b1 = b1f.data(); W.w2 = w2.data(); W.b2 = b2.data(); W.w3 = w3.data(); W.b3 = b3.data();
        for (int k = 0; k < 5; ++k) W.obs_scale[k] = 0.01f;
        W.action_bound[0] = 20.0f; W.action_bound[1] = 6.2831853f;          // env.action_space.high, RL/MR_ddpg.py:345
        std::vector<float> blob(MRSIM_ACTOR_BLOB_FLOATS);
        SIM_OK(mrsim_actor_pack_host(&W, blob.data()));
        float *blob_d = nullptr, *ou_a = nullptr, *ou_b = nullptr, *act_a = nullptr, *act_T = nullptr;
        HIP_OK(hipMalloc(&blob_d, blob.size() * sizeof(float)));
        HIP_OK(hipMemcpy(blob_d, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&ou_a, n * 2 * sizeof(float)));
        HIP_OK(hipMalloc(&ou_b, n * 2 * sizeof(float)));
        HIP_OK(hipMemset(ou_a, 0, n * 2 * sizeof(float)));
        HIP_OK(hipMemset(ou_b, 0, n * 2 * sizeof(float)));
        HIP_OK(hipMalloc(&act_a, n * 2 * sizeof(float)));
        HIP_OK(hipMalloc(&act_T, (size_t)T * n * 2 * sizeof(float)));
        MrsimActor actor_a = {blob_d, ou_a, 0.15f, 0.3f, 1e-2f, 0, MRSIM_ACTOR_F32, 0};   // OUNoise defaults, :60
        MrsimActor actor_b = actor_a;
        actor_b.ou_state = ou_b;
        // gym-loop form on env a
        SIM_OK(mrsim_reset(&p, n, 0, &a.st, nullptr, nullptr, nullptr, a.obs, 0, seed, 100, stream));
        MrsimStepIO ia;
        std::memset(&ia, 0, sizeof(ia));
        ia.obs = a.obs; ia.rew = a.rew; ia.done = a.done; ia.actions = act_a;
        std::vector<float> last_a(2 * n), last_b(2 * n);
        for (int t = 1; t <= T; ++t) {
            SIM_OK(mrsim_actor_forward(&p, n, 0, &actor_a, nullptr, a.obs, act_a, seed, 100 + (uint64_t)t, stream));
            SIM_OK(mrsim_step(&p, n, 0, &a.st, &ia, seed, 100 + (uint64_t)t, stream));
        }
        HIP_OK(hipMemcpyAsync(last_a.data(), act_a, 2 * n * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(pos_a.data(), a.st.pos, 2 * n * sizeof(double), hipMemcpyDeviceToHost, stream));
        // fused form on env b: the whole episode in one launch, the actor evaluated on the observation held in registers
        SIM_OK(mrsim_reset(&p, n, 0, &b.st, nullptr, nullptr, nullptr, b.obs, 0, seed, 100, stream));
        MrsimRolloutIO rb;
        std::memset(&rb, 0, sizeof(rb));
        rb.T = T; rb.actions_out_T = act_T; rb.actor = &actor_b;
        SIM_OK(mrsim_rollout(&p, n, 0, &b.st, &rb, seed, 101, stream));
        HIP_OK(hipMemcpyAsync(last_b.data(), act_T + (size_t)(T - 1) * n * 2, 2 * n * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(pos_b.data(), b.st.pos, 2 * n * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (int64_t i = 0; i < n; ++i) {
            act_same += std::memcmp(&last_a[2 * i], &last_b[2 * i], 2 * sizeof(float)) == 0 &&
                        std::memcmp(&pos_a[2 * i], &pos_b[2 * i], 2 * sizeof(double)) == 0;
            act_moved += last_a[2 * i] != 0.0f && last_a[2 * i] > -21.5f && last_a[2 * i] < 21.5f;
        }
        // one policy source per launch: actions AND actor is an argument error
        rb.actions = act_T;
        if (mrsim_rollout(&p, n, 0, &b.st, &rb, seed, 200, stream) != MRSIM_EINVAL) act_same = -1;
        std::printf("actor in the loop: rollout(actor)==51 x (actor_forward -> step) (bitwise): %lld  actions in range: %lld\n",
                    (long long)act_same, (long long)act_moved);
        // (4) RL/MR_ddpg.py:277-282 in ONE launch per env step: policy, noise, env.step AND replay_buffer.add -- the step kernel writes
        // its transitions into a device ring itself (MrsimStepIO.replay).  Three steps of n envs into a ring of 4 n rows: row i of
        // the first step must hold the reset observation of env i as `state`, the last n rows the last actions.
        const int64_t cap = 4 * n;
        float *rs = nullptr, *ra = nullptr, *rr = nullptr, *rd = nullptr, *rs2 = nullptr, *ended = nullptr;
        HIP_OK(hipMalloc(&rs, cap * 5 * sizeof(float))); HIP_OK(hipMalloc(&rs2, cap * 5 * sizeof(float)));
        HIP_OK(hipMalloc(&ra, cap * 2 * sizeof(float))); HIP_OK(hipMalloc(&rr, cap * sizeof(float)));
        HIP_OK(hipMalloc(&rd, cap * sizeof(float))); HIP_OK(hipMalloc(&ended, 2 * sizeof(float)));
        HIP_OK(hipMemset(ended, 0, 2 * sizeof(float)));
        HIP_OK(hipMemsetAsync(ou_a, 0, n * 2 * sizeof(float), stream));
        SIM_OK(mrsim_reset(&p, n, 0, &a.st, nullptr, nullptr, nullptr, a.obs, 0, seed, 300, stream));
        std::vector<float> obs0(5 * n), ring_s(5 * n), ring_a(2 * n), ring_r(n);
        HIP_OK(hipMemcpyAsync(obs0.data(), a.obs, 5 * n * sizeof(float), hipMemcpyDeviceToHost, stream));
        MrsimReplaySink sink;
        std::memset(&sink, 0, sizeof(sink));
        sink.s = rs; sink.a = ra; sink.r = rr; sink.done = rd; sink.s2 = rs2; sink.ended2 = ended; sink.capacity = (int32_t)cap;
        for (int k = 0; k < 5; ++k) sink.obs_scale[k] = 1.0f;
        MrsimStepIO ic;
        std::memset(&ic, 0, sizeof(ic));
        ic.obs = a.obs; ic.rew = a.rew; ic.done = a.done; ic.actions_out = act_a; ic.actor = &actor_a; ic.replay = &sink;
        for (int t = 1; t <= 3; ++t) {
            sink.head = (int32_t)((t - 1) * n);
            SIM_OK(mrsim_step(&p, n, 0, &a.st, &ic, seed, 300 + (uint64_t)t, stream));
        }
        HIP_OK(hipMemcpyAsync(ring_s.data(), rs, 5 * n * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(ring_a.data(), ra + 2 * 2 * n, 2 * n * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(ring_r.data(), rr + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(last_a.data(), act_a, 2 * n * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        int64_t ring_ok = 0;
        for (int64_t i = 0; i < n; ++i)
            ring_ok += std::memcmp(&ring_s[5 * i], &obs0[5 * i], 5 * sizeof(float)) == 0 &&
                       std::memcmp(&ring_a[2 * i], &last_a[2 * i], 2 * sizeof(float)) == 0 && ring_r[i] == 10.0f;
        ic.actor = nullptr;     // the sink stores the observation the kernel's OWN policy saw: without an actor it is an argument error
        if (mrsim_step(&p, n, 0, &a.st, &ic, seed, 400, stream) != MRSIM_EINVAL) ring_ok = -1;
        std::printf("replay ring written by the step kernel: rows as expected: %lld\n", (long long)ring_ok);
        if (ring_ok != n) act_same = -1;
    }
    // (5) one env driven from the host (the drop-in MR_Env.step): state, action and outputs in ONE pinned block, the kernel stores a
    // step word after its outputs and the host polls it -- no copy call, no stream wait
    int host_ok = 0;
    {
        void *hp = nullptr, *dp = nullptr;
        SIM_OK(mrsim_host_alloc(192, &hp, &dp));
        char* h = static_cast<char*>(hp);
        char* d = static_cast<char*>(dp);
        reinterpret_cast<float*>(h + 16)[2] = 1.0f;                                   // aux: h_abs / dt
        reinterpret_cast<double*>(h + 144)[0] = 110.0; reinterpret_cast<double*>(h + 144)[1] = 115.0;   // init_xy
        MrsimState s1 = {reinterpret_cast<double*>(d), reinterpret_cast<float*>(d + 16), reinterpret_cast<float*>(d + 32)};
        SIM_OK(mrsim_reset(&p, 1, 0, &s1, nullptr, reinterpret_cast<double*>(d + 144), nullptr, reinterpret_cast<float*>(d + 64), 0, seed, 500, stream));
        SIM_OK(mrsim_stream_synchronize(stream));
        MrsimStepIO i1;
        std::memset(&i1, 0, sizeof(i1));
        i1.actions = reinterpret_cast<float*>(d + 48); i1.obs = reinterpret_cast<float*>(d + 64); i1.rew = reinterpret_cast<float*>(d + 96);
        i1.done = reinterpret_cast<uint8_t*>(d + 100); i1.done_word = reinterpret_cast<int32_t*>(d + 160);
        int polled = 0;
        for (int k = 1; k <= 20; ++k) {
            reinterpret_cast<float*>(h + 48)[0] = 4.0f; reinterpret_cast<float*>(h + 48)[1] = 0.785398f;      // the action, written by the host
            i1.done_value = k;
            SIM_OK(mrsim_step(&p, 1, 0, &s1, &i1, seed, 500 + (uint64_t)k, stream));
            polled += mrsim_host_wait_word(reinterpret_cast<const int32_t*>(h + 160), k, 2000000) == MRSIM_OK;
        }
        const float* ob = reinterpret_cast<const float*>(h + 64);
        const int32_t counter = reinterpret_cast<const int32_t*>(h + 16)[3];
        host_ok = polled == 20 && counter == 20 && ob[0] > 110.0f && ob[0] < 116.0f && reinterpret_cast<const float*>(h + 96)[0] == 10.0f;
        std::printf("one env on a host record, 20 steps polled on the record's step word: %s (x = %.4f, counter = %d)\n", host_ok ? "ok" : "FAIL",
                    ob[0], counter);
        SIM_OK(mrsim_stream_synchronize(stream));
        SIM_OK(mrsim_host_free(hp));
    }
    // error behaviour of the boundary: bad arguments come back as codes, never as exceptions
    const bool errs = mrsim_step(nullptr, n, 0, &a.st, &io, seed, 0, stream) == MRSIM_EINVAL &&
                      mrsim_step(&p, -1, 0, &a.st, &io, seed, 0, stream) == MRSIM_EINVAL &&
                      mrsim_random_policy(&p, n, 0, a.rew + 1, seed, 0, stream) == MRSIM_EALIGN;
    const bool ok = n_done == n && done_before_end == 0 && n_ret == n && n_same == n && errs && act_same == n && act_moved == n && host_ok;
    std::printf(ok ? "ABI_DEMO_OK\n" : "ABI_DEMO_FAIL\n");
    return ok ? 0 : 1;
}
