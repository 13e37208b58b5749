// membench.hip -- measurement aid (not product): the step kernel's HBM access pattern with the
// arithmetic stripped, to find the memory/latency floor of one launch at N envs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench tools/membench.hip ; rocprofv3 --kernel-trace --stats -- tools/membench N
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int BLOCK, int VARIANT>
__global__ __launch_bounds__(BLOCK) void pattern(long long n, double2* __restrict__ pos, float4* __restrict__ aux,
                                                 float* __restrict__ ep, const float2* __restrict__ act,
                                                 float* __restrict__ obs, float* __restrict__ rew, unsigned char* __restrict__ done) {
    __shared__ __attribute__((aligned(16))) float s_obs[BLOCK * 5];
    const long long base = (long long)blockIdx.x * BLOCK;
    const long long i = base + threadIdx.x;
    if (i < n) {
        double2 p = pos[i]; float4 a = aux[i]; float e = ep[i]; float2 c = act[i];
        p.x += (double)c.x * 1e-3; p.y += (double)c.y * 1e-3;
        a.x = c.x; a.y = c.y; a.w = __int_as_float(__float_as_int(a.w) + 1); e += 10.f;
        if (VARIANT == 2) {  // everything nontemporal
            __builtin_nontemporal_store(p.x, &pos[i].x); __builtin_nontemporal_store(p.y, &pos[i].y);
            __builtin_nontemporal_store(a.x, &aux[i].x); __builtin_nontemporal_store(a.y, &aux[i].y);
            __builtin_nontemporal_store(a.z, &aux[i].z); __builtin_nontemporal_store(a.w, &aux[i].w);
            __builtin_nontemporal_store(e, &ep[i]); __builtin_nontemporal_store(10.f, &rew[i]);
            __builtin_nontemporal_store((unsigned char)(e > 500.f), &done[i]);
        } else if (VARIANT == 3) {  // agent-scope relaxed atomic stores = write-through (sc1)
            __hip_atomic_store(&pos[i].x, p.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&pos[i].y, p.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<double*>(&aux[i].x), *reinterpret_cast<double*>(&a.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<double*>(&aux[i].z), *reinterpret_cast<double*>(&a.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ep[i], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&rew[i], 10.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            done[i] = (unsigned char)(e > 500.f);
        } else if (VARIANT == 1) {  // nontemporal stores
            __builtin_nontemporal_store(p.x, &pos[i].x); __builtin_nontemporal_store(p.y, &pos[i].y);
            __builtin_nontemporal_store(a.x, &aux[i].x); __builtin_nontemporal_store(a.y, &aux[i].y);
            __builtin_nontemporal_store(a.z, &aux[i].z); __builtin_nontemporal_store(a.w, &aux[i].w);
        } else { pos[i] = p; aux[i] = a; }
        if (VARIANT < 2) { ep[i] = e; rew[i] = 10.f; done[i] = (unsigned char)(e > 500.f); }
        s_obs[threadIdx.x * 5 + 0] = (float)p.x; s_obs[threadIdx.x * 5 + 1] = (float)p.y; s_obs[threadIdx.x * 5 + 2] = 0.f;
        s_obs[threadIdx.x * 5 + 3] = 0.f; s_obs[threadIdx.x * 5 + 4] = (float)(p.x + p.y);
    }
    __syncthreads();
    const long long rows = (n - base) < BLOCK ? (n - base) : BLOCK;
    const int nvalid = (int)rows * 5;
    float* dst = obs + base * 5;
    for (int q = threadIdx.x * 4; q < nvalid; q += BLOCK * 4)
        if (q + 4 <= nvalid) {
            const float4 v = *reinterpret_cast<const float4*>(s_obs + q);
            if (VARIANT == 2) { __builtin_nontemporal_store(v.x, dst + q); __builtin_nontemporal_store(v.y, dst + q + 1);
                                __builtin_nontemporal_store(v.z, dst + q + 2); __builtin_nontemporal_store(v.w, dst + q + 3); }
            else if (VARIANT == 3) { __hip_atomic_store(reinterpret_cast<double*>(dst + q), *reinterpret_cast<const double*>(&v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                     __hip_atomic_store(reinterpret_cast<double*>(dst + q + 2), *reinterpret_cast<const double*>(&v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else *reinterpret_cast<float4*>(dst + q) = v;
        }
}

// read-only / write-only halves
__global__ __launch_bounds__(256) void rd_only(long long n, const double2* pos, const float4* aux, const float* ep, const float2* act, float* sink) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { double2 p = pos[i]; float4 a = aux[i]; float e = ep[i]; float2 c = act[i];
        float v = (float)p.x + (float)p.y + a.x + a.y + a.z + a.w + e + c.x + c.y; if (v == 1.2345f) sink[i] = v; }
}
__global__ __launch_bounds__(256) void wr_only(long long n, double2* pos, float4* aux, float* ep, float* obs, float* rew, unsigned char* done) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { pos[i] = make_double2(1.0, 2.0); aux[i] = make_float4(1, 2, 3, 4); ep[i] = 1.f; rew[i] = 10.f; done[i] = 0;
        for (int j = 0; j < 5; ++j) obs[j * n + i] = 1.f; }
}
__global__ void empty_k(long long n, float* x) { if (n < 0) x[0] = 1.f; }

int main(int argc, char** argv) {
    long long n = argc > 1 ? atoll(argv[1]) : 262144;
    int reps = argc > 2 ? atoi(argv[2]) : 200;
    double2* pos; float4* aux; float* ep; float2* act; float* obs; float* rew; unsigned char* done;
    CK(hipMalloc(&pos, n * 16)); CK(hipMalloc(&aux, n * 16)); CK(hipMalloc(&ep, n * 4)); CK(hipMalloc(&act, n * 8));
    CK(hipMalloc(&obs, n * 20)); CK(hipMalloc(&rew, n * 4)); CK(hipMalloc(&done, n));
    CK(hipMemset(pos, 0, n * 16)); CK(hipMemset(aux, 0, n * 16)); CK(hipMemset(ep, 0, n * 4)); CK(hipMemset(act, 0, n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int r = 0; r < 20; ++r) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s n=%lld  %.3f us/launch (back-to-back incl. gaps)\n", name, n, ms * 1e3 / reps);
    };
    unsigned g256 = (unsigned)((n + 255) / 256), g512 = (unsigned)((n + 511) / 512), g128 = (unsigned)((n + 127) / 128), g1024 = (unsigned)((n + 1023) / 1024);
    run("empty", [&] { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, 0, n, rew); });
    run("pattern<256>", [&] { hipLaunchKernelGGL((pattern<256, 0>), dim3(g256), dim3(256), 0, 0, n, pos, aux, ep, act, obs, rew, done); });
    run("pattern<256,nt>", [&] { hipLaunchKernelGGL((pattern<256, 1>), dim3(g256), dim3(256), 0, 0, n, pos, aux, ep, act, obs, rew, done); });
    run("pattern<256,nt-all>", [&] { hipLaunchKernelGGL((pattern<256, 2>), dim3(g256), dim3(256), 0, 0, n, pos, aux, ep, act, obs, rew, done); });
    run("pattern<256,sc1>", [&] { hipLaunchKernelGGL((pattern<256, 3>), dim3(g256), dim3(256), 0, 0, n, pos, aux, ep, act, obs, rew, done); });
    run("pattern<128>", [&] { hipLaunchKernelGGL((pattern<128, 0>), dim3(g128), dim3(128), 0, 0, n, pos, aux, ep, act, obs, rew, done); });
    run("pattern<512>", [&] { hipLaunchKernelGGL((pattern<512, 0>), dim3(g512), dim3(512), 0, 0, n, pos, aux, ep, act, obs, rew, done); });
    run("pattern<1024>", [&] { hipLaunchKernelGGL((pattern<1024, 0>), dim3(g1024), dim3(1024), 0, 0, n, pos, aux, ep, act, obs, rew, done); });
    run("rd_only", [&] { hipLaunchKernelGGL(rd_only, dim3(g256), dim3(256), 0, 0, n, pos, aux, ep, act, rew); });
    run("wr_only", [&] { hipLaunchKernelGGL(wr_only, dim3(g256), dim3(256), 0, 0, n, pos, aux, ep, obs, rew, done); });
    return 0;
}
