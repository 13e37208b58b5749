// mfmabench.hip -- measurement aid: how the matrix pipe and the vector unit of one gfx950 SIMD share time, for the actor-in-the-
// loop kernels (mr_rl_amd/csrc/mrsim_actor.h).  W waves per SIMD each run R rounds of [P MFMAs on two alternating accumulators]
// followed by [Q independent v_fma_f32] -- the shape of one env step with the actor (MFMA phase, then vector phase) -- and the
// program prints shader cycles per round per SIMD (s_memtime around the loop, the longest-living wave) for
//   f32  = v_mfma_f32_32x32x2_f32  (64 cycles of matrix pipe each, the f32 vector rate)
//   bf16 = v_mfma_f32_32x32x16_bf16 (32 cycles each)
// against the two models  max(W P c, W Q v)  (phases of different waves overlap)  and  W (P c + Q v)  (they do not).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfmabench tools/mfmabench.hip ; tools/mfmabench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// STAG: 0 = every wave runs [MFMA phase, vector phase]; 1 = the waves of odd blocks start with the vector phase (anti-phase);
// 2 = same phase, but odd blocks run at s_setprio 3 (do distinct priorities de-synchronise the waves by themselves?)
template <int KIND, int P, int Q, int STAG = 0>
__global__ __launch_bounds__(256) void k(float* out, int rounds, unsigned long long* cyc) {
    f32x16 acc0 = {0}, acc1 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 a8, b8;
    for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(a + j); b8[j] = (__bf16)(b - j); }
    float f0 = a, f1 = b, f2 = a + 1, f3 = b + 1, f4 = a + 2, f5 = b + 2, f6 = a + 3, f7 = b + 3;
    if (STAG == 2 && (blockIdx.x & 1)) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = clock64();
    if (STAG == 1 && (blockIdx.x & 1)) {
#pragma unroll
        for (int i = 0; i < Q; i += 8) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(f2), "v"(f3));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2) : "v"(f3), "v"(f4)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(f4), "v"(f5));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f4) : "v"(f5), "v"(f6)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f5) : "v"(f6), "v"(f7));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f6) : "v"(f7), "v"(f0)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f7) : "v"(f0), "v"(f1));
        }
    }
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < P; i += 2) {
            if (KIND == 0) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8, a8, acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < Q; i += 8) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(f2), "v"(f3));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2) : "v"(f3), "v"(f4)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(f4), "v"(f5));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f4) : "v"(f5), "v"(f6)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f5) : "v"(f6), "v"(f7));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f6) : "v"(f7), "v"(f0)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f7) : "v"(f0), "v"(f1));
        }
    }
    const unsigned long long t1 = clock64();
    float s = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int P, int Q, int STAG = 0>
static void run(const char* name, int W, int cus) {
    const int rounds = 60, blocks = cus * W;      // one wave per SIMD per block, W blocks per CU
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
    CK(hipMalloc(&cyc, (size_t)blocks * 4 * sizeof(unsigned long long)));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<KIND, P, Q, STAG>), dim3(blocks), dim3(256), 0, 0, out, rounds, cyc);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 4);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double c = (KIND == 0) ? 64.0 : 32.0, v = 2.45;      // matrix cycles per MFMA; measured v_fma_f32 issue cost (instbench)
    const double per_round = (double)h[h.size() / 2] / rounds; // median wave lifetime per round = SIMD cycles per round of W waves
    printf("%-5s W=%d stag=%d P=%3d Q=%4d : %8.0f cycles per round per SIMD   overlap model %8.0f   serial model %8.0f   matrix pipe busy %.2f\n",
           name, W, STAG, P, Q, per_round, std::max(W * P * c, W * Q * v), W * (P * c + Q * v), W * P * c / per_round);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.name, cus);
    run<0, 140, 0>("f32", 1, cus); run<0, 140, 0>("f32", 2, cus); run<0, 140, 0>("f32", 4, cus);
    run<0, 140, 600>("f32", 1, cus); run<0, 140, 600>("f32", 2, cus); run<0, 140, 600>("f32", 4, cus);
    run<0, 140, 1200>("f32", 4, cus);
    run<0, 0, 600>("valu", 4, cus);
    run<1, 108, 0>("bf16", 1, cus); run<1, 108, 0>("bf16", 4, cus);
    run<1, 108, 1000>("bf16", 1, cus); run<1, 108, 1000>("bf16", 2, cus); run<1, 108, 1000>("bf16", 4, cus);
    printf("-- two waves per SIMD in anti-phase (stag=1) / with distinct priorities (stag=2)\n");
    run<0, 140, 600, 1>("f32", 2, cus); run<0, 140, 600, 2>("f32", 2, cus);
    run<0, 140, 2000, 0>("f32", 2, cus); run<0, 140, 2000, 1>("f32", 2, cus); run<0, 140, 2000, 2>("f32", 2, cus);
    run<1, 108, 1000, 1>("bf16", 2, cus); run<1, 108, 1000, 2>("bf16", 2, cus);
    return 0;
}
