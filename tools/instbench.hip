// instbench.hip -- measurement aid: issue cost (cycles per wave-instruction per SIMD) of the VALU ops the
// step kernel is made of, at W waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o tools/instbench tools/instbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 2000;
#define REP8(x) x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    unsigned long long d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    float f0 = a0 * 1e-9f, f1 = a1 * 1e-9f, f2 = a2 * 1e-9f, f3 = a3 * 1e-9f, f4 = a4 * 1e-9f, f5 = a5 * 1e-9f, f6 = a6 * 1e-9f, f7 = a7 * 1e-9f;
    double g0 = f0, g1 = f1, g2 = f2, g3 = f3, g4 = f4, g5 = f5, g6 = f6, g7 = f7;
    const unsigned M = 0xD2511F53u;
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) { // v_mad_u64_u32
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d0) : "v"((unsigned)d0), "s"(M) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d1) : "v"((unsigned)d1), "s"(M) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d2) : "v"((unsigned)d2), "s"(M) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d3) : "v"((unsigned)d3), "s"(M) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d4) : "v"((unsigned)d4), "s"(M) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d5) : "v"((unsigned)d5), "s"(M) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d6) : "v"((unsigned)d6), "s"(M) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d7) : "v"((unsigned)d7), "s"(M) : "vcc");
        } else if (OP == 1) { // v_mul_hi_u32
            asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a0) : "v"(a0), "s"(M)); asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a1) : "v"(a1), "s"(M));
            asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a2) : "v"(a2), "s"(M)); asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a3) : "v"(a3), "s"(M));
            asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a4) : "v"(a4), "s"(M)); asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a5) : "v"(a5), "s"(M));
            asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a6) : "v"(a6), "s"(M)); asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a7) : "v"(a7), "s"(M));
        } else if (OP == 2) { // v_mul_lo_u32
            asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a0) : "v"(a0), "s"(M)); asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a1) : "v"(a1), "s"(M));
            asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a2) : "v"(a2), "s"(M)); asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a3) : "v"(a3), "s"(M));
            asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a4) : "v"(a4), "s"(M)); asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a5) : "v"(a5), "s"(M));
            asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a6) : "v"(a6), "s"(M)); asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a7) : "v"(a7), "s"(M));
        } else if (OP == 3) { // v_xor_b32
            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a0) : "v"(a0), "v"(a1)); asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a1) : "v"(a1), "v"(a2));
            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a2) : "v"(a2), "v"(a3)); asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a3) : "v"(a3), "v"(a4));
            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a4) : "v"(a4), "v"(a5)); asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a5) : "v"(a5), "v"(a6));
            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a6) : "v"(a6), "v"(a7)); asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a7) : "v"(a7), "v"(a0));
        } else if (OP == 4) { // v_fma_f64
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(g0) : "v"(g1), "v"(g2)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(g1) : "v"(g2), "v"(g3));
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(g2) : "v"(g3), "v"(g4)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(g3) : "v"(g4), "v"(g5));
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(g4) : "v"(g5), "v"(g6)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(g5) : "v"(g6), "v"(g7));
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(g6) : "v"(g7), "v"(g0)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(g7) : "v"(g0), "v"(g1));
        } else if (OP == 5) { // v_fma_f32
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f0) : "v"(f1), "v"(f2)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f1) : "v"(f2), "v"(f3));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f2) : "v"(f3), "v"(f4)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f3) : "v"(f4), "v"(f5));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f4) : "v"(f5), "v"(f6)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f5) : "v"(f6), "v"(f7));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f6) : "v"(f7), "v"(f0)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f7) : "v"(f0), "v"(f1));
        } else if (OP == 6) { // v_sin_f32
            asm volatile("v_sin_f32 %0, %0" : "+v"(f0)); asm volatile("v_sin_f32 %0, %0" : "+v"(f1)); asm volatile("v_sin_f32 %0, %0" : "+v"(f2)); asm volatile("v_sin_f32 %0, %0" : "+v"(f3));
            asm volatile("v_sin_f32 %0, %0" : "+v"(f4)); asm volatile("v_sin_f32 %0, %0" : "+v"(f5)); asm volatile("v_sin_f32 %0, %0" : "+v"(f6)); asm volatile("v_sin_f32 %0, %0" : "+v"(f7));
        } else if (OP == 7) { // v_mul_u32_u24
            asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a0) : "v"(a0), "v"(a1)); asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a1) : "v"(a1), "v"(a2));
            asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a2) : "v"(a2), "v"(a3)); asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a3) : "v"(a3), "v"(a4));
            asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a4) : "v"(a4), "v"(a5)); asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a5) : "v"(a5), "v"(a6));
            asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a6) : "v"(a6), "v"(a7)); asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a7) : "v"(a7), "v"(a0));
        } else if (OP == 8) { // v_mul_f64
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(g0) : "v"(g0), "v"(g1)); asm volatile("v_mul_f64 %0, %1, %2" : "=v"(g1) : "v"(g1), "v"(g2));
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(g2) : "v"(g2), "v"(g3)); asm volatile("v_mul_f64 %0, %1, %2" : "=v"(g3) : "v"(g3), "v"(g4));
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(g4) : "v"(g4), "v"(g5)); asm volatile("v_mul_f64 %0, %1, %2" : "=v"(g5) : "v"(g5), "v"(g6));
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(g6) : "v"(g6), "v"(g7)); asm volatile("v_mul_f64 %0, %1, %2" : "=v"(g7) : "v"(g7), "v"(g0));
        } else if (OP == 9) { // v_cvt_f64_f32
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g0) : "v"(f0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g1) : "v"(f1));
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g2) : "v"(f2)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g3) : "v"(f3));
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g4) : "v"(f4)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g5) : "v"(f5));
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g6) : "v"(f6)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g7) : "v"(f7));
        } else if (OP == 10) { // v_add_f64
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(g0) : "v"(g0), "v"(g1)); asm volatile("v_add_f64 %0, %1, %2" : "=v"(g1) : "v"(g1), "v"(g2));
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(g2) : "v"(g2), "v"(g3)); asm volatile("v_add_f64 %0, %1, %2" : "=v"(g3) : "v"(g3), "v"(g4));
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(g4) : "v"(g4), "v"(g5)); asm volatile("v_add_f64 %0, %1, %2" : "=v"(g5) : "v"(g5), "v"(g6));
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(g6) : "v"(g6), "v"(g7)); asm volatile("v_add_f64 %0, %1, %2" : "=v"(g7) : "v"(g7), "v"(g0));
        }
    }
    unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7);
    r ^= __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) ^ (unsigned)__double_as_longlong(g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7);
    if (r == 0x12345678u) out[threadIdx.x] = r;
}

template <int OP>
void run(const char* name, unsigned* out, int waves_per_simd) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd blocks of 4 waves) -> waves_per_simd per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 2u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double inst_per_simd = (double)waves_per_simd * ITERS * 8;
    printf("%-16s waves/SIMD=%d  %.2f ns per wave-instr per SIMD (= %.1f cycles @2.4GHz)\n", name, waves_per_simd,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}

int main() {
    unsigned* out; CK(hipMalloc(&out, 4096));
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_mad_u64_u32", out, w); run<1>("v_mul_hi_u32", out, w); run<2>("v_mul_lo_u32", out, w); run<7>("v_mul_u32_u24", out, w);
        run<3>("v_xor_b32", out, w); run<5>("v_fma_f32", out, w); run<4>("v_fma_f64", out, w); run<8>("v_mul_f64", out, w);
        run<10>("v_add_f64", out, w); run<9>("v_cvt_f64_f32", out, w); run<6>("v_sin_f32", out, w);
    }
    return 0;
}
