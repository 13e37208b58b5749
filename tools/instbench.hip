// instbench.hip -- measurement aid: issue cost (ns and cycles per wave-instruction per SIMD) of the VALU instructions
// the rollout / step kernels are made of, at W = 1, 2, 4, 8 waves per SIMD, eight independent chains per wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/instbench tools/instbench.hip ; tools/instbench [--json]
// Two figures per instruction: ns (HIP events around the launch) and shader cycles (s_memtime around the loop, lifetime of the
// last-finishing waves).  The CYCLE figures price the kernel's instruction mix in tools/collect_profiles.py -> profiles/rNN/pmc_valu.json
// (the "VALU issue floor" of bench.py's roofline, compared with the kernel's own SQ_WAVE_CYCLES -- a ratio the clock the
// chip happens to hold (DVFS) cancels out of); cycles / ns = the clock this short burst ran at.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 2000;

// one instruction per chain, 8 chains; operands: a* u32, d* u64, f* f32, g* f64, p* f32x2
#define A8(T, C) asm volatile(T : C(a0) : "v"(a0), "v"(a1)); asm volatile(T : C(a1) : "v"(a1), "v"(a2)); asm volatile(T : C(a2) : "v"(a2), "v"(a3)); asm volatile(T : C(a3) : "v"(a3), "v"(a4)); \
                 asm volatile(T : C(a4) : "v"(a4), "v"(a5)); asm volatile(T : C(a5) : "v"(a5), "v"(a6)); asm volatile(T : C(a6) : "v"(a6), "v"(a7)); asm volatile(T : C(a7) : "v"(a7), "v"(a0));
#define F8(T) asm volatile(T : "=v"(f0) : "v"(f0), "v"(f1)); asm volatile(T : "=v"(f1) : "v"(f1), "v"(f2)); asm volatile(T : "=v"(f2) : "v"(f2), "v"(f3)); asm volatile(T : "=v"(f3) : "v"(f3), "v"(f4)); \
              asm volatile(T : "=v"(f4) : "v"(f4), "v"(f5)); asm volatile(T : "=v"(f5) : "v"(f5), "v"(f6)); asm volatile(T : "=v"(f6) : "v"(f6), "v"(f7)); asm volatile(T : "=v"(f7) : "v"(f7), "v"(f0));
#define F8U(T) asm volatile(T : "+v"(f0)); asm volatile(T : "+v"(f1)); asm volatile(T : "+v"(f2)); asm volatile(T : "+v"(f3)); asm volatile(T : "+v"(f4)); asm volatile(T : "+v"(f5)); asm volatile(T : "+v"(f6)); asm volatile(T : "+v"(f7));
#define G8(T) asm volatile(T : "=v"(g0) : "v"(g0), "v"(g1)); asm volatile(T : "=v"(g1) : "v"(g1), "v"(g2)); asm volatile(T : "=v"(g2) : "v"(g2), "v"(g3)); asm volatile(T : "=v"(g3) : "v"(g3), "v"(g4)); \
              asm volatile(T : "=v"(g4) : "v"(g4), "v"(g5)); asm volatile(T : "=v"(g5) : "v"(g5), "v"(g6)); asm volatile(T : "=v"(g6) : "v"(g6), "v"(g7)); asm volatile(T : "=v"(g7) : "v"(g7), "v"(g0));
#define G8A(T) asm volatile(T : "+v"(g0) : "v"(g1), "v"(g2)); asm volatile(T : "+v"(g1) : "v"(g2), "v"(g3)); asm volatile(T : "+v"(g2) : "v"(g3), "v"(g4)); asm volatile(T : "+v"(g3) : "v"(g4), "v"(g5)); \
               asm volatile(T : "+v"(g4) : "v"(g5), "v"(g6)); asm volatile(T : "+v"(g5) : "v"(g6), "v"(g7)); asm volatile(T : "+v"(g6) : "v"(g7), "v"(g0)); asm volatile(T : "+v"(g7) : "v"(g0), "v"(g1));
#define F8A(T) asm volatile(T : "+v"(f0) : "v"(f1), "v"(f2)); asm volatile(T : "+v"(f1) : "v"(f2), "v"(f3)); asm volatile(T : "+v"(f2) : "v"(f3), "v"(f4)); asm volatile(T : "+v"(f3) : "v"(f4), "v"(f5)); \
               asm volatile(T : "+v"(f4) : "v"(f5), "v"(f6)); asm volatile(T : "+v"(f5) : "v"(f6), "v"(f7)); asm volatile(T : "+v"(f6) : "v"(f7), "v"(f0)); asm volatile(T : "+v"(f7) : "v"(f0), "v"(f1));
#define P8A(T) asm volatile(T : "+v"(p0) : "v"(p1), "v"(p2)); asm volatile(T : "+v"(p1) : "v"(p2), "v"(p3)); asm volatile(T : "+v"(p2) : "v"(p3), "v"(p4)); asm volatile(T : "+v"(p3) : "v"(p4), "v"(p5)); \
               asm volatile(T : "+v"(p4) : "v"(p5), "v"(p6)); asm volatile(T : "+v"(p5) : "v"(p6), "v"(p7)); asm volatile(T : "+v"(p6) : "v"(p7), "v"(p0)); asm volatile(T : "+v"(p7) : "v"(p0), "v"(p1));

typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Op { MAD_U64, MUL_HI, MUL_LO, MUL_U24, XOR, BITOP3, ADD_U32, MOV, CNDMASK, FMA_F32, MUL_F32, ADD_F32, PK_FMA_F32, PK_MUL_F32,
          FMA_F64, MUL_F64, ADD_F64, MAX_F64, CMP_F64, CVT_F64_F32, CVT_F32_F64, CVT_F32_U32, SIN, COS, LOG, SQRT, RCP, N_OPS };
static const char* kNames[N_OPS] = {"v_mad_u64_u32", "v_mul_hi_u32", "v_mul_lo_u32", "v_mul_u32_u24", "v_xor_b32", "v_bitop3_b32", "v_add_u32",
    "v_mov_b32", "v_cndmask_b32", "v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_fma_f64", "v_mul_f64", "v_add_f64",
    "v_max_f64", "v_cmp_lt_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_f32_u32", "v_sin_f32", "v_cos_f32", "v_log_f32", "v_sqrt_f32", "v_rcp_f32"};

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed, unsigned long long* cyc) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    unsigned long long d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    float f0 = a0 * 1e-9f + 0.5f, f1 = a1 * 1e-9f + 0.5f, f2 = a2 * 1e-9f + 0.5f, f3 = a3 * 1e-9f + 0.5f, f4 = a4 * 1e-9f + 0.5f, f5 = a5 * 1e-9f + 0.5f, f6 = a6 * 1e-9f + 0.5f, f7 = a7 * 1e-9f + 0.5f;
    double g0 = f0, g1 = f1, g2 = f2, g3 = f3, g4 = f4, g5 = f5, g6 = f6, g7 = f7;
    f32x2 p0 = {f0, f1}, p1 = {f1, f2}, p2 = {f2, f3}, p3 = {f3, f4}, p4 = {f4, f5}, p5 = {f5, f6}, p6 = {f6, f7}, p7 = {f7, f0};
    const unsigned M = 0xD2511F53u;
    const unsigned long long t0 = clock64();  // s_memtime: shader cycles
    for (int it = 0; it < ITERS; ++it) {
        if (OP == MAD_U64) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d0) : "v"((unsigned)d0), "s"(M) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d1) : "v"((unsigned)d1), "s"(M) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d2) : "v"((unsigned)d2), "s"(M) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d3) : "v"((unsigned)d3), "s"(M) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d4) : "v"((unsigned)d4), "s"(M) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d5) : "v"((unsigned)d5), "s"(M) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d6) : "v"((unsigned)d6), "s"(M) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d7) : "v"((unsigned)d7), "s"(M) : "vcc");
        } else if (OP == MUL_HI) { A8("v_mul_hi_u32 %0, %1, %2", "=v")
        } else if (OP == MUL_LO) { A8("v_mul_lo_u32 %0, %1, %2", "=v")
        } else if (OP == MUL_U24) { A8("v_mul_u32_u24 %0, %1, %2", "=v")
        } else if (OP == XOR) { A8("v_xor_b32 %0, %1, %2", "=v")
        } else if (OP == BITOP3) { A8("v_bitop3_b32 %0, %1, %2, %1 bitop3:0x96", "=v")
        } else if (OP == ADD_U32) { A8("v_add_u32 %0, %1, %2", "=v")
        } else if (OP == MOV) { A8("v_mov_b32 %0, %2", "=v")
        } else if (OP == CNDMASK) { A8("v_cndmask_b32 %0, %1, %2, vcc", "=v")
        } else if (OP == FMA_F32) { F8A("v_fma_f32 %0, %1, %2, %0")
        } else if (OP == MUL_F32) { F8("v_mul_f32 %0, %1, %2")
        } else if (OP == ADD_F32) { F8("v_add_f32 %0, %1, %2")
        } else if (OP == PK_FMA_F32) { P8A("v_pk_fma_f32 %0, %1, %2, %0")
        } else if (OP == PK_MUL_F32) { P8A("v_pk_mul_f32 %0, %1, %2")
        } else if (OP == FMA_F64) { G8A("v_fma_f64 %0, %1, %2, %0")
        } else if (OP == MUL_F64) { G8("v_mul_f64 %0, %1, %2")
        } else if (OP == ADD_F64) { G8("v_add_f64 %0, %1, %2")
        } else if (OP == MAX_F64) { G8("v_max_f64 %0, %1, %2")
        } else if (OP == CMP_F64) {
            asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(g0), "v"(g1) : "vcc"); asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(g1), "v"(g2) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(g2), "v"(g3) : "vcc"); asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(g3), "v"(g4) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(g4), "v"(g5) : "vcc"); asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(g5), "v"(g6) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(g6), "v"(g7) : "vcc"); asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(g7), "v"(g0) : "vcc");
        } else if (OP == CVT_F64_F32) {
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g0) : "v"(f0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g1) : "v"(f1)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g2) : "v"(f2)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g3) : "v"(f3));
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g4) : "v"(f4)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g5) : "v"(f5)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g6) : "v"(f6)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g7) : "v"(f7));
        } else if (OP == CVT_F32_F64) {
            asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f0) : "v"(g0)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f1) : "v"(g1)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f2) : "v"(g2)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f3) : "v"(g3));
            asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f4) : "v"(g4)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f5) : "v"(g5)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f6) : "v"(g6)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f7) : "v"(g7));
        } else if (OP == CVT_F32_U32) {
            asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f0) : "v"(a0)); asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f1) : "v"(a1)); asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f2) : "v"(a2)); asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f3) : "v"(a3));
            asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f4) : "v"(a4)); asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f5) : "v"(a5)); asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f6) : "v"(a6)); asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f7) : "v"(a7));
        } else if (OP == SIN) { F8U("v_sin_f32 %0, %0")
        } else if (OP == COS) { F8U("v_cos_f32 %0, %0")
        } else if (OP == LOG) { F8U("v_log_f32 %0, %0")
        } else if (OP == SQRT) { F8U("v_sqrt_f32 %0, %0")
        } else if (OP == RCP) { F8U("v_rcp_f32 %0, %0")
        }
    }
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63u) == 0) cyc[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
    unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7);
    r ^= __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) ^ (unsigned)__double_as_longlong(g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7);
    r ^= __float_as_uint(p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y);
    if (r == 0x12345678u) out[threadIdx.x] = r;
}

static bool g_json = false;
static bool g_first = true;

static unsigned long long* g_cyc = nullptr;  // one slot per wave (8 waves/SIMD x 1024 SIMDs at most)
static std::vector<unsigned long long> g_host(8192);

template <int OP>
void run(unsigned* out, int waves_per_simd) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd blocks of 4 waves) -> waves_per_simd per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u, g_cyc);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    double best_cyc = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 2u + rep, g_cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            const int nw = blocks * 4;
            CK(hipMemcpy(g_host.data(), g_cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::sort(g_host.begin(), g_host.begin() + nw);
            // the waves of a SIMD are served oldest-first, so the oldest finish early: the SIMD is busy for as long as its
            // LAST wave lives.  99th percentile of the wave lifetimes (shader cycles, s_memtime)
            best_cyc = (double)g_host[(nw * 99) / 100];
        }
    }
    double inst_per_simd = (double)waves_per_simd * ITERS * 8;
    double ns = best * 1e6 / inst_per_simd;
    double cyc = best_cyc / inst_per_simd;  // cycles per wave-instruction per SIMD, independent of the clock the run held
    if (g_json) {
        printf("%s{\"op\": \"%s\", \"waves_per_simd\": %d, \"ns\": %.4f, \"cycles\": %.4f}", g_first ? "" : ",\n", kNames[OP],
               waves_per_simd, ns, cyc);
        g_first = false;
    } else {
        printf("%-16s waves/SIMD=%d  %.2f ns = %.2f shader cycles per wave-instr per SIMD (clock %.2f GHz)\n", kNames[OP], waves_per_simd,
               ns, cyc, cyc / ns);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int OP>
void run_all(unsigned* out, int w) {
    if constexpr (OP < N_OPS) { run<OP>(out, w); run_all<OP + 1>(out, w); }
}

int main(int argc, char** argv) {
    g_json = argc > 1 && !strcmp(argv[1], "--json");
    unsigned* out; CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&g_cyc, 8192 * sizeof(unsigned long long)));
    if (g_json) printf("[\n");
    for (int w : {1, 2, 4, 8}) run_all<0>(out, w);
    if (g_json) printf("\n]\n");
    return 0;
}
