// powerbench.hip -- measurement aid: what the chip draws (hwmon power1_input, PPT) and what shader clock it holds (freq1_input)
// while every SIMD issues ONE kind of vector instruction back to back (4 waves per SIMD, 8 independent chains per wave, as in
// tools/instbench.hip), and while a streaming kernel moves bytes through HBM.  The rollout kernel runs at the package power
// limit (1380 W of the 1400 W cap, tools/power_probe.py), so its speed is set by energy per env-step: this prices the
// instruction classes in joules instead of cycles.
//   hipcc --offload-arch=gfx950 -O3 -o tools/powerbench tools/powerbench.hip ; tools/powerbench [seconds per op, default 0.6]
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum Op { IDLE, MAD_U64, BITOP3, XOR, MOV, FMA_F32, MUL_F32, PK_FMA_F32, FMA_F64, MUL_F64, ADD_F64, CVT_F32_U32, CVT_F64_F32, SIN, LOG, SQRT,
          MIX_PHILOX, N_OPS };
static const char* kNames[N_OPS] = {"(idle)", "v_mad_u64_u32", "v_bitop3_b32", "v_xor_b32", "v_mov_b32", "v_fma_f32", "v_mul_f32", "v_pk_fma_f32",
    "v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f32_u32", "v_cvt_f64_f32", "v_sin_f32", "v_log_f32", "v_sqrt_f32", "philox round (2 mad + 2 bitop3)"};
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define R8(STMT) STMT(0, 1) STMT(1, 2) STMT(2, 3) STMT(3, 4) STMT(4, 5) STMT(5, 6) STMT(6, 7) STMT(7, 0)

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed, int iters) {
    unsigned a[8];
    unsigned long long d[8];
    float f[8];
    double g[8];
    f32x2 p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = (threadIdx.x + seed) * (2 * j + 3) + j;
        d[j] = a[j];
        f[j] = a[j] * 1e-9f + 0.5f;
        g[j] = f[j];
        p[j] = f32x2{f[j], f[(j + 1) & 7]};
    }
    const unsigned M = 0xD2511F53u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = (j + 1) & 7;
            if (OP == MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d[j]) : "v"((unsigned)d[j]), "s"(M) : "vcc");
            else if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %1, %2, %1 bitop3:0x96" : "=v"(a[j]) : "v"(a[j]), "v"(a[n]));
            else if (OP == XOR) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[j]) : "v"(a[j]), "v"(a[n]));
            else if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[j]) : "v"(a[n]));
            else if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[j]) : "v"(f[n]), "v"(f[(j + 2) & 7]));
            else if (OP == MUL_F32) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(f[j]) : "v"(f[j]), "v"(f[n]));
            else if (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[j]) : "v"(p[n]), "v"(p[(j + 2) & 7]));
            else if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(g[j]) : "v"(g[n]), "v"(g[(j + 2) & 7]));
            else if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(g[j]) : "v"(g[j]), "v"(g[n]));
            else if (OP == ADD_F64) asm volatile("v_add_f64 %0, %1, %2" : "=v"(g[j]) : "v"(g[j]), "v"(g[n]));
            else if (OP == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[j]) : "v"(a[j]));
            else if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(g[j]) : "v"(f[j]));
            else if (OP == SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(f[j]));
            else if (OP == LOG) asm volatile("v_log_f32 %0, %0" : "+v"(f[j]));
            else if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[j]));
            else if (OP == MIX_PHILOX && j < 2) {  // two Philox calls in flight: 2 x (2 multiplies + 2 three-input xors) per iteration
                unsigned long long p0, p1;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(a[4 * j]), "s"(M) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(a[4 * j + 2]), "s"(M) : "vcc");
                asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(a[4 * j]) : "v"((unsigned)(p1 >> 32)), "v"(a[4 * j + 1]), "v"(seed));
                asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(a[4 * j + 2]) : "v"((unsigned)(p0 >> 32)), "v"(a[4 * j + 3]), "v"(seed));
                a[4 * j + 1] = (unsigned)p1; a[4 * j + 3] = (unsigned)p0;
            }
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s ^= a[j] ^ (unsigned)d[j] ^ __float_as_uint(f[j]) ^ (unsigned)__double_as_longlong(g[j]) ^ __float_as_uint(p[j].x);
    if (s == 0x12345678u) out[0] = s;
}

__global__ __launch_bounds__(256) void stream_copy(const float4* __restrict__ src, float4* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_write(float4* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, reinterpret_cast<f32x4*>(dst) + i);
}

// ---- hwmon sampling: every amdgpu hwmon with a power1_input; the one whose power moves is the GPU this process runs on
struct Mon { std::string dir; long long p_idle = 0; };
static std::vector<Mon> g_mons;
static long long read_ll(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    long long v = -1;
    if (fscanf(f, "%lld", &v) != 1) v = -1;
    fclose(f);
    return v;
}
static void find_mons() {
    DIR* d = opendir("/sys/class/drm");
    if (!d) return;
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "card", 4) != 0 || strchr(e->d_name, '-')) continue;
        std::string base = std::string("/sys/class/drm/") + e->d_name + "/device/hwmon";
        DIR* h = opendir(base.c_str());
        if (!h) continue;
        while (dirent* x = readdir(h)) {
            if (strncmp(x->d_name, "hwmon", 5) != 0) continue;
            Mon m; m.dir = base + "/" + x->d_name;
            if (read_ll(m.dir + "/power1_input") >= 0) g_mons.push_back(m);
        }
        closedir(h);
    }
    closedir(d);
}
struct Sample { double t; std::vector<long long> p, f; };

template <typename Launch>
static void measure(const char* name, double seconds, double units_per_launch, const char* unit, Launch launch, int mon_fixed, int* mon_found) {
    std::atomic<bool> stop{false};
    std::vector<Sample> samples;
    auto t0 = std::chrono::steady_clock::now();
    std::thread th([&] {
        while (!stop.load()) {
            Sample s; s.t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            for (auto& m : g_mons) { s.p.push_back(read_ll(m.dir + "/power1_input")); s.f.push_back(read_ll(m.dir + "/freq1_input")); }
            samples.push_back(s);
            usleep(10000);
        }
    });
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    long long launches = 0;
    double gpu_ms = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 8; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        gpu_ms += ms; launches += 8;
    }
    double t_end = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop = true; th.join();
    // steady part: the second half of the run
    int mon = mon_fixed;
    if (mon < 0) {  // pick the monitor with the largest mean power in the steady part
        double best = -1;
        for (size_t j = 0; j < g_mons.size(); ++j) {
            double s = 0; int c = 0;
            for (auto& x : samples) if (x.t > 0.5 * t_end && x.t < t_end) { s += x.p[j]; ++c; }
            if (c && s / c > best) { best = s / c; mon = (int)j; }
        }
        if (mon_found) *mon_found = mon;
    }
    double ps = 0, fs = 0; int c = 0;
    for (auto& x : samples) if (x.t > 0.5 * t_end && x.t < t_end && mon >= 0) { ps += x.p[mon]; fs += x.f[mon]; ++c; }
    const double watts = c ? ps / c * 1e-6 : -1, mhz = c ? fs / c * 1e-6 : -1;
    const double per_launch_us = launches ? gpu_ms * 1e3 / launches : 0;
    printf("%-34s %8.1f W  sclk %6.0f MHz  %9.2f us/launch  %10.3f %s\n", name, watts, mhz, per_launch_us,
           per_launch_us > 0 ? units_per_launch / (per_launch_us * 1e-6) * 1e-9 : 0.0, unit);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int OP>
static void run_op(unsigned* out, double seconds, int mon) {
    const int iters = 20000;  // 160 000 instructions per wave and launch: ~0.1 - 0.5 ms
    const double winst = (OP == MIX_PHILOX ? 8.0 : 8.0) * iters * 4096.0;  // wave-instructions per launch (4096 waves)
    if (OP == IDLE) { measure(kNames[OP], seconds, 0, "-", [] { usleep(2000); }, mon, nullptr); return; }
    measure(kNames[OP], seconds, winst, "G wave-instr/s", [&] { hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, out, 1u, iters); }, mon, nullptr);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 0.6;
    find_mons();
    printf("# %zu hwmon power sensors; each line: mean of the second half of a %.1f s run\n", g_mons.size(), seconds);
    unsigned* out; CK(hipMalloc(&out, 4096));
    int mon = -1;
    // find this GPU's sensor with a heavy kernel
    measure("(sensor search: v_fma_f64)", 1.0, 8.0 * 20000 * 4096, "G wave-instr/s",
            [&] { hipLaunchKernelGGL(k<FMA_F64>, dim3(1024), dim3(256), 0, 0, out, 1u, 20000); }, -1, &mon);
    printf("# using sensor %d: %s (cap %.0f W)\n", mon, mon >= 0 ? g_mons[mon].dir.c_str() : "-",
           mon >= 0 ? read_ll(g_mons[mon].dir + "/power1_cap") * 1e-6 : 0.0);
    run_op<IDLE>(out, seconds, mon);
    run_op<MOV>(out, seconds, mon); run_op<XOR>(out, seconds, mon); run_op<BITOP3>(out, seconds, mon); run_op<MAD_U64>(out, seconds, mon);
    run_op<MIX_PHILOX>(out, seconds, mon);
    run_op<MUL_F32>(out, seconds, mon); run_op<FMA_F32>(out, seconds, mon); run_op<PK_FMA_F32>(out, seconds, mon);
    run_op<ADD_F64>(out, seconds, mon); run_op<MUL_F64>(out, seconds, mon); run_op<FMA_F64>(out, seconds, mon);
    run_op<CVT_F32_U32>(out, seconds, mon); run_op<CVT_F64_F32>(out, seconds, mon);
    run_op<LOG>(out, seconds, mon); run_op<SQRT>(out, seconds, mon); run_op<SIN>(out, seconds, mon);
    // HBM streams: 1 GiB copy (read + write) and 1 GiB non-temporal write
    const long long n = (1ll << 30) / 16;
    float4 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMemset(a, 0, n * 16));
    measure("HBM copy (1 GiB read + 1 GiB write)", seconds, 2.0 * n * 16, "GB/s",
            [&] { hipLaunchKernelGGL(stream_copy, dim3(4096), dim3(256), 0, 0, a, b, n); }, mon, nullptr);
    measure("HBM non-temporal write (1 GiB)", seconds, 1.0 * n * 16, "GB/s",
            [&] { hipLaunchKernelGGL(stream_write, dim3(4096), dim3(256), 0, 0, b, n); }, mon, nullptr);
    run_op<IDLE>(out, seconds, mon);
    return 0;
}
