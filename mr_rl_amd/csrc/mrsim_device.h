// mrsim_device.h -- device-side building blocks of the MR_env.step() hot path
// for gfx950 (MI355X, wave64).  One lane advances one environment.
//
// What is restated here (reference citations are /root/reference/<file>:<line>):
//   Simulator.simulate      MR_simulator.py:58-88   the ODE right-hand side (+ per-eval noise)
//   Simulator.step          MR_simulator.py:36-52   integrate one time_span with SciPy RK45,
//                                                    then build the NEXT RK45 object
//   scipy RK45 semantics    rk.py rk_step/_step_impl, common.py select_initial_step
//   MR_Env.step/end/...     MR_env.py:70-152
//
// Algebra used by the kernel.  The RHS ignores (t, y), so every stage value is
//   K[i] = V(action) + N_i ,   N_i = fresh noise of that RHS evaluation,
// except K[0], the derivative the RK45 object computed when it was constructed -- at the
// end of the PREVIOUS env step, with the previous action; it is carried in HBM.
// With the Dormand-Prince weights  sum(B) = 1, sum(E) = 0, B[1] = E[1] = 0:
//   y_new = y + h * ( V + B0*(K0 - V) + sum_{i=2..5} B_i N_i )
//   err   =     h * (     E0*(K0 - V) + sum_{i=2..6} E_i N_i )
// The two noise sums are accumulated in fp32 (they are sums of fp32 normals); everything that
// touches positions is fp64.  sigma = 0 reduces to the closed form of SURVEY 3.2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrsim {

#ifndef MRSIM_ROLLOUT_TABLE
#define MRSIM_ROLLOUT_TABLE 1
#endif
#ifndef MRSIM_PRIO_MODE  // wave-priority rotation of the rollout kernel: 0 = none, 1 = every step, 2 / 4 = every 2nd / 4th step
#define MRSIM_PRIO_MODE 1
#endif
#ifndef MRSIM_FAST_STEP  // A/B switch: 0 = no straight-line fast step, every env step takes the general path
#define MRSIM_FAST_STEP 1
#endif
#ifndef MRSIM_AB_CHEAP  // A/B switch (tools/ab_rollout.py): 0 = the cheap first-level tests are predicted not-taken and skipped
#define MRSIM_AB_CHEAP 1
#endif

constexpr int kBlock = 256;  // 4 waves; 16-B records => every wave moves 1 KiB per array

// noise generator variants (template parameter NZ)
constexpr int kNoNoise = 0;    // sigma == 0: no RNG work at all
constexpr int kNoiseSpec = 1;  // Box-Muller in specified fp32 arithmetic: bit-identical to the CPU oracle
constexpr int kNoiseFast = 2;  // Box-Muller on the hardware transcendentals (v_log/v_sqrt/v_sin/v_cos)
// the same two with the COLLAPSED noise law (MrsimParams.noise_law; RK45 only): per rk_step attempt the B- and E-weighted sums
// of the stage noise are drawn directly from their joint Gaussian instead of stage by stage -- see "collapsed law" below
constexpr int kNoiseSpecC = 3;
constexpr int kNoiseFastC = 4;
__host__ __device__ constexpr bool nz_fast(int nz) { return nz == kNoiseFast || nz == kNoiseFastC; }
__host__ __device__ constexpr bool nz_coll(int nz) { return nz == kNoiseSpecC || nz == kNoiseFastC; }

// ---------------------------------------------------------------------------
// kernel-side parameter block (passed by value -> kernarg/SGPRs)
// ---------------------------------------------------------------------------
struct KParams {
    double dt, inv_dt, rtol, atol, a0, sigma, sigma4;
    double min_dist2;  // min_dist2goal^2
    double obs_lo[4], obs_hi[4];
    double dmin2, dmax2;  // obs_low[4]^2 (0 if <= 0), obs_high[4]^2
    double sym_bound;     // valid when sym_bounds: obs_lo[j] == -sym_bound, obs_hi[j] == sym_bound for j < 4
    double init_lo[2], init_span[2];
    float act_lo_f[2], act_span_f[2];
    double h1_thresh;  // 0.01 / dt^5 : select_initial_step's h1 >= dt  <=>  max(d1,d2) <= h1_thresh
    double h1_thresh_m;    // h1_thresh / 1.05                  (first-level tests, 5 % margins)
    double k_h0;           // max(105 * dt, 1): |y| >= k_h0 * F  =>  0.01 * d0/d1 >= dt, i.e. h0 == dt, and |y| >= 2e-5 max(scale)
    double gmax_dt;        // 2 * Zmax * sigma / dt: worst case of |f1 - f0| / dt under the nominal law
    double reset_fmin;     // >= 0: over the whole init box construct_level0 of a nominal reset constructor reduces to
                           // max|f0| >= reset_fmin (make_kparams certifies the other three conditions); < 0: not certified
    double zmax2_dt;       // 2 * Zmax / dt
    double zmax_e6_sigma;  // Zmax * E6 * sigma: worst-case contribution of K6 to the error estimate
    float h1_thresh2_f, dt2_f;
    // level -1 test of the fused rollout's fast step (rk45_fast_step, Lm1): fp32 copies with their safety margins folded in
    float lm_a0, lm_sigma, lm_da, lm_dk, lm_dc, lm_es, lm_ed, lm_rt, lm_at, lm_kh, lm_mg;
    double lm_ccap;
    double acc_lim_dt2;    // 1.96 / dt^2: the exact accept test of a whole-interval attempt with h = dt moved to the right-hand side
    int32_t substeps, reward_mode, max_timesteps, auto_reset, goal_K, goal_T;
    int32_t integrator;
    uint32_t flags;  // kF* bits: every wave-uniform yes/no of the launch in ONE scalar register
    uint32_t seed_lo, seed_hi, step_lo, step_hi;
    uint32_t env_id0, pad2;
    long long n;
    const unsigned long long* step_base;  // optional device word added to (step_hi:step_lo)
};

// Wave-uniform launch conditions.  Testing a bit of one SGPR at the point of use (s_bitcmp + branch) costs
// nothing; letting the compiler hoist a dozen `ptr != nullptr` / `mode == x` comparisons out of the time
// loop makes it keep each one as a 64-bit lane mask, which overflows the SGPR file and comes back through
// v_readlane on the hot path.  live_flags() launders the word so the tests stay where they are written.
enum : uint32_t {
    kFAutoReset = 1u << 0, kFRewardGoal = 1u << 1, kFSymBounds = 1u << 2, kFGoalTable = 1u << 3,
    kFStepBase = 1u << 4, kFActions = 1u << 5, kFSharedActions = 1u << 6, kFObsAos = 1u << 7,
    kFOutTraj = 1u << 8, kFOutStatePrime = 1u << 9, kFOutObs = 1u << 10, kFOutRew = 1u << 11,
    kFOutDone = 1u << 12, kFOutActions = 1u << 13, kFOutFinalRet = 1u << 14, kFOutFinalLen = 1u << 15,
    kFOutFinalObs = 1u << 16, kFOutStatus = 1u << 17, kFRk4 = 1u << 18, kFCarry64 = 1u << 19, kFActions64 = 1u << 20,
    // policy source = the in-kernel DDPG actor (mrsim_actor.h); + Ornstein-Uhlenbeck exploration noise; + OU state zeroed at
    // the first step of every episode (the reference never resets the process, RL/MR_ddpg.py:270-311)
    kFActor = 1u << 21, kFActorOU = 1u << 22, kFOUReset = 1u << 23,
    kFResetFresh = 1u << 24,  // auto-reset = a fresh MR_Env (nominal-law constructor) instead of the re-used one
    kFActorBf16 = 1u << 25,   // the actor's 64 x 64 layer in bf16 x 3 arithmetic (mrsim_actor.h: kActBf16x3)
    kFActorBf16s = 1u << 26,  // ... in plain bf16 (kActBf16)
    kFOutAttempts = 1u << 27, // step kernel: rk_step attempts of the env step (MrsimStepIO.attempts; generic kernels only)
    kFStatusPlain = 1u << 28, // step kernel, n == 1: the status word is OR-ed with a plain load / store (one lane writes; the word may
                              // live in pinned host memory -- mr_rl_amd/env.py -- where a device atomic would need PCIe atomics)
};
__device__ __forceinline__ uint32_t live_flags(uint32_t f) {
    asm volatile("" : "+s"(f));
    return f;
}

// ---------------------------------------------------------------------------
// RNG: Philox4x32-10, counter = {c0, step lo, step hi, global env id}, key = seed
// c0 = stream<<28 | block<<4 | call.  Same definition as oracle/mrsim_oracle.c.
// ---------------------------------------------------------------------------
enum : uint32_t { kStreamDyn = 0, kStreamCtor = 1, kStreamResetPos = 2, kStreamResetCtor = 3, kStreamPolicy = 4 };
__device__ __forceinline__ constexpr uint32_t c0_of(uint32_t stream, uint32_t block, uint32_t call) {
    return (stream << 28) | (block << 4) | call;
}

#ifndef MRSIM_PHILOX_ROUNDS     // measurement builds only (tools/ab_rollout.py): what the generator's ten rounds cost at the power
#define MRSIM_PHILOX_ROUNDS 10  // cap.  The product and the oracle use 10 (Random123's default; 7 is its Crush-resistant minimum).
#endif
struct Rng {
    uint32_t k0, k1, step_lo, step_hi, env;
};

// effective 64-bit step index of this launch (+ t for the fused rollout)
__device__ __forceinline__ void step_words(const KParams& P, unsigned long long t, uint32_t& lo, uint32_t& hi) {
    unsigned long long s = (((unsigned long long)P.step_hi << 32) | P.step_lo) + t;
    if (P.flags & kFStepBase) s += *P.step_base;  // uniform scalar load
    lo = (uint32_t)s;
    hi = (uint32_t)(s >> 32);
}

__device__ __forceinline__ Rng make_rng(const KParams& P, long long i, unsigned long long t = 0) {
    Rng R;
    R.k0 = P.seed_lo; R.k1 = P.seed_hi;
    step_words(P, t, R.step_lo, R.step_hi);
    R.env = P.env_id0 + (uint32_t)i;
    return R;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < MRSIM_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // Rounds 0-2 still have wave-uniform operands (step index, seed, call id live in SGPRs) and are
        // left to the scalar unit; from round 3 on everything is per-lane and the three-input xor is one
        // VALU op (v_bitop3_b32, truth table 0x96; gfx950 has no v_xor3).
        uint32_t n0, n2;
        if (r < 3) {
            n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
            n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        } else {
            n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
            n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        }
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ __forceinline__ void philox_call(const Rng& R, uint32_t c0, uint32_t (&o)[4]) {
    // The key is laundered through an empty asm so that the 20 round keys (k + r*W) are re-derived by
    // two s_add per round at every call.  Otherwise they are hoisted out of the rollout's time loop as
    // loop invariants, overflow the SGPR file and come back through v_readlane on the hot path.
    uint32_t k0 = R.k0, k1 = R.k1;
    asm volatile("" : "+s"(k0), "+s"(k1));
    philox4x32_10(c0, R.step_lo, R.step_hi, R.env, k0, k1, o);
}

// N independent Philox calls advanced round by round, call-innermost: 2N multiplies are in flight per round,
// which hides the v_mad_u64_u32 -> v_bitop3 -> v_mad dependency chain that a single call (ILP 2) exposes at
// 4 waves per SIMD.  Same words as N separate philox_call()s.
template <int N>
__device__ __forceinline__ void philox_multi(const Rng& R, const uint32_t (&c0s)[N], uint32_t (&o)[N][4]) {
    uint32_t k0 = R.k0, k1 = R.k1;
    asm volatile("" : "+s"(k0), "+s"(k1));  // see philox_call
    uint32_t c[N][4];
#pragma unroll
    for (int j = 0; j < N; ++j) { c[j][0] = c0s[j]; c[j][1] = R.step_lo; c[j][2] = R.step_hi; c[j][3] = R.env; }
#pragma unroll
    for (int r = 0; r < MRSIM_PHILOX_ROUNDS; ++r) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c[j][0];
            const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[j][2];
            uint32_t n0, n2;
            if (r < 3) {
                n0 = (uint32_t)(p1 >> 32) ^ c[j][1] ^ k0;
                n2 = (uint32_t)(p0 >> 32) ^ c[j][3] ^ k1;
            } else {
                n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c[j][1], k0, 0x96);
                n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c[j][3], k1, 0x96);
            }
            c[j][1] = (uint32_t)p1; c[j][3] = (uint32_t)p0; c[j][0] = n0; c[j][2] = n2;
        }
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) { o[j][0] = c[j][0]; o[j][1] = c[j][1]; o[j][2] = c[j][2]; o[j][3] = c[j][3]; }
}

// ln(u), u in [2^-33, 1]; Cephes logf polynomial, explicit fma => bit-identical to the oracle.
__device__ __forceinline__ float spec_logf(float u) {
    const uint32_t b = __float_as_uint(u);
    int e = (int)(b >> 23) - 127;
    float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m *= 0.5f; e += 1; }
    const float t = m - 1.0f;
    const float z = t * t;
    float p = 7.0376836292E-2f;
    p = __builtin_fmaf(p, t, -1.1514610310E-1f);
    p = __builtin_fmaf(p, t, 1.1676998740E-1f);
    p = __builtin_fmaf(p, t, -1.2420140846E-1f);
    p = __builtin_fmaf(p, t, 1.4249322787E-1f);
    p = __builtin_fmaf(p, t, -1.6668057665E-1f);
    p = __builtin_fmaf(p, t, 2.0000714765E-1f);
    p = __builtin_fmaf(p, t, -2.4999993993E-1f);
    p = __builtin_fmaf(p, t, 3.3333331174E-1f);
    float y = (t * z) * p;
    y = __builtin_fmaf(-0.5f, z, y);
    const float lm = t + y;
    return __builtin_fmaf((float)e, 0.693147180559945f, lm);
}

// IEEE-correct sqrt for w in [0, 64): v_sqrt_f32 (1 ulp) + the one-ulp fix-up hipcc's own sqrtf
// uses, without its denormal/inf handling (never needed here).  Bit-identical to sqrtf().
__device__ __forceinline__ float sqrt_rn_small(float w) {
    const float s = __builtin_amdgcn_sqrtf(w);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u);
    const float sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, w);
    const float rp = __builtin_fmaf(-sp, s, w);
    float r = (rm <= 0.0f) ? sm : s;
    r = (rp > 0.0f) ? sp : r;
    return r;
}

// Box-Muller.  kNoiseSpec: specified operation by operation (oracle/mrsim_oracle.c: orc_box_muller),
// bit-identical to the oracle.  kNoiseFast: same uniforms, same formula, hardware transcendentals
// (v_log_f32, v_sqrt_f32, v_sin_f32/v_cos_f32 take the angle in revolutions): within 1e-6 + 7.5e-7 r of the spec.
template <int NZ>
__device__ __forceinline__ void box_muller(uint32_t ua, uint32_t ub, float& z0, float& z1, float* radius = nullptr) {
    const float u = __builtin_fmaf((float)ua, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    if constexpr (nz_fast(NZ)) {
        const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));  // -2 ln2 log2 u
        // angle in revolutions for v_sin / v_cos, which are 1-periodic: 1 + (ub >> 9) * 2^-23 in [1, 2), assembled by ONE
        // v_alignbit_b32 ({0x7F, ub} >> 9 = 0x3F800000 | ub >> 9) instead of v_cvt_f32_u32 + v_fma.  Top 23 bits of ub:
        // within 2^-23 revolutions of the spec's (ub + 0.5) 2^-32, i.e. a normal within 7.5e-7 r of the spec's.
        const float t = __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, ub, 9u));
        z0 = r * __builtin_amdgcn_cosf(t);
        z1 = r * __builtin_amdgcn_sinf(t);
        if (radius != nullptr) *radius = r;
    } else {
        const float r = sqrt_rn_small(-2.0f * spec_logf(u));
        const uint32_t oct = ub >> 29;
        uint32_t rem = ub & 0x1FFFFFFFu;
        if (oct & 1u) rem = 0x1FFFFFFFu - rem;
        const float x = __builtin_fmaf((float)rem, 1.862645149230957e-09f, 9.313225746154785e-10f);
        const float phi = x * 0.78539816339744831f;
        const float zz = phi * phi;
        float ps = -1.9515295891E-4f;
        ps = __builtin_fmaf(ps, zz, 8.3321608736E-3f);
        ps = __builtin_fmaf(ps, zz, -1.6666654611E-1f);
        const float s = __builtin_fmaf(ps * zz, phi, phi);
        float pc = 2.443315711809948E-005f;
        pc = __builtin_fmaf(pc, zz, -1.388731625493765E-003f);
        pc = __builtin_fmaf(pc, zz, 4.166664568298827E-002f);
        const float c = __builtin_fmaf(pc * zz, zz, __builtin_fmaf(-0.5f, zz, 1.0f));
        const uint32_t swap = ((oct + 1u) >> 1) & 1u;
        const uint32_t cneg = ((oct + 2u) >> 2) & 1u;
        const uint32_t sneg = oct >> 2;
        float cc = swap ? s : c;
        float ss = swap ? c : s;
        if (cneg) cc = -cc;
        if (sneg) ss = -ss;
        z0 = r * cc;
        z1 = r * ss;
        if (radius != nullptr) *radius = r;
    }
}

// the first NCALLS*4 normals of one block (rk_step attempt / constructor / fixed sub-step)
template <int NZ, int NCALLS>
__device__ __forceinline__ void block_normals(const Rng& R, uint32_t c0base, float (&z)[NCALLS * 4]) {
#pragma unroll
    for (int j = 0; j < NCALLS; ++j) {
        uint32_t o[4];
        philox_call(R, c0base | (uint32_t)j, o);
        box_muller<NZ>(o[0], o[1], z[4 * j + 0], z[4 * j + 1]);
        box_muller<NZ>(o[2], o[3], z[4 * j + 2], z[4 * j + 3]);
    }
}

// normals of NCALLS calls whose Philox words are already there
template <int NZ, int NCALLS>
__device__ __forceinline__ void normals_from_words(const uint32_t (*w)[4], float (&z)[NCALLS * 4]) {
#pragma unroll
    for (int j = 0; j < NCALLS; ++j) {
        box_muller<NZ>(w[j][0], w[j][1], z[4 * j + 0], z[4 * j + 1]);
        box_muller<NZ>(w[j][2], w[j][3], z[4 * j + 2], z[4 * j + 3]);
    }
}

// ---------------------------------------------------------------------------
// small fp64 math, written for few registers (ocml's pow/sincos inline to hundreds of
// instructions and ~150 VGPRs, which caps the kernel at 3 waves/SIMD)
// ---------------------------------------------------------------------------
// x^(-1/5): fp32 exp2/log2 seed refined by two Newton steps y <- y + y(1 - x y^5)/5 (quadratic:
// 1e-6 -> 1e-12 -> fp64 rounding).  0, inf and NaN seeds are returned as they are, which keeps
// pow()'s limits: x -> 0 gives +inf, x -> inf gives 0 (the callers clamp with min/max).
__device__ __forceinline__ double inv_fifth_root(double x) {
    const float g = __builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf((float)x));
    if (!(g > 0.0f && g < __builtin_inff())) return (double)g;
    double y = (double)g;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double y2 = y * y;
        const double y5 = y2 * y2 * y;
        y = __builtin_fma(y * 0.2, __builtin_fma(-x, y5, 1.0), y);
    }
    return y;
}
// x^(1/5) = x * (x^(-1/5))^4
__device__ __forceinline__ double fifth_root(double x) {
    const double r = inv_fifth_root(x);
    const double r2 = r * r;
    return x * (r2 * r2);
}

// a*b + c with the wave-uniform constant c read straight from a scalar register pair: hipcc otherwise
// emits v_fmac_f64 and has to v_mov every 64-bit Horner coefficient into the accumulator first.
__device__ __forceinline__ double fma_sc(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}

// sin and cos of a double: two-term fma Cody-Waite reduction by pi/2 (exact products, so the reduced
// argument is good to ~2e-16 absolute for |a| < 2^30) + fdlibm __kernel_sin/__kernel_cos polynomials on
// [-pi/4, pi/4].  NaN/inf give NaN; |a| >= 2^30 rad is outside the supported range (the quadrant
// conversion saturates) -- numpy would Payne-Hanek there.
__device__ __forceinline__ void sincos_f64(double a, double& s, double& c) {
    const double k = __builtin_rint(a * 0.63661977236758138);
    double r = __builtin_fma(-k, 1.5707963267948966, a);
    r = __builtin_fma(-k, 6.123233995736766e-17, r);
    const double z = r * r;
    double ps = fma_sc(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma_sc(ps, z, 2.75573137070700676789e-06);
    ps = fma_sc(ps, z, -1.98412698298579493134e-04);
    ps = fma_sc(ps, z, 8.33333333332248946124e-03);
    ps = fma_sc(ps, z, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(ps * z, r, r);
    double pc = fma_sc(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma_sc(pc, z, -2.75573143513906633035e-07);
    pc = fma_sc(pc, z, 2.48015872894767294178e-05);
    pc = fma_sc(pc, z, -1.38888888888741095749e-03);
    pc = fma_sc(pc, z, 4.16666666666666019037e-02);
    const double cr = __builtin_fma(pc * z, z, __builtin_fma(-0.5, z, 1.0));
    const int q = (int)k;
    const double s0 = (q & 1) ? cr : sr;
    const double c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// Table-driven sin/cos for the fused rollout kernel: a = k*(2 pi/1024) + r, |r| <= pi/1024, {sin, cos}(k) from a
// 16 KiB table the block keeps in LDS (correctly rounded fp64 entries, tools/gen_sincos_table.py), sin r and cos r
// from degree-5 / degree-4 Taylor polynomials (truncation 2.3e-15 / 1.2e-18), recombined by angle addition.
// 15 fp64 operations + one ds_read_b128 instead of 24 fp64 + 15 quadrant-selection operations; same accuracy class
// as sincos_f64 (<= 2 ulp).
#include "mrsim_sincos_table.h"
__device__ __forceinline__ void sincos_tab(const double2* __restrict__ tab, double a, double& s, double& c) {
    // Branch-free on purpose: a range check with a polynomial fall-back would end the basic block and keep the scheduler
    // from interleaving this dependent fp64 chain with the Philox tail and the Box-Muller code around it (only four waves
    // share a SIMD at N = 262 144, so in-wave ILP is what hides the fp64 latency).  The table index is reduced in fp64
    // (k - 1024 rint(k / 1024), exact), so it is right for every finite angle whose k = rint(a * 1024 / 2 pi) is an exact
    // integer (|a| < 2.7e13 rad); r = a - k * (2 pi / 1024) is formed with exact products (fma) and is good to 1e-16 up to
    // |a| ~ 1e9 rad; NaN / inf give NaN.
    const double k = __builtin_rint(a * MRSIM_SINCOS_INV_STEP);
    double r = __builtin_fma(-k, MRSIM_SINCOS_STEP_HI, a);
    r = __builtin_fma(-k, MRSIM_SINCOS_STEP_LO, r);
    const double kk = __builtin_fma(-(double)MRSIM_SINCOS_N, __builtin_rint(k * (1.0 / MRSIM_SINCOS_N)), k);  // in [-512, 512]
    const double2 t = tab[(int)kk & (MRSIM_SINCOS_N - 1)];
    const double z = r * r;
    const double ps = __builtin_fma(z, 1.0 / 120, -1.0 / 6) * z;
    const double sr = __builtin_fma(ps, r, r);
    const double cr = __builtin_fma(__builtin_fma(z, 1.0 / 24, -0.5), z, 1.0);
    s = __builtin_fma(t.y, sr, t.x * cr);
    c = __builtin_fma(t.y, cr, -(t.x * sr));
}

// scipy common.norm of a 2-vector
__device__ __forceinline__ double rms2(double a, double b) { return sqrt(a * a + b * b) / 1.4142135623730951; }

// fmax / fmin of two doubles (or of their magnitudes) as ONE instruction.  hipcc's fmax() / fmin() canonicalise each operand
// first (v_max_f64 x, x, x: quieting signalling NaNs), three fp64-rate instructions where one does; v_max_f64 / v_min_f64
// themselves return the non-NaN operand, which is all the first-level tests below rely on (a NaN fails a later comparison).
__device__ __forceinline__ double max2(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double min2(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double max_abs2(double a, double b) { double r; asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double min_abs2(double a, double b) { double r; asm("v_min_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }

// ---------------------------------------------------------------------------
// Simulator.simulate (MR_simulator.py:58-88) as  K = V + N:
//   nominal:     V = (a0*f)*(cos a, sin a),            N = sigma*(z_x, z_y)                  :82-83
//   mismatched:  a0b = a0 + (f/4)*0.8                                                        :55-56
//                V = ((a0b*f)*cos(a+0.1) + 0.2, (a0b*f)*sin(a-0.15) - 0.1)                   :79-80
//                N = (gx*z_a + sigma*z_x, gy*z_a + sigma*z_y),  (gx, gy) = (sigma/4)*f*(cos(a+0.1), sin(a-0.15))
// ---------------------------------------------------------------------------
template <bool MIS>
struct RhsCtx {
    double vx, vy;
    double gx, gy;  // mismatched only
    // mismatched model under the collapsed law: fp32 copies of g and c = 1 / (sigma + sqrt(sigma^2 + |g|^2)), the coefficient of
    // the 2-D noise map M(u) = sigma u + c g (g . u) (see "collapsed law" below); dead code elsewhere
    float gxf, gyf, cg;
};

template <bool MIS>
__device__ __forceinline__ RhsCtx<MIS> make_ctx(const KParams& P, double f_t, double al,
                                                const double2* __restrict__ tab = nullptr) {
    RhsCtx<MIS> C;
    if constexpr (MIS) {
        // a0 + (f/4)*0.8 (MR_simulator.py:55-56) as ONE multiply: the double 0.2 is the double 0.8 with its exponent lowered by two,
        // so f * 0.2 and (f / 4) * 0.8 round the same real number -- equal bits (barring denormal f)
        const double a0b = P.a0 + f_t * 0.2;      // (product rounded, then the sum: as the reference; -ffp-contract=off)
        // cos(al + 0.1) and sin(al - 0.15) from ONE sincos by the angle-addition formulas (<= 2 ulp from the
        // direct evaluation; an fp64 sincos costs about as much as a Philox call)
        constexpr double kC01 = 0.99500416527802577, kS01 = 0.099833416646828152;   // cos, sin of 0.1
        constexpr double kC015 = 0.98877107793604229, kS015 = 0.14943813247359922;  // cos, sin of 0.15
        double s, c;
#if MRSIM_ROLLOUT_TABLE
        sincos_tab(tab, al, s, c);
#else
        sincos_f64(al, s, c);
#endif
        const double cA = __builtin_fma(c, kC01, -(s * kS01));
        const double sB = __builtin_fma(s, kC015, -(c * kS015));
        const double af = a0b * f_t;
        C.vx = af * cA + 0.2;
        C.vy = af * sB - 0.1;
        C.gx = (P.sigma4 * f_t) * cA;
        C.gy = (P.sigma4 * f_t) * sB;
        C.gxf = (float)C.gx; C.gyf = (float)C.gy;
        const float sf = (float)P.sigma;
        C.cg = __builtin_amdgcn_rcpf(sf + __builtin_amdgcn_sqrtf(__builtin_fmaf(sf, sf, __builtin_fmaf(C.gxf, C.gxf, C.gyf * C.gyf))));
    } else {
        double s, c;
#if MRSIM_ROLLOUT_TABLE
        sincos_tab(tab, al, s, c);
#else
        sincos_f64(al, s, c);
#endif
        const double af = P.a0 * f_t;
        C.vx = af * c;
        C.vy = af * s;
        C.gx = C.gy = 0.0;
        C.gxf = C.gyf = C.cg = 0.f;
    }
    return C;
}

// zero action (Simulator.reset_start_pos, MR_simulator.py:30): the velocity terms vanish
template <bool MIS>
__device__ __forceinline__ RhsCtx<MIS> zero_ctx(const KParams&) {
    RhsCtx<MIS> C;
    C.vx = MIS ? 0.2 : 0.0;
    C.vy = MIS ? -0.1 : 0.0;
    C.gx = C.gy = 0.0;
    C.gxf = C.gyf = C.cg = 0.f;
    return C;
}

// N for one RHS evaluation from its normals (za only used when MIS)
template <bool MIS>
__device__ __forceinline__ void noise_vec(const KParams& P, const RhsCtx<MIS>& C, float za, float zx, float zy,
                                          double& nx, double& ny) {
    nx = P.sigma * (double)zx;
    ny = P.sigma * (double)zy;
    if constexpr (MIS) {
        nx = __builtin_fma(C.gx, (double)za, nx);
        ny = __builtin_fma(C.gy, (double)za, ny);
    }
}

// K = V + N for one RHS evaluation, one fma per term
template <bool MIS>
__device__ __forceinline__ void rhs_value(const KParams& P, const RhsCtx<MIS>& C, float za, float zx, float zy,
                                          double& kx, double& ky) {
    kx = __builtin_fma(P.sigma, (double)zx, C.vx);
    ky = __builtin_fma(P.sigma, (double)zy, C.vy);
    if constexpr (MIS) {
        kx = __builtin_fma(C.gx, (double)za, kx);
        ky = __builtin_fma(C.gy, (double)za, ky);
    }
}

// Mismatched model under the collapsed law: the a0 draw is not a normal of its own -- the evaluation's noise 2-vector is
// M(u) = sigma u + c g (g . u) for a standard normal pair u (exact in law: its covariance is sigma^2 I + g g^T, what the three
// normals of MR_simulator.py:55-56,77-80 give).  In the (z_a, z_x, z_y) form every formula of this file uses, that is
// z_x = u_x, z_y = u_y and the DERIVED  z_a = c (g . u):
template <bool MIS>
__device__ __forceinline__ float derived_za(const RhsCtx<MIS>& C, float ux, float uy) {
    return C.cg * __builtin_fmaf(C.gxf, ux, C.gyf * uy);
}

// progress of the sub-step loop of one env step (+ what the constructor needs from its last attempt)
struct SubStep {
    double tau, h_abs;
    uint32_t attempt;
    bool rejected;
    // F0 / F1 of the last accepted attempt's block (see attempt_noise): (f0a, f0b) = the words of F0's first
    // Box-Muller pair, f0y = F0's third normal (mismatched law only, already evaluated by the attempt);
    // w3 = the block's lazily fetched last call (call 3 nominal, call 5 mismatched)
    uint32_t last_attempt, f0a, f0b, w3[4];
    float f0y;
    bool have3;
};

// ---- first-level test of select_initial_step (fp64, a dozen operations, no conversions): a SUFFICIENT condition for
// h_abs == interval.  With m = min(|x|,|y|), sm / sM = the smaller / larger scale, F = max |f0|, G >= max |f1 - f0|:
//   d0 >= m / sm,  F / (sM sqrt2) <= d1 <= F / sm,  d2 <= G / (sm h0),  h0 = min(0.01 d0/d1, dt) >= min(0.01 m/F, dt)
//   F >= 2e-5 sM and m >= 2e-5 sM   =>  d1 >= 1.4e-5, d0 >= m / sm >= 2e-5          (no 1e-6 branch)
//   TH sm / 1.05 >= F               =>  d1 <= TH
// nominal law (|f0| <= a0 |f| + noise, a few tens):
//   m >= 105 dt F  =>  0.01 d0/d1 >= dt  =>  h0 = dt;   TH sm / 1.05 >= G / dt  =>  d2 <= TH  =>  h1 >= dt
// mismatched law (|f0| up to 100: m >= 3.15 F fails for half the actions at |y| ~ 100), the same without h0 = dt:
//   m >= 1.05 dt F  =>  100 h0 >= dt;   d2 <= G / (sm min(0.01 m / F, dt)) <= TH  <=  TH sm / 1.05 >= G / dt  and
//   (TH sm / 1.05) m >= 105 dt (G / dt) F
// Gd = G / dt: exact when f1 was evaluated; otherwise its worst case over |z| <= Zmax (construct_gd_bound):
// |f1 - f0| <= 2 Zmax (sigma + |g|).  Anything that fails -- NaNs included, every comparison is false on them -- goes on
// to the fp32 test of rk45_construct and from there to the exact formulas.
template <bool MIS>
__device__ __forceinline__ double construct_gd_bound(const KParams& P, const RhsCtx<MIS>& C) {
    if constexpr (MIS) return __builtin_fma(max_abs2(C.gx, C.gy), P.zmax2_dt, P.gmax_dt);
    else return P.gmax_dt;
}
template <bool MIS>
__device__ __forceinline__ bool construct_level0(const KParams& P, double x, double y, double sc0, double sc1, double f0x,
                                                 double f0y, double Gd) {
    const double F = max_abs2(f0x, f0y);
    if constexpr (MIS) {
        // scale is increasing in |coordinate|: min / max of the two scales are the scales of min / max |coordinate| (same fma, same
        // bits as sc0 / sc1, which this branch leaves unused)
        const double m = min_abs2(x, y), M = max_abs2(x, y);
        const double c = 2e-5 * __builtin_fma(M, P.rtol, P.atol);
        const double u = P.h1_thresh_m * __builtin_fma(m, P.rtol, P.atol);
        const double K = max2(0.01 * P.k_h0 * F, c);      // |x| >= K and |y| >= K  <=>  min(|x|, |y|) >= K
        return (m >= K) && (F >= c) && (u >= max2(F, Gd)) && (u * m >= P.k_h0 * (Gd * F));
    }
    const double c = 2e-5 * max2(sc0, sc1);
    const double u = P.h1_thresh_m * min2(sc0, sc1);
    {
        const double kF = P.k_h0 * F;
        return (__builtin_fabs(x) >= kF) && (__builtin_fabs(y) >= kF) && (F >= c) && (u >= max2(F, Gd));
    }
}

// the three normals (z_a, z_x, z_y) of the constructor's F0 from what the attempt kept: the words of one Box-Muller pair
// (f0a, f0b) and, under the mismatched law, a third normal f0k already evaluated with its pair partner
//   per-stage law: pair = (F0a, F0x), f0k = F0y        collapsed law: pair = (F0x, F0y), z_a derived from the pair (derived_za)
template <int NZ, bool MIS>
__device__ __forceinline__ void f0_normals(const RhsCtx<MIS>& C, uint32_t f0a, uint32_t f0b, float f0k, float& za, float& zx, float& zy,
                                           float* radius = nullptr) {
    float p, q;
    box_muller<NZ>(f0a, f0b, p, q, radius);
    if constexpr (!MIS) { za = 0.f; zx = p; zy = q; }
    else if constexpr (nz_coll(NZ)) { zx = p; zy = q; za = derived_za<MIS>(C, p, q); }
    else { za = p; zx = q; zy = f0k; }
}

// ---------------------------------------------------------------------------
// RungeKutta.__init__ + select_initial_step: what Simulator.step does after integrating
// (MR_simulator.py:46-50) and what reset_start_pos does (:31-34).  Two RHS evaluations:
// f0 (-> K[0] of the next step) and f1 (-> Simulator.state_prime, and d2).
// ---------------------------------------------------------------------------
template <int NZ, bool MIS>
__device__ __forceinline__ void rk45_construct(const KParams& P, const RhsCtx<MIS>& C, const Rng& R, uint32_t stream,
                                               double x, double y, double& f0x, double& f0y, double& h_abs,
                                               double& spx, double& spy, bool need_f1 = true,
                                               SubStep* LS = nullptr, const uint32_t* wr = nullptr,
                                               bool in_init_box = false) {
    double n0x = 0.0, n0y = 0.0, n1x = 0.0, n1y = 0.0;
    bool have1 = true;
    // After a step (LS != nullptr) F0 and F1 live in the last attempt's block.  Nominal law: F0 = words 2,3 of
    // call 2, F1 = words 2,3 of call 3.  Mismatched law: F0 = draws 14..16 (call 3 words 2,3 + call 4 words 0,1),
    // F1 = draws 20..22 (call 5).  F1 (and with it the block's last call) is only evaluated when
    // Simulator.state_prime is wanted or when the bound below cannot certify h_abs == interval.
    auto eval_f1 = [&]() {
        if constexpr (NZ != kNoNoise) {
            if (LS == nullptr) {  // nominal reset constructor: F1 = draws 0,1 of its own block
                if constexpr (!MIS) {
                    uint32_t o[4];
                    float z2, z3;
                    philox_call(R, c0_of(stream, 0, 0), o);
                    box_muller<NZ>(o[0], o[1], z2, z3);
                    noise_vec<MIS>(P, C, 0.f, z2, z3, n1x, n1y);
                }
                return;
            }
            if (!LS->have3) {
                philox_call(R, c0_of(kStreamDyn, LS->last_attempt, nz_coll(NZ) ? 2u : (MIS ? 5u : 3u)), LS->w3);
                LS->have3 = true;
            }
            if constexpr (MIS && nz_coll(NZ)) {      // F1 = the pair of words 2,3 of call 2, through M
                float zx, zy;
                box_muller<NZ>(LS->w3[2], LS->w3[3], zx, zy);
                noise_vec<MIS>(P, C, derived_za<MIS>(C, zx, zy), zx, zy, n1x, n1y);
            } else if constexpr (MIS) {
                float za, zx, zy, zs;
                box_muller<NZ>(LS->w3[0], LS->w3[1], za, zx);
                box_muller<NZ>(LS->w3[2], LS->w3[3], zy, zs);
                noise_vec<MIS>(P, C, za, zx, zy, n1x, n1y);
            } else {
                float z2, z3;
                box_muller<NZ>(LS->w3[2], LS->w3[3], z2, z3);
                noise_vec<MIS>(P, C, 0.f, z2, z3, n1x, n1y);
            }
        }
    };
    if constexpr (NZ != kNoNoise) {
        if (LS != nullptr) {
            float za, zx, zy;
            f0_normals<NZ, MIS>(C, LS->f0a, LS->f0b, LS->f0y, za, zx, zy);
            noise_vec<MIS>(P, C, za, zx, zy, n0x, n0y);  // n0 itself only feeds the fp32 fallback test
            rhs_value<MIS>(P, C, za, zx, zy, f0x, f0y);
            have1 = need_f1;
            if (have1) eval_f1();
        }
    }
    if constexpr (NZ != kNoNoise) if (LS == nullptr) {  // reset constructor
        if constexpr (MIS) {  // sequential draws of its own block
            float z[8];
            block_normals<NZ, 2>(R, c0_of(stream, 0, 0), z);
            noise_vec<MIS>(P, C, z[0], z[1], z[2], n0x, n0y);
            rhs_value<MIS>(P, C, z[0], z[1], z[2], f0x, f0y);
            noise_vec<MIS>(P, C, z[3], z[4], z[5], n1x, n1y);
        } else {
            // F0 = the Box-Muller pair of words 2,3 of the reset call `wr` (its words 0,1 are the init position, sample_init):
            // an auto-reset costs one Philox call; F1 as lazily as after a step
            float z0, z1;
            box_muller<NZ>(wr[2], wr[3], z0, z1);
            noise_vec<MIS>(P, C, 0.f, z0, z1, n0x, n0y);
            rhs_value<MIS>(P, C, 0.f, z0, z1, f0x, f0y);
            have1 = need_f1;
            if (have1) eval_f1();
            // Position sampled from the init box, zero action, F1 not needed: with |f0| <= sigma Zmax and |y| confined to the
            // box, three of the four conditions of construct_level0 hold for every draw (checked once on the host,
            // make_kparams) and the fourth, F >= 2e-5 max(scale), is implied by F >= reset_fmin.  Same outcome
            // (h_abs = dt) as the full test whenever this passes; anything else goes on to the full test.
            if (!have1 && in_init_box && P.reset_fmin >= 0.0 &&
                ((__builtin_fabs(f0x) >= P.reset_fmin) | (__builtin_fabs(f0y) >= P.reset_fmin))) {  // max|f0| >= fmin, two compares
                spx = C.vx; spy = C.vy;
                h_abs = P.dt;
                return;
            }
        }
    }
    if constexpr (NZ == kNoNoise) { f0x = C.vx; f0y = C.vy; }
    spx = C.vx + n1x; spy = C.vy + n1y;

    const double sc0 = __builtin_fma(__builtin_fabs(x), P.rtol, P.atol);
    const double sc1 = __builtin_fma(__builtin_fabs(y), P.rtol, P.atol);
#if MRSIM_AB_CHEAP
    {
        double Gd;  // max |f1 - f0| / dt: exact when f1 was evaluated, else its worst case
        if (have1) Gd = fmax(__builtin_fabs(spx - f0x), __builtin_fabs(spy - f0y)) * P.inv_dt;
        else Gd = construct_gd_bound<MIS>(P, C);
        if (__builtin_expect(construct_level0<MIS>(P, x, y, sc0, sc1, f0x, f0y, Gd), 1)) { h_abs = P.dt; return; }
    }
#endif
    // Fast path (fp32, 5 % margins): decide "h_abs == interval_length" without divisions, square
    // roots or pow.  With r = 1/scale: d0^2 = D0/2, d1^2 = D1/2, (d2*h0)^2 = DD/2.
    //   100*h0 >= dt   <=  d0,d1 >= 1e-5  and  d0/d1 >= dt
    //   h1 >= dt       <=  d1 <= TH and d2 <= TH, h0 = min(0.01 d0/d1, dt),  TH = 0.01/dt^5
    // Any NaN/inf makes a comparison false and falls through to the exact path.
    {
        const float r0 = __builtin_amdgcn_rcpf((float)sc0), r1 = __builtin_amdgcn_rcpf((float)sc1);
        const float y0s = (float)x * r0, y1s = (float)y * r1;
        const float g0 = (float)f0x * r0, g1 = (float)f0y * r1;
        const float D0 = __builtin_fmaf(y0s, y0s, y1s * y1s);
        const float D1 = __builtin_fmaf(g0, g0, g1 * g1);
        const float TH2 = P.h1_thresh2_f;
        const bool fast01 = (D0 > 1e-9f) && (D1 > 1e-9f) && (D0 >= 1.05f * P.dt2_f * D1) && (D1 <= 1.9f * TH2);
        if (!have1) {
            // f1 not evaluated yet: worst case |n1 - n0| <= |n0| + amp*Zmax per axis (|z| <= 6.763), amp = sigma
            // (+ |g| in the mismatched law)
            const float zbx = (float)((MIS ? P.sigma + __builtin_fabs(C.gx) : P.sigma) * 6.78);
            const float zby = (float)((MIS ? P.sigma + __builtin_fabs(C.gy) : P.sigma) * 6.78);
            const float w0 = ((float)__builtin_fabs(n0x) + zbx) * r0, w1 = ((float)__builtin_fabs(n0y) + zby) * r1;
            const float DW = __builtin_fmaf(w0, w0, w1 * w1);
            if (__builtin_expect(fast01 && (DW <= 1.9f * TH2 * P.dt2_f) && (DW * D1 <= 1.9e-4f * TH2 * D0), 1)) {
                h_abs = P.dt;  // same outcome as with f1 evaluated: every admissible f1 passes the test
                return;
            }
            eval_f1();  // the bound cannot decide: evaluate F1 after all
            spx = C.vx + n1x; spy = C.vy + n1y;
        }
        const float e0 = (float)(n1x - n0x) * r0, e1 = (float)(n1y - n0y) * r1;
        const float DD = __builtin_fmaf(e0, e0, e1 * e1);
        const bool fast = fast01 && (DD <= 1.9f * TH2 * P.dt2_f) && (DD * D1 <= 1.9e-4f * TH2 * D0);
        if (__builtin_expect(fast, 1)) { h_abs = P.dt; return; }
    }
    // exact path: select_initial_step(fun, t0, y0, t_bound, inf, f0, +1, order=4, rtol, atol)
    const double d0 = rms2(x / sc0, y / sc1);
    const double d1 = rms2(f0x / sc0, f0y / sc1);
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    h0 = fmin(h0, P.dt);
    const double d2 = rms2((spx - f0x) / sc0, (spy - f0y) / sc1) / h0;
    double h1;
    if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
    else h1 = fifth_root(0.01 / fmax(d1, d2));
    h_abs = fmin(fmin(100 * h0, h1), P.dt);
}

// ---------------------------------------------------------------------------
// One RK45 env step: OdeSolver.step loop + RungeKutta._step_impl + rk_step, in time
// relative to the start of the env step (tau in [0, dt]).
// ---------------------------------------------------------------------------
constexpr double kB0 = 35.0 / 384, kE0 = -71.0 / 57600;
constexpr float kB2f = (float)(500.0 / 1113), kB3f = (float)(125.0 / 192), kB4f = (float)(-2187.0 / 6784),
                kB5f = (float)(11.0 / 84);
constexpr float kE2f = (float)(71.0 / 16695), kE3f = (float)(-71.0 / 1920), kE4f = (float)(17253.0 / 339200),
                kE5f = (float)(-22.0 / 525), kE6f = (float)(1.0 / 40);
// ---- collapsed law (nz_coll(NZ); oracle/mrsim_oracle.c: COL_*).  The stage normals N_2..N_5 of an rk_step attempt reach a result
// only through S_B = sum B_i N_i (position) and S_E = sum E_i N_i (error estimate); per noise component these are jointly
// Gaussian with the tableau's fixed covariance, so the law draws  S_B = cB z1,  S_E = cE1 z1 + cE2 z2  (z1, z2 iid N(0,1))
// directly.  K6, F0, F1 keep their own draws.  Equal in law to the per-stage noise of MR_simulator.py:73-83 for everything a
// step returns or carries (tests/test_noise_law_cpu.py), not draw for draw.  Philox layout of block DYN(attempt):
//   nominal:     call 0 = [policy words | z1],  call 1 = [F0 | z2],  call 2 = [K6 | F1]          (eager: calls 0, 1; 2 pairs)
//   mismatched:  call 0 = [policy words | u1],  call 1 = [F0 | u2'],  call 2 = [w | F1]         (eager: calls 0, 1; 3 pairs)
//                -- pairs through M (derived_za), f_new's noise folded into the error sum (kCE2Pf); w only when a sub-step follows
constexpr float kCBf = 0.8641431770614779f;     // sqrt(sum_{2..5} B_i^2)
constexpr float kCE1f = -0.05097452091652899f;  // sum_{2..5} B_i E_i / cB
constexpr float kCE2f = 0.05594888714408681f;   // sqrt(sum_{2..5} E_i^2 - cE1^2)
// mismatched model (oracle/mrsim_oracle.c: COL_CE2P): f_new's noise folded into the error sum, S_E' = S_E + E6 N_6 = M(cE1 u1 +
// cE2' u2'); when N_6 itself is needed (another sub-step follows) it comes from its conditional law given S_E':
// N_6 = M(kK6u u2' + kK6w w), w a fresh pair
constexpr float kCE2Pf = 0.06128032288313894f;                          // sqrt(cE2^2 + E6^2)
constexpr float kK6uf = (float)((1.0 / 40) / 0.06128032288313894);      // E6 / cE2'
constexpr float kK6wf = (float)(0.05594888714408681 / 0.06128032288313894);   // cE2 / cE2'
constexpr int kMaxAttempts = 4096;  // every lane leaves the loop: bounded spin
// |z| of this generator never exceeds sqrt(-2 ln 2^-33) = 6.763 (u >= 2^-33); bound with margin
constexpr double kZmaxE6 = 6.78 * (1.0 / 40);  // Zmax * E6
constexpr double kZmaxE6C = 6.78 * (1.0 / 40 + 0.05594888714408681);  // collapsed law: Zmax * (E6 + cE2), z2 bounded like K6

// weighted noise sums of one rk_step attempt:  nb = sum_{2..5} B_i N_i,  ne = sum_{2..6} E_i N_i,
// n6 = N_6 (f_new's noise; only matters when another sub-step follows)
struct AttemptNoise {
    float bx32, by32, ba32;  // sum_{2..5} B_i z_i per component (z_a: mismatched law only)
    double nex, ney;
    float z6a, z6x, z6y;  // f_new's normals; only needed when another sub-step follows
    // lazy K6 (nominal law, first attempt): error sums WITHOUT the E6*z6 term and the two Philox words of
    // K6's Box-Muller pair; nex/ney/z6* are filled by finish_k6() only if the bound test cannot decide
    float ex32, ey32, ea32;
    bool lazy6;
    // block layout (see oracle/mrsim_oracle.c).  nominal: call 2 = [K5 | F0], call 3 = [K6 | F1];
    // mismatched: call 3 = [K5x K5y | F0a F0x], call 4 = [F0y K6a | K6x K6y], call 5 = [F1a F1x | F1y -]
    uint32_t f0a, f0b;   // F0's first pair of words
    float f0y;           // mismatched: F0's third normal (its pair also yields K6a, which is needed eagerly)
    uint32_t k6a, k6b;   // mismatched: the words of the pair (K6x, K6y)
    uint32_t w3[4];      // the block's last call (3 nominal / 5 mismatched), valid when have3
    bool have3;
    // nominal law: the E-weighted sums (ex32, ey32) are formed lazily (finish_e) from the kept stage normals kz; the
    // level-0 accept test only needs R32 = Zmax E6 + sum_{2..5} |E_i| r_i >= |sum_{2..6} E_i z_i| (r_i = the Box-Muller
    // radius of stage i: |z| = r |cos| <= r)
    // collapsed law: kz[0..1] = z1 (x, y), (k6a, k6b) = the words of z2's pair (nominal; finish_e evaluates it),
    // R32 = Zmax (E6 + cE2) + |cE1| r1
    float kz[8], R32;
    bool lazyE;
    bool need6;   // mismatched model, collapsed law: f_new's normals (z6*) have not been formed yet (finish_k6 does, from kz[2..3] = u2' and w)
};

template <int NZ, bool MIS, bool FIRST>
__device__ __forceinline__ AttemptNoise attempt_noise(const KParams& P, const RhsCtx<MIS>& C, const Rng& R,
                                                      uint32_t attempt, const uint32_t* d0) {
    AttemptNoise A;
    if constexpr (NZ == kNoNoise) {
        A.bx32 = A.by32 = A.ba32 = 0.f; A.nex = A.ney = 0.0;
        A.z6a = A.z6x = A.z6y = 0.f;
        A.ex32 = A.ey32 = A.ea32 = 0.f; A.lazy6 = false;
        A.f0a = A.f0b = A.k6a = A.k6b = 0u; A.f0y = 0.f;
        A.w3[0] = A.w3[1] = A.w3[2] = A.w3[3] = 0u; A.have3 = false;
        A.R32 = 0.f; A.lazyE = false; A.need6 = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) A.kz[j] = 0.f;
        return A;
    } else if constexpr (nz_coll(NZ) && !MIS) {
        // collapsed nominal law: call 0 = [policy | z1], call 1 = [F0 | z2]; call 2 = [K6 | F1] is only fetched if K6 or F1 is
        // ever needed.  TWO Philox calls and two Box-Muller pairs (z1 here, F0 in the constructor) per step in the common case.
        uint32_t wl[3][4];
        const uint32_t (*w)[4];
        if constexpr (FIRST) {
            w = reinterpret_cast<const uint32_t (*)[4]>(d0);
            A.have3 = false;
            A.w3[0] = A.w3[1] = A.w3[2] = A.w3[3] = 0u;
        } else {
            const uint32_t c0s[3] = {c0_of(kStreamDyn, attempt, 0), c0_of(kStreamDyn, attempt, 1), c0_of(kStreamDyn, attempt, 2)};
            philox_multi<3>(R, c0s, wl);
            w = wl;
            A.have3 = true;
            A.w3[0] = wl[2][0]; A.w3[1] = wl[2][1]; A.w3[2] = wl[2][2]; A.w3[3] = wl[2][3];
        }
        A.f0a = w[1][0]; A.f0b = w[1][1];
        A.k6a = w[1][2]; A.k6b = w[1][3];   // z2's pair, evaluated by finish_e
        A.f0y = 0.f; A.ea32 = 0.f;
        float z1x, z1y, r1;
        box_muller<NZ>(w[0][2], w[0][3], z1x, z1y, &r1);
        A.bx32 = kCBf * z1x; A.by32 = kCBf * z1y; A.ba32 = 0.f;
        // |S_E / sigma + E6 z6| <= |cE1| r1 + cE2 Zmax + E6 Zmax  (|z1| <= r1: the pair's radius)
        A.R32 = __builtin_fmaf(-kCE1f, r1, (float)kZmaxE6C) * 1.0001f;
#pragma unroll
        for (int j = 0; j < 8; ++j) A.kz[j] = 0.f;
        A.kz[0] = z1x; A.kz[1] = z1y;
        A.lazyE = true;
        A.ex32 = A.ey32 = 0.f;
        A.nex = A.ney = 0.0; A.z6a = A.z6x = A.z6y = 0.f;
        A.lazy6 = true; A.need6 = false;
        return A;
    } else if constexpr (nz_coll(NZ)) {
        // collapsed law, mismatched model: call 0 = [policy | u1], call 1 = [F0 | u2'] eager -- TWO Philox calls and three
        // Box-Muller pairs (u1, u2' here, F0 in the constructor) per step in the common case; call 2 = [w | F1] only when f_new's
        // noise or F1 is ever needed.  The error sums are formed eagerly (they include f_new's E6 N_6: nothing about the accept
        // decision is left to a bound): under this model error_norm sits near 0.5 and a one-sided test would fail half the time.
        uint32_t wl[3][4];
        const uint32_t (*w)[4];
        if constexpr (FIRST) {
            w = reinterpret_cast<const uint32_t (*)[4]>(d0);
            A.have3 = false;
            A.w3[0] = A.w3[1] = A.w3[2] = A.w3[3] = 0u;
        } else {
            const uint32_t c0s[3] = {c0_of(kStreamDyn, attempt, 0), c0_of(kStreamDyn, attempt, 1), c0_of(kStreamDyn, attempt, 2)};
            philox_multi<3>(R, c0s, wl);
            w = wl;
            A.have3 = true;
            A.w3[0] = wl[2][0]; A.w3[1] = wl[2][1]; A.w3[2] = wl[2][2]; A.w3[3] = wl[2][3];
        }
        A.f0a = w[1][0]; A.f0b = w[1][1]; A.k6a = A.k6b = 0u; A.f0y = 0.f;
        float u1x, u1y, u2x, u2y;
        box_muller<NZ>(w[0][2], w[0][3], u1x, u1y);
        box_muller<NZ>(w[1][2], w[1][3], u2x, u2y);
        A.bx32 = kCBf * u1x; A.by32 = kCBf * u1y;
        A.ba32 = derived_za<MIS>(C, A.bx32, A.by32);                  // M is linear: the a-term of cB u1
        A.ex32 = __builtin_fmaf(kCE2Pf, u2x, kCE1f * u1x);
        A.ey32 = __builtin_fmaf(kCE2Pf, u2y, kCE1f * u1y);
        A.ea32 = derived_za<MIS>(C, A.ex32, A.ey32);
        // the exact error-noise terms (f_new included): rk45_attempt's decision needs nothing more
        A.nex = __builtin_fma(C.gx, (double)A.ea32, P.sigma * (double)A.ex32);
        A.ney = __builtin_fma(C.gy, (double)A.ea32, P.sigma * (double)A.ey32);
        A.z6a = A.z6x = A.z6y = 0.f;
        A.lazy6 = false; A.need6 = true;
        A.R32 = 0.f; A.lazyE = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) A.kz[j] = 0.f;
        A.kz[2] = u2x; A.kz[3] = u2y;
        return A;
    } else if constexpr (!MIS) {
        // nominal law: call 0 = [K1 (dead) | K2], call 1 = [K3 | K4], call 2 = [K5 | F0], call 3 = [K6 | F1].
        // First attempt: calls 0..2 were drawn up front (d0) and call 3 is fetched only if K6 or F1 is ever
        // needed (finish_k6 / rk45_construct).  Later attempts (rare) draw all four calls here.
        uint32_t wl[4][4];
        const uint32_t (*w)[4];
        if constexpr (FIRST) {
            w = reinterpret_cast<const uint32_t (*)[4]>(d0);
            A.have3 = false;
            A.w3[0] = A.w3[1] = A.w3[2] = A.w3[3] = 0u;
        } else {
            const uint32_t c0s[4] = {c0_of(kStreamDyn, attempt, 0), c0_of(kStreamDyn, attempt, 1),
                                     c0_of(kStreamDyn, attempt, 2), c0_of(kStreamDyn, attempt, 3)};
            philox_multi<4>(R, c0s, wl);
            w = wl;
            A.have3 = true;
            A.w3[0] = wl[3][0]; A.w3[1] = wl[3][1]; A.w3[2] = wl[3][2]; A.w3[3] = wl[3][3];
        }
        A.f0a = w[2][2]; A.f0b = w[2][3];
        A.k6a = A.k6b = 0u; A.f0y = 0.f; A.ea32 = 0.f;
        float k2x, k2y, k3x, k3y, k4x, k4y, k5x, k5y, r2, r3, r4, r5;
        box_muller<NZ>(w[0][2], w[0][3], k2x, k2y, &r2);
        box_muller<NZ>(w[1][0], w[1][1], k3x, k3y, &r3);
        box_muller<NZ>(w[1][2], w[1][3], k4x, k4y, &r4);
        box_muller<NZ>(w[2][0], w[2][1], k5x, k5y, &r5);
        float bx = kB2f * k2x, by = kB2f * k2y;
        bx = __builtin_fmaf(kB3f, k3x, bx); by = __builtin_fmaf(kB3f, k3y, by);
        bx = __builtin_fmaf(kB4f, k4x, bx); by = __builtin_fmaf(kB4f, k4y, by);
        bx = __builtin_fmaf(kB5f, k5x, bx); by = __builtin_fmaf(kB5f, k5y, by);
        float Rb = __builtin_fmaf(kE2f, r2, (float)kZmaxE6);   // E2, E4, E6 > 0 > E3, E5
        Rb = __builtin_fmaf(-kE3f, r3, Rb);
        Rb = __builtin_fmaf(kE4f, r4, Rb);
        Rb = __builtin_fmaf(-kE5f, r5, Rb);
        A.R32 = Rb * 1.0001f;  // fp32 rounding of the four terms
        A.kz[0] = k2x; A.kz[1] = k2y; A.kz[2] = k3x; A.kz[3] = k3y; A.kz[4] = k4x; A.kz[5] = k4y; A.kz[6] = k5x; A.kz[7] = k5y;
        A.lazyE = true;
        A.bx32 = bx; A.by32 = by; A.ba32 = 0.f;
        A.ex32 = A.ey32 = 0.f;
        A.nex = A.ney = 0.0; A.z6a = A.z6x = A.z6y = 0.f;
        A.lazy6 = true; A.need6 = false;
        return A;
    } else {
        // mismatched law, draws (z_a, z_x, z_y) per evaluation, draw index = 4*call + lane:
        //   0..1 K1 (dead; policy words)  2..4 K2  5..7 K3  8..10 K4  11..13 K5  14..16 F0  17..19 K6  20..22 F1
        // Calls 0..4 are drawn here (first attempt: up front, d0).  The pair (F0y, K6a) is evaluated eagerly, so
        // K6's g*z_a term is exact in the error sum and only sigma*(K6x, K6y) is left to the lazy bound (as in
        // the nominal law); K6's (x, y) pair is only Box-Muller'd by finish_k6 and call 5 only fetched by
        // rk45_construct when F1 is needed.
        uint32_t wl[5][4];
        const uint32_t (*w)[4];
        if constexpr (FIRST) {
            w = reinterpret_cast<const uint32_t (*)[4]>(d0);
        } else {
            const uint32_t c0s[5] = {c0_of(kStreamDyn, attempt, 0), c0_of(kStreamDyn, attempt, 1),
                                     c0_of(kStreamDyn, attempt, 2), c0_of(kStreamDyn, attempt, 3),
                                     c0_of(kStreamDyn, attempt, 4)};
            philox_multi<5>(R, c0s, wl);
            w = wl;
        }
        A.have3 = false;
        A.w3[0] = A.w3[1] = A.w3[2] = A.w3[3] = 0u;
        A.f0a = w[3][2]; A.f0b = w[3][3]; A.k6a = w[4][2]; A.k6b = w[4][3];
        box_muller<NZ>(w[4][0], w[4][1], A.f0y, A.z6a);
        float z[12];  // draws 2..13
        box_muller<NZ>(w[0][2], w[0][3], z[0], z[1]);
        box_muller<NZ>(w[1][0], w[1][1], z[2], z[3]);
        box_muller<NZ>(w[1][2], w[1][3], z[4], z[5]);
        box_muller<NZ>(w[2][0], w[2][1], z[6], z[7]);
        box_muller<NZ>(w[2][2], w[2][3], z[8], z[9]);
        box_muller<NZ>(w[3][0], w[3][1], z[10], z[11]);
        const float *k2 = z, *k3 = z + 3, *k4 = z + 6, *k5 = z + 9;
        float ba = kB2f * k2[0], bx = kB2f * k2[1], by = kB2f * k2[2];
        ba = __builtin_fmaf(kB3f, k3[0], ba); bx = __builtin_fmaf(kB3f, k3[1], bx); by = __builtin_fmaf(kB3f, k3[2], by);
        ba = __builtin_fmaf(kB4f, k4[0], ba); bx = __builtin_fmaf(kB4f, k4[1], bx); by = __builtin_fmaf(kB4f, k4[2], by);
        ba = __builtin_fmaf(kB5f, k5[0], ba); bx = __builtin_fmaf(kB5f, k5[1], bx); by = __builtin_fmaf(kB5f, k5[2], by);
        float ea = kE2f * k2[0], ex = kE2f * k2[1], ey = kE2f * k2[2];
        ea = __builtin_fmaf(kE3f, k3[0], ea); ex = __builtin_fmaf(kE3f, k3[1], ex); ey = __builtin_fmaf(kE3f, k3[2], ey);
        ea = __builtin_fmaf(kE4f, k4[0], ea); ex = __builtin_fmaf(kE4f, k4[1], ex); ey = __builtin_fmaf(kE4f, k4[2], ey);
        ea = __builtin_fmaf(kE5f, k5[0], ea); ex = __builtin_fmaf(kE5f, k5[1], ex); ey = __builtin_fmaf(kE5f, k5[2], ey);
        ea = __builtin_fmaf(kE6f, A.z6a, ea);
        A.bx32 = bx; A.by32 = by; A.ba32 = ba;
        A.ex32 = ex; A.ey32 = ey; A.ea32 = ea;
        A.nex = A.ney = 0.0; A.z6x = A.z6y = 0.f;
        A.lazy6 = true; A.need6 = false;
        A.R32 = 0.f; A.lazyE = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) A.kz[j] = 0.f;
        return A;
    }
}


// the E-weighted sums of stages 2..5 from the kept normals (same fp32 chain as the eager form had)
template <int NZ>
__device__ __forceinline__ void finish_e(AttemptNoise& A) {
    if (!A.lazyE) return;
    if constexpr (nz_coll(NZ)) {  // S_E = cE1 z1 + cE2 z2: evaluate z2's pair
        float z2x, z2y;
        box_muller<NZ>(A.k6a, A.k6b, z2x, z2y);
        A.ex32 = __builtin_fmaf(kCE2f, z2x, kCE1f * A.kz[0]);
        A.ey32 = __builtin_fmaf(kCE2f, z2y, kCE1f * A.kz[1]);
        A.lazyE = false;
        return;
    }
    float ex = kE2f * A.kz[0], ey = kE2f * A.kz[1];
    ex = __builtin_fmaf(kE3f, A.kz[2], ex); ey = __builtin_fmaf(kE3f, A.kz[3], ey);
    ex = __builtin_fmaf(kE4f, A.kz[4], ex); ey = __builtin_fmaf(kE4f, A.kz[5], ey);
    ex = __builtin_fmaf(kE5f, A.kz[6], ex); ey = __builtin_fmaf(kE5f, A.kz[7], ey);
    A.ex32 = ex; A.ey32 = ey;
    A.lazyE = false;
}

// evaluate K6's Box-Muller pair and complete the error sums (same fp32 chain as the eager form: E6 is its last term)
template <int NZ, bool MIS>
__device__ __forceinline__ void finish_k6(const KParams& P, const RhsCtx<MIS>& C, const Rng& R, uint32_t attempt,
                                          AttemptNoise& A) {
    if constexpr (MIS && nz_coll(NZ)) {
        // f_new's noise given the error sum it is already part of: N_6 = M(kK6u u2' + kK6w w), w = the pair of words 0,1 of call 2.
        // Only a following sub-step sees it (as its K0); the error sums stay as they are.
        if (!A.have3) { philox_call(R, c0_of(kStreamDyn, attempt, 2u), A.w3); A.have3 = true; }
        float wx, wy;
        box_muller<NZ>(A.w3[0], A.w3[1], wx, wy);
        A.z6x = __builtin_fmaf(kK6wf, wx, kK6uf * A.kz[2]);
        A.z6y = __builtin_fmaf(kK6wf, wy, kK6uf * A.kz[3]);
        A.z6a = derived_za<MIS>(C, A.z6x, A.z6y);
        A.need6 = false;
        return;
    } else if constexpr (MIS) {
        box_muller<NZ>(A.k6a, A.k6b, A.z6x, A.z6y);
        A.nex = __builtin_fma(C.gx, (double)A.ea32, P.sigma * (double)__builtin_fmaf(kE6f, A.z6x, A.ex32));
        A.ney = __builtin_fma(C.gy, (double)A.ea32, P.sigma * (double)__builtin_fmaf(kE6f, A.z6y, A.ey32));
    } else {
        if (!A.have3) { philox_call(R, c0_of(kStreamDyn, attempt, nz_coll(NZ) ? 2u : 3u), A.w3); A.have3 = true; }
        box_muller<NZ>(A.w3[0], A.w3[1], A.z6x, A.z6y);
        A.nex = P.sigma * (double)__builtin_fmaf(kE6f, A.z6x, A.ex32);
        A.ney = P.sigma * (double)__builtin_fmaf(kE6f, A.z6y, A.ey32);
    }
    A.lazy6 = false;
}


// ---- one-sided accept tests on the LAST sub-step of an env step (K6 = f_new only enters the error estimate then; its
// contribution is bounded by |z| <= Zmax).  Both return true only if error_norm < 1 is certain; false decides nothing.
// They work on scale_lo = atol + rtol |y| <= the real scale atol + rtol max(|y|, |y_new|): a smaller scale only makes the
// tests harder to pass.  Shared by the general attempt (rk45_attempt) and the straight-line fast step (rk45_fast_step).
//
// level 0 (max norm, nominal law): |err| <= h (|E0| max|K0 - V| + sigma (Zmax E6 + sum |E_i| r_i)) per component
// (R32, attempt_noise); below 0.99 x the smaller scale, each ratio of the rms norm is < 1.  The E-weighted sums of the
// stage normals are not even formed.  Not used under the mismatched law: its velocities (a0' = a0 + 0.2 f: up to 100)
// and the extra g*z_a noise put error_norm near 0.5, where this test fails for half the waves and only adds work.
__device__ __forceinline__ bool accept_level0(const KParams& P, const AttemptNoise& A, double dfx, double dfy, double h,
                                              double l0, double l1) {
    const double dmax = max_abs2(dfx, dfy);
    const double eb = __builtin_fma(P.sigma, (double)A.R32, -kE0 * dmax);
    return h * eb <= 0.99 * min2(l0, l1);
}
// level 1 (rms norm with the exact stage sums ex32 / ey32 / ea32 and the worst-case K6): sqrt(2) less pessimistic
template <bool MIS, int NZ>
__device__ __forceinline__ bool accept_level1(const KParams& P, const RhsCtx<MIS>& C, const AttemptNoise& A, double dfx,
                                              double dfy, double h, double l0, double l1) {
    double pex = __builtin_fma(P.sigma, (double)A.ex32, kE0 * dfx);
    double pey = __builtin_fma(P.sigma, (double)A.ey32, kE0 * dfy);
    if constexpr (MIS) {  // the g*z_a terms (K6a included) are already exact in ea32
        pex = __builtin_fma(C.gx, (double)A.ea32, pex);
        pey = __builtin_fma(C.gy, (double)A.ea32, pey);
    }
    const double l00 = l0 * l0, l11 = l1 * l1;
    if constexpr (MIS && nz_coll(NZ)) {
        // mismatched model under the collapsed law: f_new's term is part of the sums -- nothing is left to bound, and the squares
        // need no magnitudes
        const double ax = h * pex, ay = h * pey;
        return __builtin_fma(ax * ax, l11, (ay * ay) * l00) < 1.96 * (l00 * l11);
    }
    const double b6 = h * P.zmax_e6_sigma;
    const double axw = __builtin_fabs(h * pex) + b6;
    const double ayw = __builtin_fabs(h * pey) + b6;
    return __builtin_fma(axw * axw, l11, (ayw * ayw) * l00) < 1.96 * (l00 * l11);
}

// one rk_step attempt + the accept / reject decision of _step_impl.  Returns true when the
// attempt was accepted (state advanced to tau = tn).
template <int NZ, bool MIS, bool FIRST>
__device__ __forceinline__ bool rk45_attempt(const KParams& P, const RhsCtx<MIS>& C, const Rng& R, SubStep& S,
                                             double& x, double& y, double& f0x, double& f0y, int& fail,
                                             const uint32_t* d0) {
    double tn = S.tau + S.h_abs;
    if (tn > P.dt) tn = P.dt;
    const double h = tn - S.tau;
    S.h_abs = h;
    AttemptNoise A = attempt_noise<NZ, MIS, FIRST>(P, C, R, S.attempt, d0);
    S.attempt += 1;
    const double dfx = f0x - C.vx, dfy = f0y - C.vy;
    double sx = __builtin_fma(kB0, dfx, C.vx), sy = __builtin_fma(kB0, dfy, C.vy);
    if constexpr (NZ != kNoNoise) {  // + sum_{2..5} B_i N_i, one fma per term (N = sigma z [+ g z_a])
        sx = __builtin_fma(P.sigma, (double)A.bx32, sx); sy = __builtin_fma(P.sigma, (double)A.by32, sy);
        if constexpr (MIS) { sx = __builtin_fma(C.gx, (double)A.ba32, sx); sy = __builtin_fma(C.gy, (double)A.ba32, sy); }
    }
    const double xn = __builtin_fma(h, sx, x);
    const double yn = __builtin_fma(h, sy, y);
    const bool last = !(tn < P.dt);
    bool accepted = false, decided = false;
    if constexpr (NZ != kNoNoise) {
        if (A.lazy6) {
            // K6 (= f_new) only enters the error estimate (weight E6) and, when another sub-step follows, the
            // next K0.  On the last sub-step, bound its contribution by |z| <= Zmax: if even the worst case
            // passes the accept test, the outcome is the eager one and K6's Box-Muller pair is never evaluated.
            // The bound uses scale_lo = atol + rtol*|y| <= the real scale atol + rtol*max(|y|, |y_new|): a smaller
            // scale only makes the test harder to pass, so it stays one-sided and spares four v_max_f64.
            if (last) {
                const double l0 = __builtin_fma(__builtin_fabs(x), P.rtol, P.atol);
                const double l1 = __builtin_fma(__builtin_fabs(y), P.rtol, P.atol);
                if constexpr (!MIS && MRSIM_AB_CHEAP) {
                    if (__builtin_expect(accept_level0(P, A, dfx, dfy, h, l0, l1), 1)) { accepted = true; decided = true; }
                }
                if (!decided) {
                    finish_e<NZ>(A);
                    if (accept_level1<MIS, NZ>(P, C, A, dfx, dfy, h, l0, l1)) { accepted = true; decided = true; }
                }
            }
            if (!decided) finish_e<NZ>(A);
            if (!decided) finish_k6<NZ, MIS>(P, C, R, S.attempt - 1, A);
        }
    }
    if (!decided) {
        const double sc0 = __builtin_fma(fmax(__builtin_fabs(x), __builtin_fabs(xn)), P.rtol, P.atol);
        const double sc1 = __builtin_fma(fmax(__builtin_fabs(y), __builtin_fabs(yn)), P.rtol, P.atol);
        const double s00 = sc0 * sc0, s11 = sc1 * sc1;
        const double lim = 2.0 * s00 * s11;
        double ex = kE0 * dfx, ey = kE0 * dfy;
        if constexpr (NZ != kNoNoise) { ex += A.nex; ey += A.ney; }
        ex *= h; ey *= h;
        // fast accept (no division / sqrt / pow): error_norm^2 = q / lim, 2 % margin; the step-size
        // factor is only needed when another sub-step follows.
        const double q = __builtin_fma(ex * ex, s11, (ey * ey) * s00);
        if (__builtin_expect(last && q < 0.98 * lim, 1)) {
            accepted = true;
        } else {
            const double error_norm = rms2(ex / sc0, ey / sc1);
            if (error_norm < 1.0) {
                double factor = (error_norm == 0.0) ? 10.0 : fmin(10.0, 0.9 * inv_fifth_root(error_norm));
                if (S.rejected) factor = fmin(1.0, factor);
                S.h_abs *= factor;
                accepted = true;
            } else {
                S.h_abs *= fmax(0.2, 0.9 * inv_fifth_root(error_norm));
                S.rejected = true;
                accepted = false;
            }
        }
    }
    if (__builtin_expect(!accepted && S.attempt >= (uint32_t)kMaxAttempts, 0)) {
        // The reference would fail with TOO_SMALL_STEP / spin on NaN input: flag it and finish.
        fail |= 1;
        accepted = true;
        tn = P.dt;
    }
    if (accepted) {
        S.tau = tn; x = xn; y = yn;
        if constexpr (NZ != kNoNoise) {
            S.last_attempt = S.attempt - 1; S.f0a = A.f0a; S.f0b = A.f0b; S.f0y = A.f0y;
            S.have3 = A.have3;
            S.w3[0] = A.w3[0]; S.w3[1] = A.w3[1]; S.w3[2] = A.w3[2]; S.w3[3] = A.w3[3];
        }
        if (!last) {  // f = f_new = K[6]; after the last sub-step the constructor replaces f anyway
            double n6x = 0.0, n6y = 0.0;
            if constexpr (NZ != kNoNoise && MIS && nz_coll(NZ)) { if (A.need6) finish_k6<NZ, MIS>(P, C, R, S.attempt - 1, A); }
            if constexpr (NZ != kNoNoise) noise_vec<MIS>(P, C, A.z6a, A.z6x, A.z6y, n6x, n6y);
            f0x = C.vx + n6x; f0y = C.vy + n6y;
        }
        S.rejected = false;
    }
    return accepted;
}

template <int NZ, bool MIS>
__device__ __forceinline__ SubStep rk45_integrate(const KParams& P, const RhsCtx<MIS>& C, const Rng& R, double& x,
                                                  double& y, double f0x, double f0y, double h_abs, int& fail,
                                                  const uint32_t* d0) {
    SubStep S;
    S.tau = 0.0; S.h_abs = h_abs; S.attempt = 0u; S.rejected = false;
    S.last_attempt = 0u; S.f0a = S.f0b = 0u; S.f0y = 0.f; S.w3[0] = S.w3[1] = S.w3[2] = S.w3[3] = 0u; S.have3 = false;
    // first attempt peeled: its RNG counters are wave-uniform (attempt = 0), and in the common
    // regime (|y| >~ 1) it is the only one
    rk45_attempt<NZ, MIS, true>(P, C, R, S, x, y, f0x, f0y, fail, d0);
    while (__builtin_expect(S.tau < P.dt, 0)) {
        rk45_attempt<NZ, MIS, false>(P, C, R, S, x, y, f0x, f0y, fail, nullptr);
        if (S.attempt >= (uint32_t)kMaxAttempts) { fail |= 1; break; }
    }
    return S;
}

// ---------------------------------------------------------------------------
// per-env registers
// ---------------------------------------------------------------------------
struct EnvRegs {
    double x, y;        // integrator.y
    double f0x, f0y;    // integrator.f (K[0] of the next step), stored fp32
    double h_abs;       // integrator.h_abs, stored as fp32 ratio h_abs/dt
    int32_t counter;    // MR_Env.counter
    float ep_ret;
};

// ---------------------------------------------------------------------------
// The common env step as ONE straight-line block.  Away from the coordinate axes practically every step is: carried
// h_abs >= dt, ONE rk_step attempt over the whole interval, accepted by the first-level test, and a constructor whose
// first-level test certifies h_abs == dt for the next step.  rk45_fast_step evaluates exactly that -- the same helper
// functions and the same arithmetic the general path (rk45_integrate + rk45_construct) uses -- with no branch in it, so
// the scheduler can interleave the Philox tail, the table sin/cos, five Box-Muller pairs and the fp64 chains of the
// tests (only four waves share a SIMD at N = 262 144: in-wave ILP is what hides the fp64 / transcendental latencies; every
// branch of the general path ends a scheduling region).  Returns true and commits the new state if all three conditions
// hold; otherwise returns false with the env untouched and the caller runs the general path, which reaches the same bits
// whenever the fast step would have (tested: MRSIM_FAST_STEP = 0 builds, tools/ab_rollout.py reports max |pos| difference 0).
// Needs nothing of F1 (Simulator.state_prime): callers that want it use the general path.
// ---------------------------------------------------------------------------
#ifndef MRSIM_LM1   // A/B switch: 0 = no level -1 test, the fast step always evaluates the fp64 first-level tests
#define MRSIM_LM1 1
#endif
// What a fused rollout carries in fp32 beside the env state so that the fast step can certify its two first-level tests
// (accept_level0, construct_level0: ~25 fp64-rate instructions) from BOUNDS instead: |x|, |y| as the last observation holds them,
// an upper bound kb of max|K0|, and |f| of this step's action.
struct Lm1 {
    float m, kb, fa;   // min(|x|, |y|), the bound of max|K0|, |f|
    uint32_t xy_or;    // bits(x) | bits(y) of the float32 coordinates: as a float, its magnitude is >= max(|x|, |y|)
};
__device__ __forceinline__ void lm1_position(Lm1& lm, float ox, float oy) {
    lm.m = fminf(__builtin_fabsf(ox), __builtin_fabsf(oy));
    lm.xy_or = __float_as_uint(ox) | __float_as_uint(oy);
}
__device__ __forceinline__ float lm1_bound(double f0x, double f0y) {
    return fmaxf(__builtin_fabsf((float)f0x), __builtin_fabsf((float)f0y)) * 1.000001f;
}

template <int NZ, bool MIS>
__device__ __forceinline__ bool rk45_fast_step(const KParams& P, const RhsCtx<MIS>& C, const Rng& R, const uint32_t* d0,
                                               EnvRegs& e, Lm1* lm, int& fail) {
    static_assert(NZ != kNoNoise, "the sigma = 0 kernels have nothing to hide: general path");
    AttemptNoise A = attempt_noise<NZ, MIS, true>(P, C, R, 0u, d0);
    const double h = P.dt;
    // rk_step, first (and last) attempt: same expressions as rk45_attempt
    const double dfx = e.f0x - C.vx, dfy = e.f0y - C.vy;
    double sx = __builtin_fma(kB0, dfx, C.vx), sy = __builtin_fma(kB0, dfy, C.vy);
    sx = __builtin_fma(P.sigma, (double)A.bx32, sx); sy = __builtin_fma(P.sigma, (double)A.by32, sy);
    if constexpr (MIS) { sx = __builtin_fma(C.gx, (double)A.ba32, sx); sy = __builtin_fma(C.gy, (double)A.ba32, sy); }
    const double xn = __builtin_fma(h, sx, e.x);
    const double yn = __builtin_fma(h, sy, e.y);
    // RungeKutta.__init__ of the next step: f0 = simulate() with the F0 draws of this attempt's block
    float za, zx, zy, rF0 = 0.f;
    f0_normals<NZ, MIS>(C, A.f0a, A.f0b, A.f0y, za, zx, zy, &rF0);
    double f0x, f0y;
    rhs_value<MIS>(P, C, za, zx, zy, f0x, f0y);
    // ---- level -1 (nominal law, fused rollout): the first-level tests below hold for EVERY admissible value of the quantities
    // they read when these fp32 inequalities hold (each bound errs to the safe side; margins 1e-5 .. 1e-3 against fp32 rounding):
    //   dmax = max|K0 - V| <= Dhi = kb + |a0 f|;   F = max|f0| <= Fhi = |a0 f| + sigma r_F0;   |step| <= dlt = dt (|V| + B0 Dhi + 10.6 sigma)
    //   (sum |B_i| r_i <= 1.5536 Zmax and cB r_1 <= 0.8642 Zmax are both below 10.6);  m = min(|x|, |y|) -> mn = m - dlt after the step
    //   accept_level0:     dt (sigma R32 + |E0| Dhi) <= 0.99 (atol + rtol m)
    //   construct_level0:  mn >= k_h0 Fhi;  h1_thresh_m (atol + rtol mn) >= max(Fhi, Gd);  max|f0| >= 2e-5 (atol + rtol 16385) with max(|x|, |y|) + dlt <= 16384
    //                      (max(|x|, |y|) < 8192 and dlt <= m, which mn >= 0 implies)
    // The host folds the constants and their margins (make_kparams: lm_*): dlt = lm_da |f| + lm_dk kb + lm_dc; the two construct
    // conditions on mn become mn >= max(lm_kh Fhi, lm_mg) with lm_kh = max(k_h0, 1 / (h1_thresh_m rtol)), lm_mg = (Gd / h1_thresh_m - atol) / rtol.
    // When every lane of the wave passes, the wave skips the fp64 tests (same outcome: commit); otherwise all lanes evaluate them.
    float Fhi = 0.f;
    if constexpr (!MIS && MRSIM_LM1 != 0) {
        if (lm != nullptr) {
            const float Vhi = P.lm_a0 * lm->fa;
            const float Dhi = lm->kb + Vhi;
            Fhi = __builtin_fmaf(P.lm_sigma, rF0, Vhi);
            const float dlt = __builtin_fmaf(P.lm_da, lm->fa, __builtin_fmaf(P.lm_dk, lm->kb, P.lm_dc));
            const float mn = __builtin_fmaf(lm->m, 0.99999f, -dlt);
            const bool cb = __builtin_fmaf(P.lm_es, A.R32, P.lm_ed * Dhi) <= __builtin_fmaf(P.lm_rt, lm->m, P.lm_at);
            const bool c12 = mn >= fmaxf(P.lm_kh * Fhi, P.lm_mg);
            // max(|x|, |y|) < 8192 (the OR of the two bit patterns has the larger exponent or more) and, from c12, dlt <= m: |xn|, |yn| < 16384
            const bool c3 = (__builtin_fabsf(__uint_as_float(lm->xy_or)) < 8192.0f) & ((__builtin_fabs(f0x) >= P.lm_ccap) | (__builtin_fabs(f0y) >= P.lm_ccap));
            const bool lm1 = cb & c12 & c3 & (e.h_abs >= P.dt);
#ifdef MRSIM_VERIFY_LM1   // verification build (tests): the bounds must never certify a step the fp64 tests would refuse
            {
                const double l0v = __builtin_fma(__builtin_fabs(e.x), P.rtol, P.atol), l1v = __builtin_fma(__builtin_fabs(e.y), P.rtol, P.atol);
                const double s0v = __builtin_fma(__builtin_fabs(xn), P.rtol, P.atol), s1v = __builtin_fma(__builtin_fabs(yn), P.rtol, P.atol);
                const bool full = (e.h_abs >= P.dt) && accept_level0(P, A, dfx, dfy, h, l0v, l1v) &&
                                  construct_level0<MIS>(P, xn, yn, s0v, s1v, f0x, f0y, construct_gd_bound<MIS>(P, C));
                if (lm1 && !full) fail |= 2;
            }
#endif
            if (__all(lm1)) {
                e.x = xn; e.y = yn; e.f0x = f0x; e.f0y = f0y; e.h_abs = P.dt;
                lm->kb = Fhi;
                return true;
            }
        }
    }
    const double l0 = __builtin_fma(__builtin_fabs(e.x), P.rtol, P.atol);
    const double l1 = __builtin_fma(__builtin_fabs(e.y), P.rtol, P.atol);
    bool acc;
    if constexpr (!MIS) {
        acc = accept_level0(P, A, dfx, dfy, h, l0, l1);
    } else if constexpr (nz_coll(NZ)) {
        // collapsed law: the error sums are complete (f_new folded in), the test is the exact one on scale_lo, and with h = dt the
        // step size moves to the right-hand side as a host-folded constant:  pex^2 l1^2 + pey^2 l0^2 < (1.96 / dt^2) l0^2 l1^2
        const double pex = __builtin_fma(C.gx, (double)A.ea32, __builtin_fma(P.sigma, (double)A.ex32, kE0 * dfx));
        const double pey = __builtin_fma(C.gy, (double)A.ea32, __builtin_fma(P.sigma, (double)A.ey32, kE0 * dfy));
        const double p = pex * l1, q = pey * l0, r = l0 * l1;
        acc = __builtin_fma(p, p, q * q) < P.acc_lim_dt2 * (r * r);
    } else {
        acc = accept_level1<MIS, NZ>(P, C, A, dfx, dfy, h, l0, l1);  // the mismatched attempt forms its E sums eagerly
    }
    const double sc0 = __builtin_fma(__builtin_fabs(xn), P.rtol, P.atol);
    const double sc1 = __builtin_fma(__builtin_fabs(yn), P.rtol, P.atol);
    const bool pass = construct_level0<MIS>(P, xn, yn, sc0, sc1, f0x, f0y, construct_gd_bound<MIS>(P, C));
    const bool ok = (e.h_abs >= P.dt) && acc && pass;
    if (ok) {
        e.x = xn; e.y = yn; e.f0x = f0x; e.f0y = f0y; e.h_abs = P.dt;
        if constexpr (!MIS && MRSIM_LM1 != 0) { if (lm != nullptr) lm->kb = Fhi; }
    }
    return ok;
}

// Build extension (BASELINE configs 2/3): fixed-step Euler / classical RK4, noise added to the
// derivative at every RHS evaluation exactly as `simulate` does.  Mirrors the oracle's loop.
template <int NZ, bool MIS>
__device__ __forceinline__ void fixed_integrate(const KParams& P, const RhsCtx<MIS>& C, const Rng& R, double& x,
                                                double& y, double& spx, double& spy) {
    const int S = P.substeps > 0 ? P.substeps : 1;
    const double h = P.dt / S;
    const bool rk4 = (P.flags & kFRk4) != 0;
    spx = C.vx; spy = C.vy;
    for (int s = 0; s < S; ++s) {
        double n1x = 0, n1y = 0, n2x = 0, n2y = 0, n3x = 0, n3y = 0, n4x = 0, n4y = 0;
        if constexpr (NZ != kNoNoise) {
            constexpr int NC = MIS ? 3 : 2;
            float z[NC * 4];
            block_normals<NZ, NC>(R, c0_of(kStreamDyn, (uint32_t)s, 0), z);
            if constexpr (MIS) {
                noise_vec<MIS>(P, C, z[0], z[1], z[2], n1x, n1y);
                noise_vec<MIS>(P, C, z[3], z[4], z[5], n2x, n2y);
                noise_vec<MIS>(P, C, z[6], z[7], z[8], n3x, n3y);
                noise_vec<MIS>(P, C, z[9], z[10], z[11], n4x, n4y);
            } else {
                noise_vec<MIS>(P, C, 0.f, z[0], z[1], n1x, n1y);
                noise_vec<MIS>(P, C, 0.f, z[2], z[3], n2x, n2y);
                noise_vec<MIS>(P, C, 0.f, z[4], z[5], n3x, n3y);
                noise_vec<MIS>(P, C, 0.f, z[6], z[7], n4x, n4y);
            }
        }
        const double k1x = C.vx + n1x, k1y = C.vy + n1y;
        if (rk4) {
            const double k2x = C.vx + n2x, k2y = C.vy + n2y, k3x = C.vx + n3x, k3y = C.vy + n3y;
            const double k4x = C.vx + n4x, k4y = C.vy + n4y;
            x = x + (h / 6) * (k1x + 2 * k2x + 2 * k3x + k4x);
            y = y + (h / 6) * (k1y + 2 * k2y + 2 * k3y + k4y);
            spx = k4x; spy = k4y;
        } else {
            x = x + h * k1x;
            y = y + h * k1y;
            spx = k1x; spy = k1y;
        }
    }
}


__device__ __forceinline__ void load_env(const double* __restrict__ pos, const float* __restrict__ aux,
                                         const float* __restrict__ ep_ret, long long i, const KParams& P, EnvRegs& e) {
    const double2 p = reinterpret_cast<const double2*>(pos)[i];
    const float4 a = reinterpret_cast<const float4*>(aux)[i];
    e.x = p.x; e.y = p.y;
    e.f0x = (double)a.x; e.f0y = (double)a.y;
    e.h_abs = (double)a.z * P.dt;
    e.counter = __float_as_int(a.w);
    e.ep_ret = ep_ret[i];
}

// the carried RK45 state exactly as it is stored in HBM (so a fused rollout and a sequence of
// single steps produce identical bits)
__device__ __forceinline__ float hq_of(const KParams& P, double h_abs) { return (float)(h_abs * P.inv_dt); }

__device__ __forceinline__ void quantise_env(const KParams& P, EnvRegs& e) {
    e.f0x = (double)(float)e.f0x;
    e.f0y = (double)(float)e.f0y;
    e.h_abs = (double)hq_of(P, e.h_abs) * P.dt;
}

__device__ __forceinline__ void store_env(double* __restrict__ pos, float* __restrict__ aux, float* __restrict__ ep_ret,
                                          long long i, const KParams& P, const EnvRegs& e) {
    reinterpret_cast<double2*>(pos)[i] = make_double2(e.x, e.y);
    reinterpret_cast<float4*>(aux)[i] =
        make_float4((float)e.f0x, (float)e.f0y, hq_of(P, e.h_abs), __int_as_float(e.counter));
    ep_ret[i] = e.ep_ret;
}

// goal of (env, episode step): MR_Env.init_goal = (0,0) (MR_env.py:57) or a trajectory table
// row pointer of an env's trajectory in the goal table (nullptr: constant goal (0, 0)); loop-invariant for a lane
__device__ __forceinline__ const float2* goal_row_of(const KParams& P, uint32_t fl, const float* __restrict__ goal_table,
                                                     uint32_t env) {
    if (!(fl & kFGoalTable)) return nullptr;
    const int K = P.goal_K > 0 ? P.goal_K : 1, T = P.goal_T > 0 ? P.goal_T : 1;
    const int k = (K == 1) ? 0 : (int)(env % (uint32_t)K);
    return reinterpret_cast<const float2*>(goal_table) + (long long)k * T;
}
__device__ __forceinline__ float2 goal_from_row(const KParams& P, const float2* __restrict__ row, int32_t counter) {
    if (row == nullptr) return make_float2(0.f, 0.f);
    const int T = P.goal_T > 0 ? P.goal_T : 1;
    int r;  // clamp(counter, 0, T - 1)
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(counter), "s"(T - 1));
    return row[r];
}
__device__ __forceinline__ float2 goal_fetch(const KParams& P, uint32_t fl, const float* __restrict__ goal_table,
                                             uint32_t env, int32_t counter) {
    return goal_from_row(P, goal_row_of(P, fl, goal_table, env), counter);
}
__device__ __forceinline__ void goal_at(const KParams& P, uint32_t fl, const float* __restrict__ goal_table,
                                        uint32_t env, int32_t counter, double& gx, double& gy) {
    const float2 g = goal_fetch(P, fl, goal_table, env, counter);
    gx = (double)g.x; gy = (double)g.y;
}

struct StepOut {
    float obs[5];
    float rew;
    uint8_t done;
    float spx, spy;     // Simulator.state_prime = last RHS value evaluated (the reset's, after an auto-reset)
    float spx0, spy0;   // state_prime of the step itself
    double px, py;      // position after the step, before any auto-reset (MR_Env.last_pos)
    bool has_final;
    float fobs[5];
    float fret;
    int32_t flen;
    int32_t attempts;   // rk_step attempts of this env step (diagnostic output of the step kernel; dead code elsewhere)
};

// MR_Env.reset body for one env (MR_env.py:164-201 -> MR_simulator.py:21-34)
template <bool RK45, int NZ, bool MIS_CTOR>
__device__ __forceinline__ void reset_env(const KParams& P, const Rng& R, double x0, double y0, EnvRegs& e,
                                          double& spx, double& spy, const uint32_t* wr, bool need_f1 = true,
                                          bool in_init_box = false) {
    e.x = x0; e.y = y0;
    e.counter = 0;
    e.ep_ret = 0.f;
    spx = spy = 0.0;
    if constexpr (RK45) {
        const RhsCtx<MIS_CTOR> Z = zero_ctx<MIS_CTOR>(P);
        rk45_construct<NZ, MIS_CTOR>(P, Z, R, kStreamResetCtor, e.x, e.y, e.f0x, e.f0y, e.h_abs, spx, spy, need_f1, nullptr,
                                     wr, in_init_box);
    } else {
        e.f0x = e.f0y = 0.0;
        e.h_abs = P.dt;
    }
}

// The one Philox call of a reset: words 0,1 -> init position, words 2,3 -> the nominal constructor's F0 normals.
__device__ __forceinline__ void reset_words(const Rng& R, uint32_t (&w)[4]) {
    philox_call(R, c0_of(kStreamResetPos, 0, 0), w);
}
__device__ __forceinline__ void sample_init(const KParams& P, const uint32_t (&w)[4], double& x0, double& y0) {
    const double u0 = ((double)w[0] + 0.5) * 2.3283064365386963e-10;
    const double u1 = ((double)w[1] + 0.5) * 2.3283064365386963e-10;
    // init_space.sample() returns float32 (MR_env.py:40-42,173)
    x0 = (double)(float)(P.init_lo[0] + P.init_span[0] * u0);
    y0 = (double)(float)(P.init_lo[1] + P.init_span[1] * u1);
}

// Exploration policy: action ~ U[act_lo, act_lo + act_span) from the first two words of one Philox call.
// In RK45 mode that call is DYN block 0 call 0: its words 0,1 would only feed stage K1 of the first
// rk_step attempt, which never reaches a result (B1 = E1 = 0), so the policy reuses them and the
// fused kernels pay one Philox call less per step.  Fixed-step modes consume those words for noise,
// so there the policy has a call of its own (POLICY stream).
__device__ __forceinline__ constexpr uint32_t policy_c0(bool rk45) {
    return rk45 ? c0_of(kStreamDyn, 0, 0) : c0_of(kStreamPolicy, 0, 0);
}
__device__ __forceinline__ void action_from_words(const KParams& P, const uint32_t* w, float& f_t, float& al) {
    // fp32 on purpose (actions are float32 at the ABI): u = fma(float(w), 2^-32, 2^-33), a = fma(span, u, lo)
    const float u0 = __builtin_fmaf((float)w[0], 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    const float u1 = __builtin_fmaf((float)w[1], 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    f_t = __builtin_fmaf(P.act_span_f[0], u0, P.act_lo_f[0]);
    al = __builtin_fmaf(P.act_span_f[1], u1, P.act_lo_f[1]);
}
// Philox words of the step's hot path, drawn together up front (philox_multi): in RK45 mode the first
// rk_step attempt's calls DYN(0, 0..NDYN-1); slot 0 also
// carries the exploration policy's two words.  Later attempts (rare) and resets draw their own.
template <bool RK45, int NZ, bool MIS>
struct StepWords {
    static constexpr int NDYN = (RK45 && NZ != kNoNoise) ? (nz_coll(NZ) ? 2 : (MIS ? 5 : 3)) : 0;
    static constexpr int N = NDYN > 0 ? NDYN : 1;  // the constructor's F0/F1 live in the DYN block too
    uint32_t w[N][4];
};

// policy_words: slot 0's first two words are wanted even when no noise block is drawn (random policy, or the actor's
// Ornstein-Uhlenbeck pair -- the exploration noise takes the place of the exploration policy's uniforms)
template <bool RK45, int NZ, bool MIS>
__device__ __forceinline__ void step_prologue(const KParams& P, const Rng& R, bool random_policy,
                                              StepWords<RK45, NZ, MIS>& W, float& af, float& aa, bool policy_words = false) {
    using SW = StepWords<RK45, NZ, MIS>;
#ifdef MRSIM_AB_SHARE   // MEASUREMENT build only (tools/ab_rollout.py): what the nominal collapsed kernel would gain if the policy's two
    // words were shared by two consecutive steps (1.5 Philox calls per step instead of 2): odd steps pay ONE call, their second
    // call's words are a scramble of the first's.  Wrong draws, right instruction mix.
    if constexpr (nz_coll(NZ) && !MIS) {
        if (R.step_lo & 1u) {
            philox_call(R, c0_of(kStreamDyn, 0, 0), W.w[0]);
#pragma unroll
            for (int j = 0; j < 4; ++j) W.w[1][j] = W.w[0][(j + 1) & 3] ^ 0x9E3779B9u;
            if (random_policy) action_from_words(P, W.w[0], af, aa);
            return;
        }
    }
#endif
    if constexpr (SW::NDYN > 0) {
        uint32_t c0s[SW::N];
#pragma unroll
        for (int j = 0; j < SW::NDYN; ++j) c0s[j] = c0_of(kStreamDyn, 0, (uint32_t)j);
        philox_multi<SW::N>(R, c0s, W.w);
    } else {
        W.w[0][0] = W.w[0][1] = W.w[0][2] = W.w[0][3] = 0u;
        if (random_policy || policy_words) philox_call(R, policy_c0(RK45), W.w[0]);
    }
    if (random_policy) action_from_words(P, W.w[0], af, aa);
}

// RNG position of an auto-reset's draws (RESET_POS / RESET_CTOR streams): the step index at which the episode that ends
// took its FIRST step = step - (length - 1), 64-bit modular (oracle: orc_env_step).  One counter block per episode, known
// from the episode's start: a fused rollout can prepare the resets of a wave's envs together (mrsim_kernels.hip, reset cache).
__device__ __forceinline__ Rng reset_rng(const Rng& R, int32_t counter) {
    Rng Q = R;
    const uint32_t back = (uint32_t)(counter - 1);
    Q.step_lo = R.step_lo - back;
    Q.step_hi = R.step_hi - (R.step_lo < back ? 1u : 0u);
    return Q;
}

__device__ __forceinline__ void pack_obs(double x, double y, double gx, double gy, double d2, float (&obs)[5]) {
    // convert_state (MR_env.py:100-116); dist emitted as the fp32 sqrt of the fp64 squared distance
    obs[0] = (float)x; obs[1] = (float)y; obs[2] = (float)gx; obs[3] = (float)gy;
    obs[4] = __builtin_amdgcn_sqrtf((float)d2);  // v_sqrt_f32 (1 ulp); inputs are >= 0 and far from denormal
}

// The state an auto-reset puts an env in (MR_Env.reset(init=None) on the env object of an episode loop): start position,
// RK45 constructor.  `counter` = length of the episode that ends at step R (its draws sit at reset_rng(R, counter)).
// Shared by the in-step reset (env_step) and the fused rollout's reset cache (mrsim_kernels.hip), which calls it for the
// envs of a wave together, ahead of time: same function of (env, first step of the episode), same bits.
template <bool RK45, int NZ, bool MIS>
__device__ __forceinline__ void auto_reset_env(const KParams& P, const Rng& R, uint32_t fl, int32_t counter, EnvRegs& e,
                                               double& rx, double& ry) {
    double x0, y0;
    uint32_t wr[4];
    const Rng Q = reset_rng(R, counter);
    reset_words(Q, wr);
    sample_init(P, wr, x0, y0);
    // The host-certified shortcut of the constructor test only where resets are hot: on a goal table episodes end at
    // different steps, so some lane of most waves resets at most steps.  With the constant goal all
    // episodes of the DDPG workload end together every max_timesteps + 1 steps, and its flag-specialised kernel keeps
    // the code (and the register allocation of its time loop) it had.
    // Which law the RK45 constructor inside reset runs under (MR_env.py:181-183 sets is_mismatched AFTER
    // reset_start_pos): the re-used env object of an episode loop (RL/MR_ddpg.py:270) still carries the previous
    // episode's law -- under is_mismatched the stale first stage of the new episode is the drift (0.2, -0.1) (+ noise) --
    // a fresh env (kFResetFresh) the nominal one.
    const bool need_sp = (fl & kFOutStatePrime) != 0;
    if constexpr (MIS) {
        if (fl & kFResetFresh) reset_env<RK45, NZ, false>(P, Q, x0, y0, e, rx, ry, wr, need_sp, (fl & kFGoalTable) != 0);
        else reset_env<RK45, NZ, true>(P, Q, x0, y0, e, rx, ry, wr, need_sp, false);
    } else {
        reset_env<RK45, NZ, false>(P, Q, x0, y0, e, rx, ry, wr, need_sp, /*in_init_box=*/(fl & kFGoalTable) != 0);
    }
}

// MR_Env.step for one env (MR_env.py:70-98).  DEFER: on a terminal step only the terminal outputs are recorded; the caller
// performs the auto-reset itself (the reset cache of the goal-table rollout, mrsim_kernels.hip).
template <bool RK45, int NZ, bool MIS, bool DEFER = false>
__device__ __forceinline__ void env_step(const KParams& P, const Rng& R, const float* __restrict__ goal_table,
                                         EnvRegs& e, double act_f, double act_a, const StepWords<RK45, NZ, MIS>& W,
                                         uint32_t fl, StepOut& o, int& fail, const double2* __restrict__ sincos_lds = nullptr,
                                         const float2* goal_pre = nullptr, const float2* goal0_pre = nullptr, Lm1* lm = nullptr) {
    e.counter += 1;  // :80
    // the goal of this step only depends on the counter: the one-launch-per-step kernel fetches it now, so that the table
    // read (an L1/L2 hit, but hundreds of cycles) completes behind the integrator instead of stalling the termination
    // check; the fused rollout hands in the goal it prefetched before the PREVIOUS step's stores (see mr_rollout_kernel)
    const float2 goal_f = goal_pre != nullptr ? *goal_pre : goal_fetch(P, fl, goal_table, R.env, e.counter);
    const RhsCtx<MIS> C = make_ctx<MIS>(P, act_f, act_a, sincos_lds);
    double spx = 0.0, spy = 0.0;
    if constexpr (RK45) {
        bool fast = false;
        o.attempts = 1;
        // (noise_math = spec is the test-oriented bit-exact mode: its long Box-Muller would only be duplicated)
        if constexpr (nz_fast(NZ) && MRSIM_FAST_STEP != 0) {
            if (!(fl & kFOutStatePrime)) fast = rk45_fast_step<NZ, MIS>(P, C, R, &W.w[0][0], e, lm, fail);
        }
#ifdef MRSIM_BUDGET_BUILD  // tools/isa_budget.py only: drop the general path so that the time loop is the common path alone
        if (!fast) __builtin_trap();
#endif
        if (__builtin_expect(!fast, 0)) {
            SubStep LS = rk45_integrate<NZ, MIS>(P, C, R, e.x, e.y, e.f0x, e.f0y, e.h_abs, fail, &W.w[0][0]);  // MR_simulator.py:42-45
            rk45_construct<NZ, MIS>(P, C, R, kStreamCtor, e.x, e.y, e.f0x, e.f0y, e.h_abs, spx, spy,           // :46-50
                                    (fl & kFOutStatePrime) != 0, NZ != kNoNoise ? &LS : nullptr);
            if (lm != nullptr) lm->kb = lm1_bound(e.f0x, e.f0y);
            o.attempts = (int32_t)LS.attempt;
        }
    } else {
        fixed_integrate<NZ, MIS>(P, C, R, e.x, e.y, spx, spy);
        e.f0x = spx; e.f0y = spy; e.h_abs = P.dt;
        o.attempts = P.substeps > 0 ? P.substeps : 1;
    }
    double gx = (double)goal_f.x, gy = (double)goal_f.y;
    const double dx = gx - e.x, dy = gy - e.y;
    const double d2 = __builtin_fma(dx, dx, dy * dy);
    // end (:136-152); Box.contains as a numeric bounds test; distances compared squared
    bool inb;
    if (__builtin_expect((fl & kFSymBounds) != 0, 1)) {
        // the reference's Box is symmetric and equal in x, y, goal_x, goal_y (+-5000): one bound, two
        // SGPRs instead of sixteen (the per-component form below spills scalar registers in the loop)
        // (two compares with |.| source modifiers each; fmax(|x|, |y|) costs two canonicalising v_max_f64 more)
        inb = (__builtin_fabs(e.x) <= P.sym_bound) && (__builtin_fabs(e.y) <= P.sym_bound) &&
              (__builtin_fabs(gx) <= P.sym_bound) && (__builtin_fabs(gy) <= P.sym_bound);
    } else {
        inb = (e.x >= P.obs_lo[0]) && (e.x <= P.obs_hi[0]) && (e.y >= P.obs_lo[1]) && (e.y <= P.obs_hi[1]) &&
              (gx >= P.obs_lo[2]) && (gx <= P.obs_hi[2]) && (gy >= P.obs_lo[3]) && (gy <= P.obs_hi[3]);
    }
    inb = inb && (d2 <= P.dmax2) && (d2 >= P.dmin2);
    const bool timeout = e.counter > P.max_timesteps;
    const bool reached = d2 < P.min_dist2;
    const bool done = (!inb) || timeout || reached;
    float rew = 10.0f;  // :89
    if (fl & kFRewardGoal) rew = reached ? 100.0f : ((!inb || timeout) ? -100.0f : -0.1f);  // :118-134
    e.ep_ret += rew;
    pack_obs(e.x, e.y, gx, gy, d2, o.obs);
    o.rew = rew;
    o.done = done ? 1 : 0;
    o.spx = o.spx0 = (float)spx; o.spy = o.spy0 = (float)spy;
    o.px = e.x; o.py = e.y;
    o.has_final = false;
#ifdef MRSIM_BUDGET_BUILD
    if (done) __builtin_trap();  // ... and the auto-reset block (cold in the DDPG workload: once per 51 steps)
#endif
    if (__builtin_expect(done && (fl & kFAutoReset), 0)) {
        // extension: same-step auto-reset; terminal values go to the final_* outputs
        o.has_final = true;
#pragma unroll
        for (int j = 0; j < 5; ++j) o.fobs[j] = o.obs[j];
        o.fret = e.ep_ret;
        o.flen = e.counter;
        if constexpr (DEFER) return;
        double rx, ry;
        auto_reset_env<RK45, NZ, MIS>(P, R, fl, e.counter, e, rx, ry);
        if (lm != nullptr) lm->kb = lm1_bound(e.f0x, e.f0y);
        if constexpr (RK45) { o.spx = (float)rx; o.spy = (float)ry; }  // state_prime = last RHS value
        if (goal0_pre != nullptr) { gx = (double)goal0_pre->x; gy = (double)goal0_pre->y; }  // row 0: loaded once per launch
        else goal_at(P, fl, goal_table, R.env, 0, gx, gy);
        const double ex = gx - e.x, ey = gy - e.y;
        pack_obs(e.x, e.y, gx, gy, __builtin_fma(ex, ex, ey * ey), o.obs);
    }
}

}  // namespace mrsim
