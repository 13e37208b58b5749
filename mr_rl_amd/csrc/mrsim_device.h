// mrsim_device.h -- device-side building blocks of the MR_env.step() hot path
// for gfx950 (MI355X, wave64).  One lane advances one environment.
//
// What is restated here (reference citations are /root/reference/<file>:<line>):
//   Simulator.simulate      MR_simulator.py:58-88   the ODE right-hand side (+ per-eval noise)
//   Simulator.step          MR_simulator.py:36-52   integrate one time_span with SciPy RK45,
//                                                    then build the NEXT RK45 object
//   scipy RK45 semantics    rk.py rk_step/_step_impl, common.py select_initial_step
//   MR_Env.step/end/...     MR_env.py:70-152
// Because the RHS ignores (t, y), the RK stage arguments never matter: only the
// stage VALUES K[i] (action velocity + fresh noise), the solution weights B and
// the error weights E do.  The first stage K[0] is the derivative the RK45 object
// computed when it was constructed -- at the end of the PREVIOUS env step, with the
// previous action -- so it is carried in HBM between steps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrsim {

constexpr int kBlock = 256;  // 4 waves; 16-B records => every wave moves 1 KiB per array

// ---------------------------------------------------------------------------
// kernel-side parameter block (passed by value -> kernarg/SGPRs)
// ---------------------------------------------------------------------------
struct KParams {
    double dt, rtol, atol, a0, sigma, sigma4;
    double min_dist2;  // min_dist2goal^2
    double obs_lo[5], obs_hi[5];
    double dmax2;      // obs_hi[4]^2
    double init_lo[2], init_span[2];
    double act_lo[2], act_span[2];
    double h1_thresh;  // 0.01 / dt^5 : select_initial_step's h1 >= dt  <=>  max(d1,d2) <= h1_thresh
    float h1_thresh2_f, dt2_f;
    int32_t substeps, reward_mode, max_timesteps, auto_reset, goal_K, goal_T;
    int32_t integrator, pad;
    uint32_t seed_lo, seed_hi, step_lo, step_hi;
    uint32_t env_id0, pad2;
    long long n;
    const unsigned long long* step_base;  // optional device word added to (step_hi:step_lo)
};

// effective 64-bit step index of this launch (+ t for the fused rollout)
__device__ __forceinline__ void step_words(const KParams& P, unsigned long long t, uint32_t& lo, uint32_t& hi) {
    unsigned long long s = (((unsigned long long)P.step_hi << 32) | P.step_lo) + t;
    if (P.step_base != nullptr) s += *P.step_base;  // uniform scalar load
    lo = (uint32_t)s;
    hi = (uint32_t)(s >> 32);
}
__device__ __forceinline__ struct Rng make_rng(const KParams& P, long long i, unsigned long long t = 0);

// ---------------------------------------------------------------------------
// RNG: Philox4x32-10, counter = {c0, step lo, step hi, global env id}, key = seed
// c0 = stream<<28 | block<<4 | call.  Same definition as oracle/mrsim_oracle.c.
// ---------------------------------------------------------------------------
enum : uint32_t { kStreamDyn = 0, kStreamCtor = 1, kStreamResetPos = 2, kStreamResetCtor = 3, kStreamPolicy = 4 };
__device__ __forceinline__ constexpr uint32_t c0_of(uint32_t stream, uint32_t block, uint32_t call) {
    return (stream << 28) | (block << 4) | call;
}

struct Rng {
    uint32_t k0, k1, step_lo, step_hi, env;
};

__device__ __forceinline__ Rng make_rng(const KParams& P, long long i, unsigned long long t) {
    Rng R;
    R.k0 = P.seed_lo; R.k1 = P.seed_hi;
    step_words(P, t, R.step_lo, R.step_hi);
    R.env = P.env_id0 + (uint32_t)i;
    return R;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ __forceinline__ void philox_call(const Rng& R, uint32_t c0, uint32_t (&o)[4]) {
    philox4x32_10(c0, R.step_lo, R.step_hi, R.env, R.k0, R.k1, o);
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

// Box-Muller, specified operation by operation (see oracle/mrsim_oracle.c: orc_box_muller).
__device__ __forceinline__ void box_muller(uint32_t ua, uint32_t ub, float& z0, float& z1) {
    const float u = __builtin_fmaf((float)ua, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    const float r = __builtin_sqrtf(-2.0f * spec_logf(u));
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
}

// the first NCALLS*4 normals of one block (rk_step attempt / constructor / fixed sub-step)
template <int NCALLS>
__device__ __forceinline__ void block_normals(const Rng& R, uint32_t c0base, float (&z)[NCALLS * 4]) {
#pragma unroll
    for (int j = 0; j < NCALLS; ++j) {
        uint32_t o[4];
        philox_call(R, c0base | (uint32_t)j, o);
        box_muller(o[0], o[1], z[4 * j + 0], z[4 * j + 1]);
        box_muller(o[2], o[3], z[4 * j + 2], z[4 * j + 3]);
    }
}

__device__ __forceinline__ void uniform2(const Rng& R, uint32_t c0, double& u0, double& u1) {
    uint32_t o[4];
    philox_call(R, c0, o);
    u0 = ((double)o[0] + 0.5) * 2.3283064365386963e-10;
    u1 = ((double)o[1] + 0.5) * 2.3283064365386963e-10;
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

// sin and cos of a double: two-term fma Cody-Waite reduction by pi/2 (exact products, so the
// reduced argument is good to ~2e-16 absolute for |a| < 1e15) + fdlibm __kernel_sin/__kernel_cos
// polynomials on [-pi/4, pi/4].  Non-finite or |a| >= 1e15: NaN (numpy would Payne-Hanek).
__device__ __forceinline__ void sincos_f64(double a, double& s, double& c) {
    const double k = __builtin_rint(a * 0.63661977236758138);
    double r = __builtin_fma(-k, 1.5707963267948966, a);
    r = __builtin_fma(-k, 6.123233995736766e-17, r);
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(ps * z, r, r);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double cr = __builtin_fma(pc * z, z, __builtin_fma(-0.5, z, 1.0));
    const int q = (int)(long long)k & 3;
    const double s0 = (q & 1) ? cr : sr;
    const double c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
    if (!(__builtin_fabs(a) < 1e15)) { s = __builtin_nan(""); c = s; }
}

// ---------------------------------------------------------------------------
// Simulator.simulate (MR_simulator.py:58-88): per-action constants + per-eval noise
// ---------------------------------------------------------------------------
template <bool MIS>
struct RhsCtx {
    // nominal:     k = ((a0*f)*cos a + n_x, (a0*f)*sin a + n_y)                       :82-83
    // mismatched:  a0' = a0 + (f/4)*0.8 + n_a;  k = ((a0'*f)*cos(a+0.1) + n_x + 0.2,
    //                                                (a0'*f)*sin(a-0.15) + n_y - 0.1)  :55-56,78-80
    double vx, vy;        // nominal base velocity
    double a0b, f, cA, sB;  // mismatched pieces
};

template <bool MIS>
__device__ __forceinline__ RhsCtx<MIS> make_ctx(const KParams& P, double f_t, double al) {
    RhsCtx<MIS> C;
    if constexpr (MIS) {
        C.a0b = P.a0 + (f_t / 4) * 0.8;
        C.f = f_t;
        double t0, t1;
        sincos_f64(al + 0.1, t0, C.cA);
        sincos_f64(al - 0.15, C.sB, t1);
        C.vx = C.vy = 0.0;
    } else {
        double s, c;
        sincos_f64(al, s, c);
        const double af = P.a0 * f_t;
        C.vx = af * c;
        C.vy = af * s;
        C.a0b = C.f = C.cA = C.sB = 0.0;
    }
    return C;
}

// zero action (Simulator.reset_start_pos, MR_simulator.py:30): velocity terms vanish
template <bool MIS>
__device__ __forceinline__ RhsCtx<MIS> zero_ctx(const KParams& P) {
    RhsCtx<MIS> C;
    C.vx = C.vy = 0.0;
    C.a0b = P.a0; C.f = 0.0; C.cA = 1.0; C.sB = 0.0;
    return C;
}

template <bool NOISE, bool MIS>
__device__ __forceinline__ void rhs_eval(const KParams& P, const RhsCtx<MIS>& C, float za, float zx, float zy,
                                         double& kx, double& ky) {
    if constexpr (MIS) {
        double a0e = C.a0b;
        if constexpr (NOISE) a0e = a0e + P.sigma4 * (double)za;
        const double af = a0e * C.f;
        kx = af * C.cA; ky = af * C.sB;
        if constexpr (NOISE) { kx = kx + P.sigma * (double)zx; ky = ky + P.sigma * (double)zy; }
        kx = kx + 0.2; ky = ky - 0.1;
    } else {
        kx = C.vx; ky = C.vy;
        if constexpr (NOISE) { kx = kx + P.sigma * (double)zx; ky = ky + P.sigma * (double)zy; }
    }
}

// scipy common.norm of a 2-vector
__device__ __forceinline__ double rms2(double a, double b) { return sqrt(a * a + b * b) / 1.4142135623730951; }

// ---------------------------------------------------------------------------
// RungeKutta.__init__ + select_initial_step: what Simulator.step does after integrating
// (MR_simulator.py:46-50) and what reset_start_pos does (:31-34).  Two RHS evaluations
// f0 (-> K[0] of the next step) and f1 (-> Simulator.state_prime).
// ---------------------------------------------------------------------------
template <bool NOISE, bool MIS>
__device__ __forceinline__ void rk45_construct(const KParams& P, const RhsCtx<MIS>& C, const Rng& R, uint32_t stream,
                                               double x, double y, double& f0x, double& f0y, double& h_abs,
                                               double& spx, double& spy) {
    double f1x, f1y;
    if constexpr (NOISE) {
        constexpr int NC = MIS ? 2 : 1;
        float z[NC * 4];
        block_normals<NC>(R, c0_of(stream, 0, 0), z);
        if constexpr (MIS) {
            rhs_eval<NOISE, MIS>(P, C, z[0], z[1], z[2], f0x, f0y);
            rhs_eval<NOISE, MIS>(P, C, z[3], z[4], z[5], f1x, f1y);
        } else {
            rhs_eval<NOISE, MIS>(P, C, 0.f, z[0], z[1], f0x, f0y);
            rhs_eval<NOISE, MIS>(P, C, 0.f, z[2], z[3], f1x, f1y);
        }
    } else {
        rhs_eval<NOISE, MIS>(P, C, 0.f, 0.f, 0.f, f0x, f0y);
        f1x = f0x; f1y = f0y;
    }
    spx = f1x; spy = f1y;

    const double sc0 = P.atol + fabs(x) * P.rtol;
    const double sc1 = P.atol + fabs(y) * P.rtol;
    // Fast path (fp32, 5 % margins): decide "h_abs == interval_length" without divisions, square
    // roots or pow.  With r = 1/scale: d0^2 = D0/2, d1^2 = D1/2, (d2*h0)^2 = DD/2.
    //   100*h0 >= dt   <=  d0,d1 >= 1e-5  and  d0/d1 >= dt
    //   h1 >= dt       <=  d1 <= TH and d2 <= TH, h0 = min(0.01 d0/d1, dt),  TH = 0.01/dt^5
    // Any NaN/inf makes a comparison false and falls through to the exact path.
    {
        const float r0 = __builtin_amdgcn_rcpf((float)sc0), r1 = __builtin_amdgcn_rcpf((float)sc1);
        const float y0s = (float)x * r0, y1s = (float)y * r1;
        const float g0 = (float)f0x * r0, g1 = (float)f0y * r1;
        const float e0 = (float)(f1x - f0x) * r0, e1 = (float)(f1y - f0y) * r1;
        const float D0 = y0s * y0s + y1s * y1s;
        const float D1 = g0 * g0 + g1 * g1;
        const float DD = e0 * e0 + e1 * e1;
        const float TH2 = P.h1_thresh2_f;
        const bool fast = (D0 > 1e-9f) && (D1 > 1e-9f) && (D0 >= 1.05f * P.dt2_f * D1) && (D1 <= 1.9f * TH2) &&
                          (DD <= 1.9f * TH2 * P.dt2_f) && (DD * D1 <= 1.9e-4f * TH2 * D0);
        if (fast) { h_abs = P.dt; return; }
    }
    // exact path: select_initial_step(fun, t0, y0, t_bound, inf, f0, +1, order=4, rtol, atol)
    const double d0 = rms2(x / sc0, y / sc1);
    const double d1 = rms2(f0x / sc0, f0y / sc1);
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    h0 = fmin(h0, P.dt);
    const double d2 = rms2((f1x - f0x) / sc0, (f1y - f0y) / sc1) / h0;
    double h1;
    if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
    else h1 = fifth_root(0.01 / fmax(d1, d2));
    h_abs = fmin(fmin(100 * h0, h1), P.dt);
}

// ---------------------------------------------------------------------------
// One RK45 env step: OdeSolver.step loop + RungeKutta._step_impl + rk_step, in time
// relative to the start of the env step (tau in [0, dt]).
// ---------------------------------------------------------------------------
constexpr double kB0 = 35.0 / 384, kB2 = 500.0 / 1113, kB3 = 125.0 / 192, kB4 = -2187.0 / 6784, kB5 = 11.0 / 84;
constexpr double kE0 = -71.0 / 57600, kE2 = 71.0 / 16695, kE3 = -71.0 / 1920, kE4 = 17253.0 / 339200,
                 kE5 = -22.0 / 525, kE6 = 1.0 / 40;
constexpr int kMaxAttempts = 4096;  // every lane leaves the loop: bounded spin

template <bool NOISE, bool MIS>
__device__ __forceinline__ void rk45_integrate(const KParams& P, const RhsCtx<MIS>& C, const Rng& R, double& x,
                                               double& y, double f0x, double f0y, double h_abs, int& fail) {
    double tau = 0.0;
    uint32_t attempt = 0;
    // stage values without noise are the same for every attempt
    double kx0, ky0;
    rhs_eval<false, MIS>(P, C, 0.f, 0.f, 0.f, kx0, ky0);
    while (tau < P.dt) {
        bool accepted = false, rejected = false;
        double xn = x, yn = y, fnx = f0x, fny = f0y, tn = tau;
        while (!accepted) {
            tn = tau + h_abs;
            if (tn > P.dt) tn = P.dt;
            const double h = tn - tau;
            h_abs = h;
            // rk_step: K[0] = f (carried), K[1..5] fresh RHS values, K[6] = f_new.
            // B[1] = E[1] = 0, so K[1] never reaches a result (its draws are still consumed).
            double k2x, k2y, k3x, k3y, k4x, k4y, k5x, k5y, k6x, k6y;
            if constexpr (NOISE) {
                constexpr int NC = MIS ? 5 : 3;
                float z[NC * 4];
                block_normals<NC>(R, c0_of(kStreamDyn, attempt, 0), z);
                if constexpr (MIS) {
                    rhs_eval<NOISE, MIS>(P, C, z[3], z[4], z[5], k2x, k2y);
                    rhs_eval<NOISE, MIS>(P, C, z[6], z[7], z[8], k3x, k3y);
                    rhs_eval<NOISE, MIS>(P, C, z[9], z[10], z[11], k4x, k4y);
                    rhs_eval<NOISE, MIS>(P, C, z[12], z[13], z[14], k5x, k5y);
                    rhs_eval<NOISE, MIS>(P, C, z[15], z[16], z[17], k6x, k6y);
                } else {
                    rhs_eval<NOISE, MIS>(P, C, 0.f, z[2], z[3], k2x, k2y);
                    rhs_eval<NOISE, MIS>(P, C, 0.f, z[4], z[5], k3x, k3y);
                    rhs_eval<NOISE, MIS>(P, C, 0.f, z[6], z[7], k4x, k4y);
                    rhs_eval<NOISE, MIS>(P, C, 0.f, z[8], z[9], k5x, k5y);
                    rhs_eval<NOISE, MIS>(P, C, 0.f, z[10], z[11], k6x, k6y);
                }
            } else {
                k2x = k3x = k4x = k5x = k6x = kx0;
                k2y = k3y = k4y = k5y = k6y = ky0;
            }
            ++attempt;
            // y_new = y + h * dot(K[:-1].T, B)   (sequential, unfused -- as the oracle)
            const double sx = f0x * kB0 + k2x * kB2 + k3x * kB3 + k4x * kB4 + k5x * kB5;
            const double sy = f0y * kB0 + k2y * kB2 + k3y * kB3 + k4y * kB4 + k5y * kB5;
            xn = x + h * sx;
            yn = y + h * sy;
            fnx = k6x; fny = k6y;
            // error_norm = norm(dot(K.T, E) * h / scale),  scale = atol + max(|y|,|y_new|) * rtol
            const double ex = (f0x * kE0 + k2x * kE2 + k3x * kE3 + k4x * kE4 + k5x * kE5 + k6x * kE6) * h;
            const double ey = (f0y * kE0 + k2y * kE2 + k3y * kE3 + k4y * kE4 + k5y * kE5 + k6y * kE6) * h;
            const double sc0 = P.atol + fmax(fabs(x), fabs(xn)) * P.rtol;
            const double sc1 = P.atol + fmax(fabs(y), fabs(yn)) * P.rtol;
            const bool last = !(tn < P.dt);
            // fast accept (no division / sqrt / pow): error_norm^2 = q / lim with a 2 % margin; the
            // step-size factor is only needed when another sub-step follows.
            const double s00 = sc0 * sc0, s11 = sc1 * sc1;
            const double q = ex * ex * s11 + ey * ey * s00;
            const double lim = 2.0 * s00 * s11;
            if (last && q < 0.98 * lim) {
                accepted = true;
            } else {
                const double error_norm = rms2(ex / sc0, ey / sc1);
                if (error_norm < 1.0) {
                    double factor = (error_norm == 0.0) ? 10.0 : fmin(10.0, 0.9 * inv_fifth_root(error_norm));
                    if (rejected) factor = fmin(1.0, factor);
                    h_abs *= factor;
                    accepted = true;
                } else {
                    h_abs *= fmax(0.2, 0.9 * inv_fifth_root(error_norm));
                    rejected = true;
                }
            }
            if (attempt >= (uint32_t)kMaxAttempts && !accepted) {
                // The reference would fail with TOO_SMALL_STEP / loop forever on NaN input.
                fail |= 1;
                accepted = true;
                tn = P.dt;
            }
        }
        tau = tn; x = xn; y = yn; f0x = fnx; f0y = fny;
        if (attempt >= (uint32_t)kMaxAttempts) { fail |= 1; break; }
    }
}

// Build extension (BASELINE configs 2/3): fixed-step Euler / classical RK4, noise added to the
// derivative at every RHS evaluation exactly as `simulate` does.  Mirrors the oracle's loop.
template <bool NOISE, bool MIS>
__device__ __forceinline__ void fixed_integrate(const KParams& P, const RhsCtx<MIS>& C, const Rng& R, double& x,
                                                double& y, double& spx, double& spy) {
    const int S = P.substeps > 0 ? P.substeps : 1;
    const double h = P.dt / S;
    const bool rk4 = (P.integrator == 2);
    for (int s = 0; s < S; ++s) {
        double k1x, k1y, k2x, k2y, k3x, k3y, k4x, k4y;
        if constexpr (NOISE) {
            constexpr int NC = MIS ? 3 : 2;
            float z[NC * 4];
            block_normals<NC>(R, c0_of(kStreamDyn, (uint32_t)s, 0), z);
            if constexpr (MIS) {
                rhs_eval<NOISE, MIS>(P, C, z[0], z[1], z[2], k1x, k1y);
                rhs_eval<NOISE, MIS>(P, C, z[3], z[4], z[5], k2x, k2y);
                rhs_eval<NOISE, MIS>(P, C, z[6], z[7], z[8], k3x, k3y);
                rhs_eval<NOISE, MIS>(P, C, z[9], z[10], z[11], k4x, k4y);
            } else {
                rhs_eval<NOISE, MIS>(P, C, 0.f, z[0], z[1], k1x, k1y);
                rhs_eval<NOISE, MIS>(P, C, 0.f, z[2], z[3], k2x, k2y);
                rhs_eval<NOISE, MIS>(P, C, 0.f, z[4], z[5], k3x, k3y);
                rhs_eval<NOISE, MIS>(P, C, 0.f, z[6], z[7], k4x, k4y);
            }
        } else {
            rhs_eval<NOISE, MIS>(P, C, 0.f, 0.f, 0.f, k1x, k1y);
            k2x = k3x = k4x = k1x; k2y = k3y = k4y = k1y;
        }
        if (rk4) {
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

// quantise the carried RK45 state exactly as it is stored in HBM (so a fused rollout and a
// sequence of single steps produce identical bits)
__device__ __forceinline__ void quantise_env(const KParams& P, EnvRegs& e) {
    e.f0x = (double)(float)e.f0x;
    e.f0y = (double)(float)e.f0y;
    e.h_abs = (double)(float)(e.h_abs / P.dt) * P.dt;
}

__device__ __forceinline__ void store_env(double* __restrict__ pos, float* __restrict__ aux, float* __restrict__ ep_ret,
                                          long long i, const KParams& P, const EnvRegs& e) {
    reinterpret_cast<double2*>(pos)[i] = make_double2(e.x, e.y);
    reinterpret_cast<float4*>(aux)[i] =
        make_float4((float)e.f0x, (float)e.f0y, (float)(e.h_abs / P.dt), __int_as_float(e.counter));
    ep_ret[i] = e.ep_ret;
}

// goal of (env, episode step): MR_Env.init_goal = (0,0) (MR_env.py:57) or a trajectory table
__device__ __forceinline__ void goal_at(const KParams& P, const float* __restrict__ goal_table, uint32_t env,
                                        int32_t counter, double& gx, double& gy) {
    if (goal_table == nullptr) { gx = 0.0; gy = 0.0; return; }
    const int K = P.goal_K > 0 ? P.goal_K : 1, T = P.goal_T > 0 ? P.goal_T : 1;
    const int k = (K == 1) ? 0 : (int)(env % (uint32_t)K);
    const int r = counter < 0 ? 0 : (counter >= T ? T - 1 : counter);
    const float2 g = reinterpret_cast<const float2*>(goal_table)[(long long)k * T + r];
    gx = (double)g.x; gy = (double)g.y;
}

struct StepOut {
    float obs[5];
    float rew;
    uint8_t done;
    float spx, spy;
    bool has_final;
    float fobs[5];
    float fret;
    int32_t flen;
    float act_f, act_a;
};

// MR_Env.reset body for one env (MR_env.py:164-201 -> MR_simulator.py:21-34)
template <bool RK45, bool NOISE, bool MIS_CTOR>
__device__ __forceinline__ void reset_env(const KParams& P, const Rng& R, double x0, double y0, EnvRegs& e,
                                          double& spx, double& spy) {
    e.x = x0; e.y = y0;
    e.counter = 0;
    e.ep_ret = 0.f;
    spx = spy = 0.0;
    if constexpr (RK45) {
        const RhsCtx<MIS_CTOR> Z = zero_ctx<MIS_CTOR>(P);
        rk45_construct<NOISE, MIS_CTOR>(P, Z, R, kStreamResetCtor, e.x, e.y, e.f0x, e.f0y, e.h_abs, spx, spy);
    } else {
        e.f0x = e.f0y = 0.0;
        e.h_abs = P.dt;
    }
}

__device__ __forceinline__ void sample_init(const KParams& P, const Rng& R, double& x0, double& y0) {
    double u0, u1;
    uniform2(R, c0_of(kStreamResetPos, 0, 0), u0, u1);
    // init_space.sample() returns float32 (MR_env.py:40-42,173)
    x0 = (double)(float)(P.init_lo[0] + P.init_span[0] * u0);
    y0 = (double)(float)(P.init_lo[1] + P.init_span[1] * u1);
}

__device__ __forceinline__ void random_action(const KParams& P, const Rng& R, float& f_t, float& al) {
    double u0, u1;
    uniform2(R, c0_of(kStreamPolicy, 0, 0), u0, u1);
    f_t = (float)(P.act_lo[0] + P.act_span[0] * u0);
    al = (float)(P.act_lo[1] + P.act_span[1] * u1);
}

// MR_Env.step for one env (MR_env.py:70-98)
template <bool RK45, bool NOISE, bool MIS>
__device__ __forceinline__ void env_step(const KParams& P, const Rng& R, const float* __restrict__ goal_table,
                                         EnvRegs& e, float act_f, float act_a, StepOut& o, int& fail) {
    e.counter += 1;  // :80
    const double f_t = (double)act_f, al = (double)act_a;
    const RhsCtx<MIS> C = make_ctx<MIS>(P, f_t, al);
    double spx = 0.0, spy = 0.0;
    if constexpr (RK45) {
        rk45_integrate<NOISE, MIS>(P, C, R, e.x, e.y, e.f0x, e.f0y, e.h_abs, fail);              // MR_simulator.py:42-45
        rk45_construct<NOISE, MIS>(P, C, R, kStreamCtor, e.x, e.y, e.f0x, e.f0y, e.h_abs, spx, spy);  // :46-50
    } else {
        fixed_integrate<NOISE, MIS>(P, C, R, e.x, e.y, spx, spy);
        e.f0x = spx; e.f0y = spy; e.h_abs = P.dt;
    }
    // convert_state (:100-116)
    double gx, gy;
    goal_at(P, goal_table, R.env, e.counter, gx, gy);
    const double dx = gx - e.x, dy = gy - e.y;
    const double d2 = dx * dx + dy * dy;
    // end (:136-152) / Box.contains as a numeric bounds test; d compared squared
    const bool inb = (e.x >= P.obs_lo[0]) && (e.x <= P.obs_hi[0]) && (e.y >= P.obs_lo[1]) && (e.y <= P.obs_hi[1]) &&
                     (gx >= P.obs_lo[2]) && (gx <= P.obs_hi[2]) && (gy >= P.obs_lo[3]) && (gy <= P.obs_hi[3]) &&
                     (d2 <= P.dmax2) && (P.obs_lo[4] <= 0.0 || d2 >= P.obs_lo[4] * P.obs_lo[4]);
    const bool timeout = e.counter > P.max_timesteps;
    const bool reached = d2 < P.min_dist2;
    const bool done = (!inb) || timeout || reached;
    float rew = 10.0f;  // :89
    if (P.reward_mode == 1) rew = reached ? 100.0f : ((!inb || timeout) ? -100.0f : -0.1f);  // :118-134
    e.ep_ret += rew;
    o.obs[0] = (float)e.x; o.obs[1] = (float)e.y; o.obs[2] = (float)gx; o.obs[3] = (float)gy;
    o.obs[4] = __builtin_sqrtf((float)d2);
    o.rew = rew;
    o.done = done ? 1 : 0;
    o.spx = (float)spx; o.spy = (float)spy;
    o.has_final = false;
    if (done && P.auto_reset) {
        // extension: same-step auto-reset; terminal values go to the final_* outputs
        o.has_final = true;
#pragma unroll
        for (int j = 0; j < 5; ++j) o.fobs[j] = o.obs[j];
        o.fret = e.ep_ret;
        o.flen = e.counter;
        double x0, y0, rx, ry;
        sample_init(P, R, x0, y0);
        reset_env<RK45, NOISE, false>(P, R, x0, y0, e, rx, ry);
        if constexpr (RK45) { o.spx = (float)rx; o.spy = (float)ry; }  // state_prime = last RHS value
        goal_at(P, goal_table, R.env, 0, gx, gy);
        const double ex = gx - e.x, ey = gy - e.y;
        o.obs[0] = (float)e.x; o.obs[1] = (float)e.y; o.obs[2] = (float)gx; o.obs[3] = (float)gy;
        o.obs[4] = __builtin_sqrtf((float)(ex * ex + ey * ey));
    }
}

}  // namespace mrsim
