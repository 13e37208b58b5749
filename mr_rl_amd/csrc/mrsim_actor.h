// mrsim_actor.h -- the DDPG actor as an in-kernel policy source for gfx950 (MI355X, wave64).
//
// What is restated here (reference citations are /root/reference/<file>:<line>):
//   ActorNetwork.create_actor_network   RL/MR_ddpg.py:120-137   5 -> FC64 -> BN -> ReLU -> FC64 -> BN -> ReLU -> FC2,
//                                                               tanh, * action_bound
//   ActorNetwork.predict                RL/MR_ddpg.py:145-148   inference: tflearn's batch_normalization runs on its
//                                                               moving statistics (the script never switches tflearn's
//                                                               training mode on), i.e. a per-feature affine map that
//                                                               the host folds into the preceding linear layer
//   OUNoise.__call__                    RL/MR_ddpg.py:69-73     x += theta (mu - x) dt + sigma sqrt(dt) N(0,1), mu = 0
//   the loop's use of both              RL/MR_ddpg.py:277       action = actor.predict(state) + actor_noise()
//
// Mapping onto the hardware.  One wave = 64 environments (one per lane for the env step).  The two 64-wide layers are the
// only dense contraction anywhere near this path and go to the matrix cores as f32-input MFMA (v_mfma_f32_32x32x2_f32:
// exact f32 products, f32 accumulate, bit-for-bit a k-ordered fmaf chain), features on the M rows, environments on the N
// columns: H[f][env] = sum_k W[f][k] X[k][env].  A wave's 64 envs are two column tiles (ct), the 64 features two row
// tiles (rt).  The 32x32 result layout puts the COLUMN (env) on the lane and 16 ROWS (features) in the lane's registers --
// exactly what the next layer's B operand wants (it sums over the feature index), so activations never leave the
// registers between layers: at k-step s lane half h supplies the feature it holds in register s, and the weights are
// stored pre-permuted to match (kperm below).  Only the 5-wide input and the 2-wide output cross lanes, with
// v_permlane32_swap.  The 64 -> 2 output layer is 2 x 32 fmas per lane on the vector unit (an MFMA tile would be 94 % padding).
//
// Summation order (part of the definition; oracle/mrsim_oracle.c: orc_actor_forward follows it, so the two agree bitwise):
//   kperm(q, h) = 32 (q / 16) + 8 ((q % 16) / 4) + 4 h + (q % 4),   q = 0..31, h = 0..1
//   layer 1:  acc = b1[f];  for k = 0..4:              acc = fmaf(W1[f][k], x[k], acc)        (k = 5 is a zero pad)
//   layer 2:  acc = b2[f];  for q = 0..31, h = 0..1:   acc = fmaf(W2[f][kperm(q,h)], relu(h1[kperm(q,h)]), acc)
//   layer 3:  p_h = 0;      for q = 0..31:             p_h = fmaf(W3[o][kperm(q,h)], relu(h2[kperm(q,h)]), p_h)
//             pre = (p_0 + p_1) + b3[o];   out[o] = tanh_spec(pre) * bound[o]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrsim {

constexpr int kActHidden = 64;
// Arithmetic of the two hidden layers (MrsimActor.math; in the bf16 arithmetics layer 1 runs on the bf16 matrix cores as well,
// see act_l1_slot_input / act_l1x3_slot_input below):
//   kActF32     exact f32 on v_mfma_f32_32x32x2_f32 (the definition above; bitwise against the oracle).  On gfx950 this
//               instruction runs at the f32 VECTOR rate and -- measured -- does not overlap the vector unit's own work: the
//               fused rollout takes (MFMA cycles + VALU cycles).
//   kActBf16x3  every f32 operand split into three bf16 terms x = x1 + x2 + x3 (round-to-nearest, 24 mantissa bits in all)
//               and the six products a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 run on v_mfma_f32_32x32x16_bf16 with f32
//               accumulation: bf16 x bf16 products are exact in f32, the dropped terms are below 2^-24 |a||b|, so the layer
//               keeps f32-class accuracy (tests: <= 5e-6 of the action bound against the F32 result and against the oracle's emulation of the same
//               splits, <= 1e-5 against PyTorch fp32) at 6/16 of the f32-MFMA cycles, on the matrix cores proper, which
//               DO run beside the vector unit.
//   kActBf16    plain bf16 operands (the first term alone), one MFMA per k-step and row tile, f32 accumulation, the OUTPUT layer
//               included (four more MFMAs per column tile instead of 32 v_max + 64 fma on the vector unit): what bf16 inference
//               is -- 1e-2-class accuracy of the action (tests: <= 6e-2 of the bound against PyTorch fp32 on nets whose output layer
//               saturates, 1e-4 on freshly initialised ones) for exploration-grade collection at twice the bf16x3 rate.
enum : int { kActOff = 0, kActF32 = 1, kActBf16x3 = 2, kActBf16 = 3 };
// packed parameter block (float offsets); the kernels copy the part their arithmetic needs to LDS once per block
constexpr int kActA1 = 0;                    // [rt 2][s 3][lane 64]      layer-1 A operands
constexpr int kActA2 = kActA1 + 2 * 3 * 64;  // [rt 2][s4 8][lane 64][4]  layer-2 A operands, four k-steps per ds_read_b128
constexpr int kActC1 = kActA2 + 64 * 64;     // [h 2][q 32]               b1 in accumulator layout
constexpr int kActC2 = kActC1 + 64;          // [h 2][q 32]               b2 in accumulator layout
constexpr int kActW3 = kActC2 + 64;          // [h 2][o 2][q 32]          output layer, per lane half
constexpr int kActTail = kActW3 + 128;       // b3[2], bound[2]
constexpr int kActF32Floats = kActTail + 8;  // end of the f32 section, padded to a multiple of 4 floats (copied as float4)
// bf16x3 section: layer-2 A operands [rt 2][s 4][part 3][lane 64][8 bf16]: one ds_read_b128 per (row tile, k-step, part)
constexpr int kActA2bf = kActF32Floats;
constexpr int kActA2bfFloats = 2 * 4 * 3 * 64 * 4;
constexpr int kActBlobFloats = kActF32Floats + kActA2bfFloats;
static_assert(kActF32Floats % 4 == 0 && kActBlobFloats % 4 == 0, "blob is copied as float4");

// LDS image per arithmetic (float offsets).  f32: the f32 section as it is.  bf16x3: the f32 section WITHOUT its 16 KiB of
// layer-2 operands, then the bf16 section (27 KiB instead of 43).
template <int MODE> struct ActLds;
template <> struct ActLds<kActF32> {
    static constexpr int A1 = kActA1, A1bf = 0, A2 = kActA2, C1 = kActC1, C2 = kActC2, W3 = kActW3, Tail = kActTail, A2bf = 0,
                         Floats = kActF32Floats;
};
template <> struct ActLds<kActBf16x3> {   // A1bf: [m 2][rt 2][lane 64][8 bf16], the layer-1 operands of its two MFMAs per tile (below)
    static constexpr int A1 = 0, A1bf = 0, A2 = 0, C1 = 1024, C2 = C1 + 64, W3 = C2 + 64, Tail = W3 + 128, A2bf = Tail + 8,
                         Floats = A2bf + kActA2bfFloats;
};
static_assert(ActLds<kActBf16x3>::A2bf % 4 == 0, "ds_read_b128 alignment");
// plain bf16: layer 1 runs on the bf16 matrix cores as well (below), so its A operands are a bf16 image built while staging
// (512 floats: [rt 2][lane 64][8 bf16]) in place of the 384 floats of f32 operands
// ... and so does the output layer: rows 0 and 1 of its 32-row A tile carry W3, the other 30 rows are zero, so the operands are a
// compact table built while staging from the f32 W3 block -- [k-step s 4][slot 5][8 bf16]: slot 2 h + i = row i of lane half h,
// slot 4 = eight zeros, which every lane of rows 2..31 reads (80 floats; a full [s][lane] image would be 4 KiB and cost the
// goal-table kernels their second block per compute unit)
template <> struct ActLds<kActBf16> {
    static constexpr int A1 = 0, A1bf = 0, A2 = 0, C1 = 512, C2 = C1 + 64, W3 = C2 + 64, Tail = W3 + 128, A2bf = Tail + 8,
                         A3bf = A2bf + kActA2bfFloats, Floats = A3bf + 4 * 5 * 4;
};
static_assert(ActLds<kActBf16>::A2bf % 4 == 0 && ActLds<kActBf16>::A3bf % 4 == 0, "ds_read_b128 alignment");
// Layer 1 of the plain-bf16 arithmetic: ONE v_mfma_f32_32x32x16_bf16 per (row tile, column tile) instead of three f32 MFMAs.
// Its 16 k-slots hold the five inputs as three bf16 terms each (hi, mid, lo: the input keeps its 24 bits) against the bf16-rounded
// weight of that input, repeated per term:  slot -> input  0 1 2 3 4 4 0 1 | 2 3 4 - 0 1 2 3   (terms: hi x5, lo of 4, mid 0 1 |
// mid 2 3 4, -, lo 0 1 2 3).  128 matrix cycles per wave and step instead of 768.
__host__ __device__ constexpr int act_l1_slot_input(int slot) { return (int)((0x3210F43210443210ull >> (4 * slot)) & 15ull); }  // 15: empty
// Layer 1 of the bf16 x 3 arithmetic: weights AND inputs as three bf16 terms, the six products above 2^-24 of each of the five
// inputs = 30 slots = TWO v_mfma_f32_32x32x16_bf16 per tile (256 matrix cycles per wave and step instead of 768), small terms first:
//   MFMA 1 (2^-16 terms)  slots 0-3 w1 x3 | 4-7 w2 x2 | 8-11 w3 x1 | 12-14 input 4: w1 x3, w2 x2, w3 x1 | 15 empty      (inputs 0..3 in 0-11)
//   MFMA 0                slots 0-3 w1 x1 | 4-7 w1 x2 | 8-11 w2 x1 | 12-14 input 4: w1 x1, w1 x2, w2 x1 | 15 empty
__host__ __device__ constexpr int act_l1x3_slot_input(int slot) { return (int)((0xF444321032103210ull >> (4 * slot)) & 15ull); }
__host__ __device__ constexpr int act_l1x3_slot_wterm(int m, int slot) {   // 0-based term of the weight in MFMA m's slot
    return (int)(((m ? 0x0210222211110000ull : 0x0100111100000000ull) >> (4 * slot)) & 15ull);
}
template <> struct ActLds<kActOff> {
    static constexpr int A1 = 0, A1bf = 0, A2 = 0, C1 = 0, C2 = 0, W3 = 0, Tail = 0, A2bf = 0, Floats = 4;
};

__host__ __device__ constexpr int act_kperm(int q, int h) { return 32 * (q / 16) + 8 * ((q % 16) / 4) + 4 * h + (q % 4); }

typedef float act_f32x16 __attribute__((ext_vector_type(16)));
typedef float act_f32x4 __attribute__((ext_vector_type(4)));

// copy the parameter block HBM -> LDS (all threads of the block; caller synchronises)
__device__ __forceinline__ void act_copy4(const float* __restrict__ src, float* __restrict__ dst, int nfloats, unsigned tid,
                                          unsigned nthreads) {
    const act_f32x4* __restrict__ s4 = reinterpret_cast<const act_f32x4*>(src);
    act_f32x4* __restrict__ d4 = reinterpret_cast<act_f32x4*>(dst);
    for (unsigned k = tid; k < (unsigned)(nfloats / 4); k += nthreads) d4[k] = s4[k];
}
template <int MODE>
__device__ __forceinline__ void actor_stage_blob(const float* __restrict__ blob, float* __restrict__ s_blob, unsigned tid,
                                                 unsigned nthreads) {
    using L = ActLds<MODE>;
    if constexpr (MODE == kActF32) {
        act_copy4(blob, s_blob, kActF32Floats, tid, nthreads);
    } else if constexpr (MODE == kActBf16x3) {
        // layer-1 A operands of the two MFMAs per tile: the bf16 terms of W1' (split here, as act_split3 splits the inputs)
        uint32_t* __restrict__ a1 = reinterpret_cast<uint32_t*>(s_blob + L::A1bf);
        for (unsigned d = tid; d < 2u * 2u * 64u * 4u; d += nthreads) {
            const unsigned m = d >> 9, rt = (d >> 8) & 1u, lane = (d >> 2) & 63u, pair = d & 3u;
            float w[2];
            int term[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int slot = (int)(8u * (lane >> 5) + 2u * pair) + e;
                const int i = act_l1x3_slot_input(slot);
                term[e] = act_l1x3_slot_wterm((int)m, slot);
                w[e] = i == 15 ? 0.0f : blob[kActA1 + (rt * 3 + (unsigned)(i >> 1)) * 64 + (lane & 31u) + 32u * (unsigned)(i & 1)];
            }
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 v = {w[0], w[1]};
            const bf2 t0 = __builtin_convertvector(v, bf2);
            const f2 r1 = v - __builtin_convertvector(t0, f2);
            const bf2 t1 = __builtin_convertvector(r1, bf2);
            const f2 r2 = r1 - __builtin_convertvector(t1, f2);
            const bf2 t2 = __builtin_convertvector(r2, bf2);
            bf2 o;
#pragma unroll
            for (int e = 0; e < 2; ++e) o[e] = term[e] == 0 ? t0[e] : (term[e] == 1 ? t1[e] : t2[e]);
            a1[d] = __builtin_bit_cast(uint32_t, o);
        }
        act_copy4(blob + kActC1, s_blob + L::C1, kActF32Floats - kActC1, tid, nthreads);      // biases, output layer, tail
        act_copy4(blob + kActA2bf, s_blob + L::A2bf, kActA2bfFloats, tid, nthreads);          // layer-2 bf16 terms
    } else if constexpr (MODE == kActBf16) {
        // layer-1 A operands, rounded to bf16 here (v_cvt_pk_bf16_f32: nearest even, as the host rounds the layer-2 terms):
        // lane (j, h) of row tile rt holds W1'[32 rt + j][input of slot 8 h + jj], jj = 0..7; one dword = two slots.
        // W1'[f][i] sits in the f32 section at [rt][s = i / 2][lane = (f & 31) + 32 (i & 1)].
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        uint32_t* __restrict__ a1 = reinterpret_cast<uint32_t*>(s_blob + L::A1bf);
        for (unsigned d = tid; d < 2u * 64u * 4u; d += nthreads) {
            const unsigned rt = d >> 8, lane = (d >> 2) & 63u, pair = d & 3u;
            f2 w;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = act_l1_slot_input((int)(8u * (lane >> 5) + 2u * pair) + e);
                w[e] = i == 15 ? 0.0f : blob[kActA1 + (rt * 3 + (unsigned)(i >> 1)) * 64 + (lane & 31u) + 32u * (unsigned)(i & 1)];
            }
            const bf2 t = __builtin_convertvector(w, bf2);
            a1[d] = __builtin_bit_cast(uint32_t, t);
        }
        // output-layer A operands: lane (i, h) of k-step s holds W3[i][kperm(8 s + jj, h)], jj = 0..7, for the two output rows
        // i = 0, 1 (the f32 block stores W3[o][kperm(q, h)] at [h][o][q]) -- slot 2 h + i of the table; slot 4 is the zero row
        uint32_t* __restrict__ a3 = reinterpret_cast<uint32_t*>(s_blob + L::A3bf);
        for (unsigned d = tid; d < 4u * 5u * 4u; d += nthreads) {
            const unsigned sx = d / 20u, slot = (d >> 2) % 5u, pair = d & 3u;
            f2 w = {0.0f, 0.0f};
            if (slot < 4u) {
                w[0] = blob[kActW3 + slot * 32u + 8u * sx + 2u * pair];          // ([h][o][q] with slot = 2 h + o)
                w[1] = blob[kActW3 + slot * 32u + 8u * sx + 2u * pair + 1u];
            }
            const bf2 t = __builtin_convertvector(w, bf2);
            a3[d] = __builtin_bit_cast(uint32_t, t);
        }
        act_copy4(blob + kActC1, s_blob + L::C1, kActF32Floats - kActC1, tid, nthreads);      // biases, output layer, tail
        act_copy4(blob + kActA2bf, s_blob + L::A2bf, kActA2bfFloats, tid, nthreads);          // layer-2 bf16 terms
    }
}

// exp(x) for x in [-20, 20]: Cephes expf (Cody-Waite reduction by ln 2, degree-5 polynomial), explicit fmaf, exact ldexp.
// The oracle has the same operations in the same order.
__device__ __forceinline__ float spec_expf(float x) {
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = __builtin_fmaf(p, r, 1.3981999507E-3f);
    p = __builtin_fmaf(p, r, 8.3334519073E-3f);
    p = __builtin_fmaf(p, r, 4.1665795894E-2f);
    p = __builtin_fmaf(p, r, 1.6666665459E-1f);
    p = __builtin_fmaf(p, r, 5.0000001201E-1f);
    const float z = r * r;
    const float e = __builtin_fmaf(p, z, r) + 1.0f;
    return __builtin_ldexpf(e, (int)n);
}

// tanh in specified fp32 arithmetic (Cephes tanhf): odd polynomial below 0.625, 1 - 2 / (exp(2|x|) + 1) above,
// +-1 from 9.0 on.  The division is IEEE (hipcc's default for `/` on float): same bits as the oracle's.
__device__ __forceinline__ float spec_tanhf(float x) {
    const float ax = __builtin_fabsf(x);
    float r;
    if (ax >= 9.0f) {
        r = 1.0f;
    } else if (ax >= 0.625f) {
        const float e = spec_expf(ax + ax);
        r = 1.0f - 2.0f / (e + 1.0f);
    } else {
        const float z = x * x;
        float p = -5.70498872745E-3f;
        p = __builtin_fmaf(p, z, 2.06390887954E-2f);
        p = __builtin_fmaf(p, z, -5.37397155531E-2f);
        p = __builtin_fmaf(p, z, 1.33314422036E-1f);
        p = __builtin_fmaf(p, z, -3.33332819422E-1f);
        return __builtin_fmaf(p * z, x, x);
    }
    return x < 0.0f ? -r : r;
}

// tanh for the plain-bf16 arithmetic: 1 - 2 / (exp(2x) + 1) on the hardware's exp2 and reciprocal (1 ulp each), branch-free, six
// instructions instead of the specified routine's ~42 per value.  Absolute error ~1e-7 -- four orders below what rounding the
// operands to bf16 does to the action; +-1 at the ends (exp2 -> inf / 0).  The exact arithmetics keep spec_tanhf (bitwise vs the oracle).
__device__ __forceinline__ float fast_tanhf(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);      // exp(2 x)
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

__device__ __forceinline__ void lds_load16(const float* __restrict__ p, act_f32x16& v) {
    const act_f32x4* __restrict__ q = reinterpret_cast<const act_f32x4*>(p);
    const act_f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
    v = act_f32x16{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
}

// ReLU as ONE instruction: v_max_i32 on the float's bits.  Non-negative floats order like their bit patterns and every
// negative float (-0 included) is a negative integer, so max(bits, 0) IS max(x, 0) for every non-NaN x, with no
// canonicalisation step.  (__builtin_fmaxf and v_med3_f32(x, 0, inf) both compile to TWO v_max_f32: canonicalise, then
// max -- 256 of the kernel's 745 vector instructions per wave and step.  An inline-asm v_max_f32 is NOT an option: the hazard
// recogniser cannot see into asm, does not pad the MFMA-result -> VALU-read wait states, and the read returns the
// accumulator's old value -- measured: actions off by 5e-4 in one kernel instantiation and right in another.)
// NaN handling differs from fmaxf(x, 0): a NaN with the sign bit set becomes 0, one without stays NaN (its bits compare above
// zero).  Finite weights and observations never produce one; include/mrsim.h says so at MrsimActorWeights.
__device__ __forceinline__ float act_relu(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}

typedef __bf16 act_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 act_bf16x2 __attribute__((ext_vector_type(2)));
typedef float act_f32x2 __attribute__((ext_vector_type(2)));

// x = x1 + x2 + x3, each term a bf16 (v_cvt_pk_bf16_f32: round to nearest even), two values at a time.  The residuals are
// exact in f32 (x1 keeps x's leading 8 bits: x - x1 has at most 16 significant bits left).
__device__ __forceinline__ void act_split3(float x0, float x1, act_bf16x2 (&t)[3]) {
    const act_f32x2 v = {x0, x1};
    t[0] = __builtin_convertvector(v, act_bf16x2);
    const act_f32x2 r1 = v - __builtin_convertvector(t[0], act_f32x2);
    t[1] = __builtin_convertvector(r1, act_bf16x2);
    const act_f32x2 r2 = r1 - __builtin_convertvector(t[1], act_f32x2);
    t[2] = __builtin_convertvector(r2, act_bf16x2);
}

// actor.predict for the 64 envs of this wave.  obs: this lane's env's observation; a[2]: scaled_out of this lane's env.
// sA: the LDS image of the parameter block for this arithmetic (actor_stage_blob<MODE>).  Must be called by ALL 64 lanes in
// uniform control flow (MFMA and v_permlane32_swap are wave-wide operations); lanes without an env pass any finite values.
template <int MODE>
__device__ __forceinline__ void actor_forward(const float* __restrict__ sA, const float (&obs)[5], float (&a)[2]) {
    using L = ActLds<MODE>;
    // layer-1 B operands: at k-step s lane (j, h) of column tile ct supplies obs[2 s + h] of env 32 ct + j.  One half swap
    // per k-step pair turns "lane = env" registers into both tiles' operands: {x[2s].lo | x[2s+1].lo}, {x[2s].hi | x[2s+1].hi}.
    float b1op[2][3];
    uint32_t b1bf[2][4];   // plain bf16: the 16 k-slots of one column tile's B operand (two bf16 per register)
    uint32_t b1x3[2][2][4];  // bf16 x 3: [MFMA m][column tile][register]
    if constexpr (MODE == kActBf16x3) {
        // slots of MFMA 0:  half 0 {x1_0 x1_1 | x1_2 x1_3 | x2_0 x2_1 | x2_2 x2_3}   half 1 {x1_0 x1_1 | x1_2 x1_3 | x1_4 x2_4 | x1_4 0}
        //          MFMA 1:  half 0 {x3_0 x3_1 | x3_2 x3_3 | x2_0 x2_1 | x2_2 x2_3}   half 1 {x1_0 x1_1 | x1_2 x1_3 | x3_4 x2_4 | x1_4 0}
        act_bf16x2 t01[3], t23[3], t4[3];
        act_split3(obs[0], obs[1], t01);
        act_split3(obs[2], obs[3], t23);
        act_split3(obs[4], 0.0f, t4);
        auto u = [](const act_bf16x2& v) { return __builtin_bit_cast(uint32_t, v); };
        const uint32_t p0[2][4] = {{u(t01[0]), u(t23[0]), u(t01[1]), u(t23[1])}, {u(t01[2]), u(t23[2]), u(t01[1]), u(t23[1])}};
        const uint32_t p1[2][4] = {{u(t01[0]), u(t23[0]), u(t4[0]) | (u(t4[1]) << 16), u(t4[0])},
                                   {u(t01[0]), u(t23[0]), u(t4[2]) | (u(t4[1]) << 16), u(t4[0])}};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const auto r = __builtin_amdgcn_permlane32_swap(p0[m][d], p1[m][d], false, false);
                b1x3[m][0][d] = r[0];
                b1x3[m][1][d] = r[1];
            }
        }
    } else if constexpr (MODE != kActBf16) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const float va = obs[2 * s];
            const float vb = (2 * s + 1 < 5) ? obs[2 * s + 1] : 0.0f;
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
            b1op[0][s] = __uint_as_float(r[0]);
            b1op[1][s] = __uint_as_float(r[1]);
        }
    } else {
        // this env's five inputs as three bf16 terms each, laid into the slots of act_l1_slot_input: lane half 0 supplies
        // slots 0..7 = {hi0 hi1 | hi2 hi3 | hi4 lo4 | mid0 mid1}, half 1 slots 8..15 = {mid2 mid3 | mid4 0 | lo0 lo1 | lo2 lo3};
        // one half swap per register turns "lane = env" into both column tiles' operands, as above
        act_bf16x2 t01[3], t23[3], t4[3];
        act_split3(obs[0], obs[1], t01);
        act_split3(obs[2], obs[3], t23);
        act_split3(obs[4], 0.0f, t4);
        auto u = [](const act_bf16x2& v) { return __builtin_bit_cast(uint32_t, v); };
        const uint32_t p0[4] = {u(t01[0]), u(t23[0]), u(t4[0]) | (u(t4[2]) << 16), u(t01[1])};
        const uint32_t p1[4] = {u(t23[1]), u(t4[1]), u(t01[2]), u(t23[2])};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const auto r = __builtin_amdgcn_permlane32_swap(p0[d], p1[d], false, false);
            b1bf[0][d] = r[0];
            b1bf[1][d] = r[1];
        }
    }
    // The two column tiles one after the other, as a real loop: unrolled, the scheduler interleaves them and keeps all
    // eight 16-register accumulators live (313 registers, one wave per SIMD); rolled, a tile needs four of them.
    float part[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma nounroll
    for (int ct = 0; ct < 2; ++ct) {
        float bop[3] = {0.f, 0.f, 0.f};
        if constexpr (MODE == kActF32) {
#pragma unroll
            for (int s = 0; s < 3; ++s) bop[s] = ct ? b1op[1][s] : b1op[0][s];
        }
        // Every LDS address below derives from a lane id that is opaque per iteration.  Otherwise the parameter reads -- the
        // same for both tiles and for every time step of a fused rollout -- are hoisted out of all loops and 260 registers
        // of weights and bias tiles stay live across the env step (measured: 313 registers, one wave per SIMD, or spills).
        unsigned lane = threadIdx.x & 63u;
        asm volatile("" : "+v"(lane));
        const unsigned h = lane >> 5;
        act_f32x16 acc1[2], acc2[2];
        typedef uint32_t act_u32x4 __attribute__((ext_vector_type(4)));
        if constexpr (MODE == kActF32) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                lds_load16(sA + L::C1 + h * 32 + rt * 16, acc1[rt]);
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    acc1[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[L::A1 + (rt * 3 + s) * 64 + lane], bop[s], acc1[rt], 0, 0, 0);
            }
        } else if constexpr (MODE == kActBf16x3) {
            act_bf16x8 bpm[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const act_u32x4 bw = {ct ? b1x3[m][1][0] : b1x3[m][0][0], ct ? b1x3[m][1][1] : b1x3[m][0][1],
                                      ct ? b1x3[m][1][2] : b1x3[m][0][2], ct ? b1x3[m][1][3] : b1x3[m][0][3]};
                bpm[m] = __builtin_bit_cast(act_bf16x8, bw);
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                lds_load16(sA + L::C1 + h * 32 + rt * 16, acc1[rt]);
#pragma unroll
                for (int m = 1; m >= 0; --m) {   // the 2^-16 terms first
                    const act_bf16x8 ap1 = *reinterpret_cast<const act_bf16x8*>(sA + L::A1bf + ((m * 2 + rt) * 64 + lane) * 4);
                    acc1[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap1, bpm[m], acc1[rt], 0, 0, 0);
                }
            }
        } else {
            const act_u32x4 bw = {ct ? b1bf[1][0] : b1bf[0][0], ct ? b1bf[1][1] : b1bf[0][1], ct ? b1bf[1][2] : b1bf[0][2],
                                  ct ? b1bf[1][3] : b1bf[0][3]};
            const act_bf16x8 bp1 = __builtin_bit_cast(act_bf16x8, bw);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                lds_load16(sA + L::C1 + h * 32 + rt * 16, acc1[rt]);
                const act_bf16x8 ap1 = *reinterpret_cast<const act_bf16x8*>(sA + L::A1bf + (rt * 64 + lane) * 4);
                acc1[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap1, bp1, acc1[rt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) lds_load16(sA + L::C2 + h * 32 + rt * 16, acc2[rt]);
        if constexpr (MODE == kActF32) {
#ifndef MRSIM_ACTOR_PREFETCH   // where the A operands of k-step group s4 + 1 are requested: 0 = right behind the eight MFMAs of group
#define MRSIM_ACTOR_PREFETCH 0 // s4, 1 = in front of them (two groups live), 2 = in front of their own MFMAs.  Built from one source and
#endif                         // run on one box the three are equal (1010 - 1016 us per launch, profiles/r03/ab_actor_prefetch.txt); an
                               // earlier A/B that had 4.5 % between two of them did not reproduce: box-to-box and build-to-build
                               // (code placement) differences of this kernel are of that size.
            act_f32x4 a4[2], a4n[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) a4[rt] = *reinterpret_cast<const act_f32x4*>(sA + L::A2 + ((rt * 8 + 0) * 64 + lane) * 4);
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                if constexpr (MRSIM_ACTOR_PREFETCH == 2) {   // as first written: each group's operands requested right in front of it
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        a4[rt] = *reinterpret_cast<const act_f32x4*>(sA + L::A2 + ((rt * 8 + s4) * 64 + lane) * 4);
                }
                if constexpr (MRSIM_ACTOR_PREFETCH == 1) {
                    if (s4 + 1 < 8) {
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
                            a4n[rt] = *reinterpret_cast<const act_f32x4*>(sA + L::A2 + ((rt * 8 + s4 + 1) * 64 + lane) * 4);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = 4 * s4 + j;
                    const float b = act_relu(acc1[q / 16][q % 16]);  // ReLU of layer 1
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rt][j], b, acc2[rt], 0, 0, 0);
                }
                if (s4 + 1 < 8 && MRSIM_ACTOR_PREFETCH != 2) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        if constexpr (MRSIM_ACTOR_PREFETCH == 1) a4[rt] = a4n[rt];
                        else a4[rt] = *reinterpret_cast<const act_f32x4*>(sA + L::A2 + ((rt * 8 + s4 + 1) * 64 + lane) * 4);
                    }
                }
            }
        } else if constexpr (MODE == kActBf16) {
            // plain bf16: the activations rounded once (v_cvt_pk_bf16_f32), the weights' first term, one MFMA per (k-step, row tile)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                act_bf16x8 bp;
#pragma unroll
                for (int jj = 0; jj < 8; jj += 2) {
                    const int q = 8 * s + jj;
                    // ReLU AFTER the rounding, on both halves at once: a bf16 is sign-magnitude, so as an int16 a negative value
                    // (or -0) is < 0 and max(., 0) is the ReLU -- v_pk_max_i16 instead of two v_max_f32 in front of the conversion.
                    // Same bits: rounding to nearest keeps the sign, and a value that rounds to -0 becomes +0 either way.
                    const act_f32x2 v = {acc1[q / 16][q % 16], acc1[(q + 1) / 16][(q + 1) % 16]};
                    typedef short act_s16x2 __attribute__((ext_vector_type(2)));
                    const act_s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(act_s16x2, __builtin_convertvector(v, act_bf16x2)),
                                                                  (act_s16x2){0, 0});
                    const act_bf16x2 t = __builtin_bit_cast(act_bf16x2, r);
                    bp[jj] = t[0]; bp[jj + 1] = t[1];
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const act_bf16x8 ap = *reinterpret_cast<const act_bf16x8*>(sA + L::A2bf + (((rt * 4 + s) * 3 + 0) * 64 + lane) * 4);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap, bp, acc2[rt], 0, 0, 0);
                }
            }
        } else {
            // bf16 x 3: k-step s of the 32x32x16 instruction sums 16 features -- lane half h supplies its registers 8 s .. 8 s + 7,
            // i.e. features kperm(8 s + jj, h), and the weights are stored pre-permuted and pre-split to match.
            // (operands of the NEXT (k-step, row tile) are requested right behind the six MFMAs of the current one, as in the f32 path)
            act_bf16x8 ap[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) ap[p] = *reinterpret_cast<const act_bf16x8*>(sA + L::A2bf + ((0 * 3 + p) * 64 + lane) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // the three bf16 terms of this k-step's eight activations, split right before their MFMAs (12 registers live
                // instead of 48 for the whole tile)
                act_bf16x8 bp[3];
#pragma unroll
                for (int jj = 0; jj < 8; jj += 2) {
                    const int q = 8 * s + jj;
                    act_bf16x2 t[3];
                    act_split3(act_relu(acc1[q / 16][q % 16]), act_relu(acc1[(q + 1) / 16][(q + 1) % 16]), t);
#pragma unroll
                    for (int p = 0; p < 3; ++p) { bp[p][jj] = t[p][0]; bp[p][jj + 1] = t[p][1]; }
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    // smallest terms first: a3 b1, a2 b2, a1 b3 (2^-16), a2 b1, a1 b2 (2^-8), a1 b1
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], bp[0], acc2[rt], 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[1], acc2[rt], 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[2], acc2[rt], 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[0], acc2[rt], 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[1], acc2[rt], 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[0], acc2[rt], 0, 0, 0);
                    const int nxt = (rt == 0) ? ((1 * 4 + s) * 3) : ((0 * 4 + s + 1) * 3);   // (rt, s) -> ((rt * 4 + s) * 3 + part)
                    if (rt == 0 || s + 1 < 4) {
#pragma unroll
                        for (int p = 0; p < 3; ++p)
                            ap[p] = *reinterpret_cast<const act_bf16x8*>(sA + L::A2bf + ((nxt + p) * 64 + lane) * 4);
                    }
                }
            }
        }
        float p0 = 0.0f, p1 = 0.0f;
        if constexpr (MODE == kActBf16) {
            // output layer on the matrix cores too: out[o][env] = sum_f W3[o][f] relu(h2[f][env]) as four MFMAs of a 32-row tile whose
            // rows 0, 1 are W3 (bf16) -- the activations rounded once, as between layers 1 and 2.  Row o of the result sits in
            // register o of the lanes of half 0 (half 1 holds rows 4, 5: zeros), i.e. exactly the "partial sums per half" the
            // epilogue below adds.  64 v_max + 128 fma per wave and step become 32 conversions + 32 packed max + 8 MFMAs.
            // (The first MFMA's C is the inline constant 0: the accumulator is born there.)
            const unsigned a3slot = (lane & 31u) < 2u ? 2u * h + (lane & 31u) : 4u;   // rows 2..31 of the tile: the zero row
            act_f32x16 acc3 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                act_bf16x8 bp;
#pragma unroll
                for (int jj = 0; jj < 8; jj += 2) {
                    const int q = 8 * s + jj;
                    const act_f32x2 v = {acc2[q / 16][q % 16], acc2[(q + 1) / 16][(q + 1) % 16]};
                    typedef short act_s16x2 __attribute__((ext_vector_type(2)));
                    const act_s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(act_s16x2, __builtin_convertvector(v, act_bf16x2)),
                                                                  (act_s16x2){0, 0});
                    const act_bf16x2 t = __builtin_bit_cast(act_bf16x2, r);
                    bp[jj] = t[0]; bp[jj + 1] = t[1];
                }
                const act_bf16x8 ap = *reinterpret_cast<const act_bf16x8*>(sA + L::A3bf + (s * 5 + a3slot) * 4);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap, bp, acc3, 0, 0, 0);
            }
            p0 = acc3[0]; p1 = acc3[1];
        } else {
        // output layer on the vector unit: this lane's 32 features of env (32 ct + j), two partial sums per output
#pragma unroll
        for (int q4 = 0; q4 < 8; ++q4) {
            const act_f32x4 w0 = *reinterpret_cast<const act_f32x4*>(sA + L::W3 + (h * 2 + 0) * 32 + q4 * 4);
            const act_f32x4 w1 = *reinterpret_cast<const act_f32x4*>(sA + L::W3 + (h * 2 + 1) * 32 + q4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = 4 * q4 + j;
                const float hq = act_relu(acc2[q / 16][q % 16]);
                p0 = __builtin_fmaf(w0[j], hq, p0);
                p1 = __builtin_fmaf(w1[j], hq, p1);
            }
        }
        }
        if (ct == 0) { part[0][0] = p0; part[0][1] = p1; }
        else { part[1][0] = p0; part[1][1] = p1; }
    }
    // lane (j, h) holds half h's partial sums of env j (ct 0) and env 32 + j (ct 1): one half swap per output puts
    // {half 0's | half 1's} partial of env = lane into the two result registers
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[0][o]), __float_as_uint(part[1][o]), false, false);
        const float pre = (__uint_as_float(r[0]) + __uint_as_float(r[1])) + sA[L::Tail + o];
        a[o] = (MODE == kActBf16 ? fast_tanhf(pre) : spec_tanhf(pre)) * sA[L::Tail + 2 + o];
    }
}

// OUNoise.__call__ (RL/MR_ddpg.py:69-73) with mu = 0, in fp32:  x <- fma(sigma sqrt(dt), z, fma(-theta dt, x, x))
struct OUParams {
    float theta_dt, sigma_sqrt_dt;
};
__device__ __forceinline__ float ou_update(const OUParams& ou, float x, float z) {
    return __builtin_fmaf(ou.sigma_sqrt_dt, z, __builtin_fmaf(-ou.theta_dt, x, x));
}

}  // namespace mrsim
