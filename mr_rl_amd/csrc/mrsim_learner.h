// mrsim_learner.h -- the DDPG learner update of RL/MR_ddpg.py:288-305 as ONE kernel for gfx950.
//
//   sample -> y = r + gamma Q'(s2, mu'(s2)) (1 - done)            :290-294   (target networks)
//          -> critic: minimise mean (y - Q(s, a))^2, Adam          :297       CriticNetwork.train
//          -> actor: ascend mean Q(s, mu(s)) through dQ/da, Adam   :300-302   action_gradients -> ActorNetwork.train
//          -> both targets <- tau online + (1 - tau) target        :305-306
// for the networks of :120-137 (actor 5-64-64-2) and :207-223 (critic 5-64-(32 + action)-1).  Batch normalisation is the
// fixed affine map it is in the script (tflearn.is_training is never switched on: moving statistics, trainable gamma / beta).
//
// One workgroup of 256 threads does the whole update for a tile of 64 samples at a time (the reference's batch is 64; larger
// batches loop over tiles and accumulate the gradients): weights and activations live in LDS, the 64 x 64 and 32 x 64 products
// are register-tiled outer-product loops (4 x 4 / 2 x 4 outputs per thread, both operands read as 16-byte LDS vectors), the
// narrow layers (5 inputs, 2 / 1 outputs) and the per-feature reductions are plain loops, Adam and the soft update of the target
// are one pass over the parameter vector.  The update is latency-bound (6 MFLOP): what matters is that it is ONE launch
// with no host round trip instead of ~150 (3.2 ms eager, 0.55 ms as a captured graph: tools/learner_probe.py).
//
// Parameter vector (floats), online / target / Adam m / Adam v all in this layout (= the nn.Module tensors of mr_rl_amd/ddpg.py,
// row-major, so the host side can alias them):
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mrsim_device.h"   // philox4x32_10
#include "mrsim_actor.h"    // the packed actor block's layout

namespace mrsim {
namespace learner {

constexpr int A_W1 = 0, A_B1 = 320, A_G1 = 384, A_BE1 = 448, A_W2 = 512, A_B2 = 4608, A_G2 = 4672, A_BE2 = 4736, A_W3 = 4800,
              A_B3 = 4928, A_END = 4930;
constexpr int C_W1 = 4932, C_B1 = 5252, C_G1 = 5316, C_BE1 = 5380, C_T1 = 5444, C_T2 = 7492, C_BT2 = 7556, C_WO = 7588, C_BO = 7620,
              C_END = 7621;
constexpr int kParams = 7680;   // padded (two unused floats between the networks keep T1's rows 16-byte aligned)
constexpr int kTile = 64;       // samples per pass
constexpr int kThreads = 256;
constexpr int kMaxBatch = 4096;
static_assert(A_W2 % 4 == 0 && C_T1 % 4 == 0 && kParams % 4 == 0, "16-byte aligned rows");

struct Args {
    float* online; float* target; float* adam_m; float* adam_v; float* grad;   // [kParams] each (grad: scratch)
    int32_t* steps;                 // [2] Adam step counts: critic, actor
    const float* bn;                // [2 online/target][3 layers: actor bn1, actor bn2, critic bn1][2 mean/var][64]
    const float* s; const float* a; const float* r; const float* d; const float* s2;   // ring arrays [.][5], [.][2], [.], [.], [.][5]
    const int32_t* idx;             // [batch] rows of the ring arrays, or null: sampled in-kernel (ring_count > 0) / rows 0 .. batch-1
    int32_t* idx_out;               // optional [batch]: the rows used
    int32_t ring_count;             // > 0 and idx == null: draw the rows from [0, ring_count) here (Philox4x32-10, key = seed,
    uint32_t seed_lo, seed_hi, ctr_lo, ctr_hi;   //   counter = (draw, round, update counter)): without replacement up to 256 rows
    float* losses;                  // [2] critic loss, actor loss (device), or null
    int32_t batch;
    int32_t n_updates;              // consecutive updates in this one launch (each draws its own rows: draw counter + i)
    float bn_eps, gamma, tau, actor_lr, critic_lr, beta1, beta2, adam_eps;
    float bound0, bound1;
    // multi-workgroup form (batch > 64 with MrsimDdpgLearner.batch_scratch): one workgroup per tile of 64 samples
    float* partial;                 // [tiles][kParams]: every workgroup's gradients of its tile
    int32_t* rows_scratch;          // [batch]: the rows of this update (critic half -> actor half)
    float* loss_partial;            // [tiles][2]
    unsigned int* counter;          // [2]: arrival tickets of the two halves (zero between launches)
    int32_t reduce_in_kernel;       // 1: the last workgroup to arrive reduces and steps (few tiles); 0: mr_ddpg_mw_step_kernel does
    float* pack_blob;               // optional: the online actor folded / packed into this block after the last update of the launch
    float pack_scale[5];            //   (single-workgroup kernel: its tail; multi-workgroup form: mr_actor_pack_kernel follows)
};

// C[m0 .. m0+TM)[n0 .. n0+TN) = sum_k A(m, k) B(k, n).  B is k-major ([K][ldb], n contiguous).  A is k-major ([K][lda], m
// contiguous) or, AT, m-major ([M][lda], k contiguous: the row-major weight matrices and sample-major activations as they are).
template <int N>
__device__ __forceinline__ void ldvec(const float* __restrict__ p, float (&v)[N]) {   // one 16- / 8-byte LDS read (p is so aligned)
    if constexpr (N == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        static_assert(N == 2, "tile widths are 2 or 4");
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
    }
}

// The 64 x 64 (x 64 / 32) and 32 x 64 products of the update run on the matrix cores: v_mfma_f32_16x16x4_f32, laid out so that every
// lane ends up with EXACTLY the TM x 4 register tile the surrounding code works on (rows m0 = ty TM .., columns n0 = 4 tx ..;
// tid = 16 ty + tx, so lane = 16 (ty & 3) + tx):
//   * a wave owns the 4 TM rows of its four ty values and all 64 columns, as four 16 x 16 MFMA tiles t = 0..3; MFMA tile t's column
//     j stands for logical column 4 j + t, i.e. the B operand of lane (k-group gk = lane >> 4, j = lane & 15) for the four tiles
//     is ONE 16-byte LDS read B[k][4 j .. 4 j + 3] -- the read the register-tiled loop did;
//   * the D operand of lane (g = lane >> 4, j) holds rows 4 g + r of tile t: acc[r][t] = tile t, register r = the thread tile
//     (TM = 2: only r < 2 carry rows, the A lanes of the other two feed zeros -- half the matrix work is idle there);
//   * step s of a 16-deep k block contracts k = 16 b + 4 gk + s: a row-major A (AT) is read 16 bytes per lane and block.
// Per wave and 64 x 64 x 64 product: 64 MFMAs (2 048 cycles) and 20 sixteen-byte LDS reads -- the register-tiled loop issued 128
// such reads per lane and was bound by the LDS return path (profiles/r04/NOTES.md section 6).  Accumulation order differs from a
// sequential fma chain (four products per instruction): parity with the PyTorch twin is within the tolerance the test states.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TM, int TN, int K, bool AT>
__device__ __forceinline__ void lgemm(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, int m0, int n0,
                                      float (&acc)[TM][TN]) {
    static_assert(TN == 4 && (TM == 4 || TM == 2) && K % 16 == 0, "thread tiles are 4 x 4 or 2 x 4, K a multiple of 16");
    const int lane = (int)threadIdx.x & 63, gk = lane >> 4, i = lane & 15;
    const int mw = m0 - gk * TM;                                   // first row of the wave (lane >> 4 == ty & 3)
    // the logical row this lane feeds as MFMA row i (D row i belongs to lane group i >> 2, register i & 3)
    const bool a_on = TM == 4 || (i & 3) < 2;
    const int a_row = TM == 4 ? mw + i : mw + 2 * (i >> 2) + (i & 1);
    f32x4 c[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) c[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < K / 16; ++b) {
        const int k0 = 16 * b + 4 * gk;
        float av[4];
        if constexpr (AT) {
            ldvec<4>(A + a_row * lda + k0, av);
        } else {
#pragma unroll
            for (int st = 0; st < 4; ++st) av[st] = A[(k0 + st) * lda + a_row];
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            float bv[4];
            ldvec<4>(B + (k0 + st) * ldb + n0, bv);
            const float a = a_on ? av[st] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[t], c[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int u = 0; u < TM; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = c[v][u];
}

// sum over the 64 lanes of a wave (every lane active), result in every lane
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// b^t in double by repeated squaring (t >= 1): ~2 log2 t multiplications; the library pow() is thousands of cycles on the one
// lane that needs it while 255 wait
__device__ __forceinline__ double ipow(double b, int t) {
    double r = 1.0;
    for (unsigned e = (unsigned)t; e != 0u; e >>= 1) {
        if (e & 1u) r *= b;
        b *= b;
    }
    return r;
}

struct Lds {
    float P[kParams];          // the staged network pair (target, then online)
    float rstd[3][64], mean[3][64], var[3][64];   // (var: the raw moving variances, for the policy upload's double-precision fold)
    float s[kTile * 5], s2[kTile * 5], a[kTile * 2], ap[kTile * 2], th[kTile * 2], dz3[kTile * 2], r[kTile], d[kTile], dq[kTile];
    float y[kMaxBatch];
    float X[5][kTile * 64];
    float loss[2];
    float bc[4];               // Adam bias corrections: critic (1 - b1^t, sqrt(1 - b2^t)), actor
    int rows[kTile];
    alignas(16) int sel[kMaxBatch];   // rows drawn in-kernel (read 16 bytes at a time by the duplicate check)
};

// parameters [p0, kParams) of `src` into L.P by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no VGPRs; the
// image is lane-linear -- a wave-instruction fills 1 KiB at its first lane's address -- and every instruction of the copy is in
// flight before the wait the next __syncthreads() brings).  As a load + ds_write loop the copy took one L2 round trip per pass.
__device__ __forceinline__ void stage_range(Lds& L, const float* __restrict__ src, int p0, int tid) {
    for (int p = p0 + tid * 4; p < kParams; p += kThreads * 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p),
                                         (__attribute__((address_space(3))) void*)(&L.P[p]), 16, 0, 0);
}

__device__ __forceinline__ void stage_params(Lds& L, const float* __restrict__ src, const float* __restrict__ bn, float eps, int tid) {
    stage_range(L, src, 0, tid);
    for (int q = tid; q < 3 * 64; q += kThreads) {
        const int l = q >> 6, k = q & 63;
        L.mean[l][k] = bn[(l * 2 + 0) * 64 + k];
        const float vr = bn[(l * 2 + 1) * 64 + k];
        L.var[l][k] = vr;
        L.rstd[l][k] = 1.0f / sqrtf(vr + eps);
    }
}

// hidden layer 1 (5 inputs) of either network for this tile: h[f][i] = relu(g (W x_i + b - mean) rstd + be).
// FM: feature-major [64][64 samples] (the B operand of the next layer's product); SM: sample-major [64 samples][64].
__device__ __forceinline__ void layer1(const Lds& L, const float* __restrict__ x, int W, int Bb, int G, int BE, int bn_l,
                                       float* __restrict__ FM, float* __restrict__ SM, int tid) {
    const int tx = tid & 15, ty = tid >> 4, f0 = ty * 4, i0 = tx * 4;
    float h[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int f = f0 + u;
        const float* w = &L.P[W + f * 5];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float* xi = x + (i0 + v) * 5;
            float z = L.P[Bb + f];
#pragma unroll
            for (int j = 0; j < 5; ++j) z = __builtin_fmaf(w[j], xi[j], z);
            const float n = __builtin_fmaf(L.P[G + f], (z - L.mean[bn_l][f]) * L.rstd[bn_l][f], L.P[BE + f]);
            h[u][v] = n > 0.f ? n : 0.f;
        }
    }
    if (FM != nullptr)
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(FM + (f0 + u) * 64 + i0) = make_float4(h[u][0], h[u][1], h[u][2], h[u][3]);
    if (SM != nullptr)
#pragma unroll
        for (int v = 0; v < 4; ++v) *reinterpret_cast<float4*>(SM + (i0 + v) * 64 + f0) = make_float4(h[0][v], h[1][v], h[2][v], h[3][v]);
}

// backward of hidden layer 1 from DN = dL/dn (sample-major [i][k], zero where the unit was off): gamma, beta, W, b gradients
__device__ __forceinline__ void layer1_backward(Lds& L, const float* __restrict__ DN, const float* __restrict__ x, int W, int Bb, int G,
                                                int BE, int bn_l, float* __restrict__ red, float* __restrict__ grad, bool first,
                                                int tid) {
    const int k = tid & 63, part = tid >> 6;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // dg, dbe, db, dW[5]
    const float g_r = L.P[G + k] * L.rstd[bn_l][k];
    for (int i = part * 16; i < part * 16 + 16; ++i) {
        const float dn = DN[i * 64 + k];
        const float* xi = x + i * 5;
        float z = L.P[Bb + k];
#pragma unroll
        for (int j = 0; j < 5; ++j) z = __builtin_fmaf(L.P[W + k * 5 + j], xi[j], z);
        const float zh = (z - L.mean[bn_l][k]) * L.rstd[bn_l][k];
        acc[0] = __builtin_fmaf(dn, zh, acc[0]);
        acc[1] += dn;
        const float dz = dn * g_r;
        acc[2] += dz;
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[3 + j] = __builtin_fmaf(dz, xi[j], acc[3 + j]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) red[(part * 64 + k) * 8 + q] = acc[q];
    __syncthreads();
    for (int o = tid; o < 64 * 8; o += kThreads) {
        const int kk = o >> 3, q = o & 7;
        const float v = red[(0 * 64 + kk) * 8 + q] + red[(1 * 64 + kk) * 8 + q] + red[(2 * 64 + kk) * 8 + q] + red[(3 * 64 + kk) * 8 + q];
        const int p = q == 0 ? G + kk : q == 1 ? BE + kk : q == 2 ? Bb + kk : W + kk * 5 + (q - 3);
        grad[p] = first ? v : grad[p] + v;
    }
    __syncthreads();
}

// Adam (torch.optim.Adam's formula: theta -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)) + soft update of the target,
// four parameters per thread and pass (16-byte accesses; [p0, p1) is 16-byte aligned -- the padding floats carry zero gradients)
template <int P0, int P1>
__device__ __forceinline__ void adam_soft(const Args& A, float lr, float bc1, float bc2s, int tid, float* __restrict__ lds_online = nullptr) {
    const float c1 = 1.0f - A.beta1, c2 = 1.0f - A.beta2, step = lr / bc1, omt = 1.0f - A.tau;
    // every load of the thread's parameters in flight before the first store (the pointers may alias as far as the compiler knows:
    // written as one loop it waits for a round trip to L2 per pass)
    constexpr int NIT = (P1 - P0 + kThreads * 4 - 1) / (kThreads * 4);
    float4 g4[NIT], m4[NIT], v4[NIT], o4[NIT], t4[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = P0 + (it * kThreads + tid) * 4;
        if (p < P1) {
            g4[it] = *reinterpret_cast<const float4*>(A.grad + p); m4[it] = *reinterpret_cast<const float4*>(A.adam_m + p);
            v4[it] = *reinterpret_cast<const float4*>(A.adam_v + p); o4[it] = *reinterpret_cast<const float4*>(A.online + p);
            t4[it] = *reinterpret_cast<const float4*>(A.target + p);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = P0 + (it * kThreads + tid) * 4;
        if (p >= P1) continue;
        const float g[4] = {g4[it].x, g4[it].y, g4[it].z, g4[it].w}, mm[4] = {m4[it].x, m4[it].y, m4[it].z, m4[it].w},
                    vv[4] = {v4[it].x, v4[it].y, v4[it].z, v4[it].w}, oo[4] = {o4[it].x, o4[it].y, o4[it].z, o4[it].w},
                    tt[4] = {t4[it].x, t4[it].y, t4[it].z, t4[it].w};
        float mo[4], vo[4], th[4], tg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mo[q] = __builtin_fmaf(A.beta1, mm[q], c1 * g[q]);
            vo[q] = __builtin_fmaf(A.beta2, vv[q], c2 * g[q] * g[q]);
            th[q] = oo[q] - step * (mo[q] / (sqrtf(vo[q]) / bc2s + A.adam_eps));
            tg[q] = __builtin_fmaf(A.tau, th[q], omt * tt[q]);
        }
        *reinterpret_cast<float4*>(A.adam_m + p) = make_float4(mo[0], mo[1], mo[2], mo[3]);
        *reinterpret_cast<float4*>(A.adam_v + p) = make_float4(vo[0], vo[1], vo[2], vo[3]);
        *reinterpret_cast<float4*>(A.online + p) = make_float4(th[0], th[1], th[2], th[3]);
        *reinterpret_cast<float4*>(A.target + p) = make_float4(tg[0], tg[1], tg[2], tg[3]);
        // (the policy upload that ends the launch reads the new actor from the staged image instead of waiting for L2)
        if (lds_online != nullptr) *reinterpret_cast<float4*>(lds_online + p) = make_float4(th[0], th[1], th[2], th[3]);
    }
}

// Measurement build (tools/learner_phase_probe.py): thread 0 leaves the s_memtime reading (100 MHz) at the end of every phase of
// the launch's LAST update in losses[2 + i], relative to that update's start; losses must then hold 32 floats.
#ifdef MRSIM_LEARNER_PROBE
#define LPROBE(i) do { if (tid == 0 && upd == A.n_updates - 1) A.losses[2 + (i)] = (float)(long long)(__builtin_readcyclecounter() - lprobe_t0); } while (0)
#else
#define LPROBE(i) do { } while (0)
#endif

// Large batches across compute units.  One update has two grid-wide dependencies -- the critic's Adam step needs the gradient of
// the WHOLE batch, and the actor's gradient is taken against the UPDATED critic (RL/MR_ddpg.py:297-302) -- so the multi-workgroup
// form is two launches of batch / 64 workgroups: MODE 1 = targets + critic forward / backward of the workgroup's tile, MODE 2 = actor
// forward, critic forward, backward through both.  Every workgroup leaves its tile's gradients in its own row of `partial`; the
// workgroup that ARRIVES LAST (a ticket from one atomic; nobody spins, so the grid needs no co-residency and cannot deadlock on a
// partitioned or busy device) sums the rows in workgroup order -- deterministic -- and does the Adam step and the soft update.
// MODE 0 = the single-workgroup kernel: all tiles in a loop, n_updates per launch.
enum { kModeAll = 0, kModeCriticHalf = 1, kModeActorHalf = 2 };

// the last workgroup to arrive: true for it alone, after the other workgroups' rows of `partial` have become visible
__device__ __forceinline__ bool mw_arrive_last(const Args& A, Lds& L, int slot, int tid) {
    __threadfence();                      // this workgroup's gradients, device-wide (L2 write-back across XCDs)
    __syncthreads();
    if (tid == 0) L.rows[0] = atomicAdd(A.counter + slot, 1u) == gridDim.x - 1u ? 1 : 0;
    __syncthreads();
    const bool last = L.rows[0] != 0;
    __syncthreads();
    if (last) __threadfence();            // acquire side: nothing of `partial` is read from a stale line
    return last;
}

// grad[P0, P1) = sum over the workgroups' rows, in workgroup order; loss likewise
template <int P0, int P1>
__device__ __forceinline__ void mw_reduce(const Args& A, Lds& L, int which_loss, int tid) {
    const int G = (int)gridDim.x;
    for (int p = P0 + tid * 4; p < P1; p += kThreads * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < G; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(A.partial + (size_t)g * kParams + p);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(A.grad + p) = acc;
    }
    if (tid == 0) {
        float l = 0.f;
        for (int g = 0; g < G; ++g) l += A.loss_partial[g * 2 + which_loss];
        L.loss[which_loss] = l;
    }
    __threadfence_block();
    __syncthreads();
}

template <int MODE>
__device__ __forceinline__ void ddpg_update_body(const Args& A, Lds& L) {
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int ntiles = A.batch / kTile;
    const int t0 = MODE == kModeAll ? 0 : (int)blockIdx.x, t1 = MODE == kModeAll ? ntiles : (int)blockIdx.x + 1;   // this workgroup's tiles
    float* const G_ = MODE == kModeAll ? A.grad : A.partial + (size_t)blockIdx.x * kParams;                        // where its gradients go
    const float invB = 1.0f / (float)A.batch;
    float* X0 = L.X[0]; float* X1 = L.X[1]; float* X2 = L.X[2]; float* X3 = L.X[3]; float* X4 = L.X[4];

    // rows of the ring this update trains on: given, drawn here, or the first `batch` rows
    uint32_t c_lo = A.ctr_lo, c_hi = A.ctr_hi;    // draw counter of the current update
    auto draw = [&](int q, uint32_t round) -> int {   // uniform row in [0, ring_count): one Philox word per (draw, round)
        uint32_t o[4];
        philox4x32_10((uint32_t)q, round, c_lo, c_hi, A.seed_lo, A.seed_hi, o);
        return (int)(((unsigned long long)o[0] * (unsigned long long)(uint32_t)A.ring_count) >> 32);
    };
    const bool sampled = A.idx == nullptr && A.ring_count > 0;
    const bool one_tile = t1 - t0 == 1;
    const int n_upd = MODE == kModeAll ? A.n_updates : 1;     // (the host launches the two halves once per update)
  for (int upd = 0; upd < n_upd; ++upd) {   // (body not re-indented: one update = everything down to the closing brace)
#ifdef MRSIM_LEARNER_PROBE
    const unsigned long long lprobe_t0 = __builtin_readcyclecounter();
#endif
    {
        const unsigned long long c = (((unsigned long long)A.ctr_hi << 32) | A.ctr_lo) + (unsigned long long)upd;
        c_lo = (uint32_t)c; c_hi = (uint32_t)(c >> 32);
    }
    if (sampled && MODE != kModeActorHalf) {
        // (multi-workgroup form: every workgroup draws the same list -- the duplicate check below needs all of it up to 256 rows;
        // beyond, rows are independent and a workgroup draws its own tile's)
        const int q0 = (MODE == kModeAll || A.batch <= 256) ? 0 : t0 * kTile, q1 = (MODE == kModeAll || A.batch <= 256) ? A.batch : t1 * kTile;
        for (int q = q0 + tid; q < q1; q += kThreads) L.sel[q] = draw(q, 0u);
        __syncthreads();
        if (A.batch <= 256 && A.ring_count >= A.batch && A.ring_count < 2 * A.batch) {
            // random.sample's law (RL/MR_ddpg.py:37-44) on a ring that holds fewer than two batches (the first updates of a run:
            // DDPG.train starts learning at min_batch = 64 stored transitions): redrawing duplicates would need ~ring_count rounds
            // when nearly every row must be taken, so the rows come from a partial Fisher-Yates shuffle of [0, ring_count)
            // instead -- exact, no rejection.  perm (< 512 entries) and the batch's Philox words live in the unused tail of sel;
            // the swaps are sequential (one lane, <= 256 LDS swaps: a few microseconds on a path a run takes a handful of times).
            int* const perm = &L.sel[1024];
            uint32_t* const uw = reinterpret_cast<uint32_t*>(&L.sel[2048]);
            for (int q = tid; q < A.ring_count; q += kThreads) perm[q] = q;
            for (int q = tid; q < A.batch; q += kThreads) {
                uint32_t o[4];
                philox4x32_10((uint32_t)q, 0x46595348u /* "FYSH" */, c_lo, c_hi, A.seed_lo, A.seed_hi, o);
                uw[q] = o[0];
            }
            __syncthreads();
            if (tid == 0) {
                for (int j = 0; j < A.batch; ++j) {
                    const int r = j + (int)(((unsigned long long)uw[j] * (unsigned long long)(uint32_t)(A.ring_count - j)) >> 32);
                    const int pj = perm[j], pr = perm[r];
                    perm[j] = pr; perm[r] = pj;
                    L.sel[j] = pr;
                }
            }
            __syncthreads();
        } else if (A.batch <= 256 && A.ring_count >= A.batch) {
            // random.sample's law (RL/MR_ddpg.py:37-44): no row twice.  A draw that repeats an EARLIER one is redrawn until the set
            // is distinct (rejection keeps the joint law uniform over ordered samples without repetition).  With ring_count >= 2 batch
            // a redraw collides with probability < 1/2: 63 rounds leave a duplicate with probability < 2^-55.
            for (uint32_t round = 1; round < 64u; ++round) {
                int dup = 0;
                if (tid < A.batch) {
                    const int mine = L.sel[tid];
                    for (int j = 0; j < tid; j += 4) {     // (sel is 16-byte aligned; entries at or past tid are masked)
                        const int4 e = *reinterpret_cast<const int4*>(&L.sel[j]);
                        dup |= (e.x == mine) | ((j + 1 < tid) & (e.y == mine)) | ((j + 2 < tid) & (e.z == mine)) | ((j + 3 < tid) & (e.w == mine));
                    }
                }
                const int any = __syncthreads_or(dup);
                if (!any) break;
                if (dup) L.sel[tid] = draw(tid, round);
                __syncthreads();
            }
        }
    }
    LPROBE(0);   // rows drawn
    auto load_tile = [&](int tile, bool force) {   // gather the tile's transitions
        if (one_tile && !force) return;            // a single tile stays in LDS for all three phases
        if (tid < kTile) {
            const int q = tile * kTile + tid;
            if constexpr (MODE == kModeActorHalf) L.rows[tid] = A.rows_scratch[q];        // what the critic half trained on
            else L.rows[tid] = A.idx != nullptr ? A.idx[q] : (sampled ? L.sel[q] : q);
            if constexpr (MODE == kModeCriticHalf) A.rows_scratch[q] = L.rows[tid];
            if (MODE != kModeActorHalf && A.idx_out != nullptr) A.idx_out[q] = L.rows[tid];
        }
        __syncthreads();
        for (int q = tid; q < kTile * 5; q += kThreads) {
            const int i = q / 5, j = q - 5 * i;
            L.s[q] = A.s[(long long)L.rows[i] * 5 + j];
            L.s2[q] = A.s2[(long long)L.rows[i] * 5 + j];
        }
        if (tid < kTile * 2) L.a[tid] = A.a[(long long)L.rows[tid >> 1] * 2 + (tid & 1)];
        if (tid < kTile) { L.r[tid] = A.r[L.rows[tid]]; L.d[tid] = A.d[L.rows[tid]]; }
        __syncthreads();
    };
    // critic hidden layer 2 + output for the tile: z2[f][i] = T1 hc1 + T2 act + bt2 over the staged critic; returns this thread's
    // 2 x 4 tile (features 2 ty .., samples 4 tx ..) of z2
    auto critic_l2 = [&](const float* HC1_FM, const float* act, float (&z)[2][4]) {
        lgemm<2, 4, 64, true>(&L.P[C_T1], 64, HC1_FM, 64, ty * 2, tx * 4, z);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f = ty * 2 + u;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = tx * 4 + v;
                z[u][v] += __builtin_fmaf(L.P[C_T2 + f * 2], act[i * 2], __builtin_fmaf(L.P[C_T2 + f * 2 + 1], act[i * 2 + 1], L.P[C_BT2 + f]));
            }
        }
    };

    if (tid == 0) { L.loss[0] = 0.f; L.loss[1] = 0.f; }
    if (tid < 2) {      // Adam bias corrections of this update (the step counters live on the device: graph replays advance them)
        const int t = A.steps[tid] + 1;
        L.bc[tid * 2 + 0] = (float)(1.0 - ipow((double)A.beta1, t));
        L.bc[tid * 2 + 1] = (float)sqrt(1.0 - ipow((double)A.beta2, t));
    }
    // ------------------------------------------------------------------------------------------------ targets y
  if constexpr (MODE != kModeActorHalf) {
    stage_params(L, A.target, A.bn + 3 * 2 * 64, A.bn_eps, tid);
    __syncthreads();
    LPROBE(1);   // target networks staged
    for (int tile = t0; tile < t1; ++tile) {
        load_tile(tile, true);
        layer1(L, L.s2, A_W1, A_B1, A_G1, A_BE1, 0, X0, nullptr, tid);                      // mu'(s2): layer 1
        __syncthreads();
        {
            float z[4][4];
            lgemm<4, 4, 64, true>(&L.P[A_W2], 64, X0, 64, ty * 4, tx * 4, z);                // layer 2: z[f][i]
            float pz[4][2] = {};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = ty * 4 + u;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float n = __builtin_fmaf(L.P[A_G2 + f], (z[u][v] + L.P[A_B2 + f] - L.mean[1][f]) * L.rstd[1][f], L.P[A_BE2 + f]);
                    const float h = n > 0.f ? n : 0.f;
                    pz[v][0] = __builtin_fmaf(L.P[A_W3 + f], h, pz[v][0]);
                    pz[v][1] = __builtin_fmaf(L.P[A_W3 + 64 + f], h, pz[v][1]);
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) { X1[(ty * 64 + tx * 4 + v) * 2] = pz[v][0]; X1[(ty * 64 + tx * 4 + v) * 2 + 1] = pz[v][1]; }
        }
        __syncthreads();
        if (tid < kTile * 2) {                                                              // output layer + tanh: a2 = mu'(s2)
            float z3 = L.P[A_B3 + (tid & 1)];
            for (int q = 0; q < 16; ++q) z3 += X1[q * 128 + tid];
            L.ap[tid] = tanhf(z3) * ((tid & 1) ? A.bound1 : A.bound0);
        }
        layer1(L, L.s2, C_W1, C_B1, C_G1, C_BE1, 2, X0, nullptr, tid);                      // Q'(s2, a2): layer 1 (X0 free: its readers passed the barrier)
        __syncthreads();
        {
            float z[2][4];
            critic_l2(X0, L.ap, z);
            float pq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) pq[v] = __builtin_fmaf(L.P[C_WO + ty * 2 + u], z[u][v] > 0.f ? z[u][v] : 0.f, pq[v]);
#pragma unroll
            for (int v = 0; v < 4; ++v) X2[ty * 64 + tx * 4 + v] = pq[v];
        }
        __syncthreads();
        if (tid < kTile) {
            float q2 = L.P[C_BO];
            for (int q = 0; q < 16; ++q) q2 += X2[q * 64 + tid];
            L.y[(tile - t0) * kTile + tid] = __builtin_fmaf(A.gamma * q2, 1.0f - L.d[tid], L.r[tid]);   // :294
        }
        __syncthreads();
    }
    LPROBE(2);   // targets y
    // ------------------------------------------------------------------------------------------------ critic step
    stage_params(L, A.online, A.bn, A.bn_eps, tid);
    __syncthreads();
    LPROBE(3);   // online networks staged
    for (int tile = t0; tile < t1; ++tile) {
        const bool first = tile == t0;
        load_tile(tile, false);
        layer1(L, L.s, C_W1, C_B1, C_G1, C_BE1, 2, X0, X1, tid);                            // hc1: feature-major X0, sample-major X1
        __syncthreads();
        {
            float z[2][4];
            critic_l2(X0, L.a, z);
            float pq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float h = z[u][v] > 0.f ? z[u][v] : 0.f;
                    X2[(tx * 4 + v) * 32 + ty * 2 + u] = h;                                  // hc2 sample-major [i][32]
                    pq[v] = __builtin_fmaf(L.P[C_WO + ty * 2 + u], h, pq[v]);
                }
#pragma unroll
            for (int v = 0; v < 4; ++v) X3[ty * 64 + tx * 4 + v] = pq[v];
        }
        __syncthreads();
        if (tid < kTile) {
            float q = L.P[C_BO];
            for (int w = 0; w < 16; ++w) q += X3[w * 64 + tid];
            const float e = q - L.y[(tile - t0) * kTile + tid];
            L.dq[tid] = 2.0f * e * invB;                                                    // d mean (y - q)^2 / dq
            const float le = wave_sum(e * e * invB);        // (tid < kTile is exactly wave 0: one writer instead of 64 LDS atomics)
            const float sdq = wave_sum(2.0f * e * invB);    // output bias gradient
            if (tid == 0) { L.loss[0] += le; G_[C_BO] = first ? sdq : G_[C_BO] + sdq; }
        }
        __syncthreads();
        {   // output layer and T2 / bt2 gradients; delta of hidden layer 2 in place of hc2
            const int f = tid & 31, part = tid >> 5;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};   // dWo, dbt2, dT2[0], dT2[1]
            const float wo = L.P[C_WO + f];
            for (int i = part * 8; i < part * 8 + 8; ++i) {
                const float h = X2[i * 32 + f], dq = L.dq[i];
                acc[0] = __builtin_fmaf(dq, h, acc[0]);
                const float dl = h > 0.f ? dq * wo : 0.f;
                X2[i * 32 + f] = dl;
                acc[1] += dl;
                acc[2] = __builtin_fmaf(dl, L.a[i * 2], acc[2]);
                acc[3] = __builtin_fmaf(dl, L.a[i * 2 + 1], acc[3]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) X3[4096 / 2 + (part * 32 + f) * 4 + q] = acc[q];     // upper half of X3: the q partials below are consumed
        }
        __syncthreads();
        if (tid < 32 * 4) {
            const int f = tid >> 2, q = tid & 3;
            float v = 0.f;
            for (int part = 0; part < 8; ++part) v += X3[2048 + (part * 32 + f) * 4 + q];
            const int p = q == 0 ? C_WO + f : q == 1 ? C_BT2 + f : C_T2 + f * 2 + (q - 2);
            G_[p] = first ? v : G_[p] + v;
        }
        {   // dT1[f][k] = sum_i delta2[i][f] hc1[i][k]
            float g[2][4];
            lgemm<2, 4, 64, false>(X2, 32, X1, 64, ty * 2, tx * 4, g);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int p = C_T1 + (ty * 2 + u) * 64 + tx * 4 + v;
                    G_[p] = first ? g[u][v] : G_[p] + g[u][v];
                }
        }
        {   // d hc1[i][k] = sum_f delta2[i][f] T1[f][k]  ->  dL/dn of layer 1 (sample-major, X4)
            float e[4][4];
            lgemm<4, 4, 32, true>(X2, 32, &L.P[C_T1], 64, ty * 4, tx * 4, e);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int o = (ty * 4 + u) * 64 + tx * 4 + v;
                    X4[o] = X1[o] > 0.f ? e[u][v] : 0.f;
                }
        }
        __syncthreads();
        layer1_backward(L, X4, L.s, C_W1, C_B1, C_G1, C_BE1, 2, X3, G_, first, tid);
    }
    __threadfence_block();
    __syncthreads();
    LPROBE(4);   // critic forward + backward
    if constexpr (MODE == kModeCriticHalf) {
        if (tid == 0) A.loss_partial[blockIdx.x * 2 + 0] = L.loss[0];
        if (!A.reduce_in_kernel) return;                 // many tiles: a launch of its own sums the rows across compute units
        if (!mw_arrive_last(A, L, 0, tid)) return;       // (uniform: the whole workgroup leaves)
        mw_reduce<C_W1, kParams>(A, L, 0, tid);
        if (tid == 0) { A.counter[0] = 0u; if (A.losses != nullptr) A.losses[0] = L.loss[0]; }
    }
    adam_soft<C_W1, kParams>(A, A.critic_lr, L.bc[0], L.bc[1], tid);
    __threadfence_block();
    __syncthreads();
    LPROBE(5);   // critic Adam + soft update
    if constexpr (MODE == kModeCriticHalf) return;
  }
    // ------------------------------------------------------------------------------------------------ actor step (against the UPDATED critic)
    if constexpr (MODE == kModeActorHalf) stage_params(L, A.online, A.bn, A.bn_eps, tid);   // a launch of its own: everything is staged here
    else stage_range(L, A.online, C_W1, tid);
    __syncthreads();
    LPROBE(6);   // updated critic re-staged
    for (int tile = t0; tile < t1; ++tile) {
        const bool first = tile == t0;
        load_tile(tile, MODE == kModeActorHalf);
        layer1(L, L.s, A_W1, A_B1, A_G1, A_BE1, 0, X0, X1, tid);                            // h1: X0 feature-major, X1 sample-major
        __syncthreads();
        LPROBE(9);
        {
            float z[4][4];
            lgemm<4, 4, 64, true>(&L.P[A_W2], 64, X0, 64, ty * 4, tx * 4, z);
            float pz[4][2] = {};
            float hh[4][4], zz[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = ty * 4 + u;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float zh = (z[u][v] + L.P[A_B2 + f] - L.mean[1][f]) * L.rstd[1][f];
                    const float n = __builtin_fmaf(L.P[A_G2 + f], zh, L.P[A_BE2 + f]);
                    const float h = n > 0.f ? n : 0.f;
                    hh[u][v] = h; zz[u][v] = zh;
                    pz[v][0] = __builtin_fmaf(L.P[A_W3 + f], h, pz[v][0]);
                    pz[v][1] = __builtin_fmaf(L.P[A_W3 + 64 + f], h, pz[v][1]);
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {                                                    // h2 (X2) and z-hat 2 (X3), sample-major
                *reinterpret_cast<float4*>(X2 + (tx * 4 + v) * 64 + ty * 4) = make_float4(hh[0][v], hh[1][v], hh[2][v], hh[3][v]);
                *reinterpret_cast<float4*>(X3 + (tx * 4 + v) * 64 + ty * 4) = make_float4(zz[0][v], zz[1][v], zz[2][v], zz[3][v]);
                X4[(ty * 64 + tx * 4 + v) * 2] = pz[v][0]; X4[(ty * 64 + tx * 4 + v) * 2 + 1] = pz[v][1];
            }
        }
        __syncthreads();
        LPROBE(10);
        if (tid < kTile * 2) {
            float z3 = L.P[A_B3 + (tid & 1)];
            for (int q = 0; q < 16; ++q) z3 += X4[q * 128 + tid];
            const float t = tanhf(z3);
            L.th[tid] = t;
            L.ap[tid] = t * ((tid & 1) ? A.bound1 : A.bound0);                               // a' = mu(s)
        }
        __syncthreads();
        LPROBE(11);
        layer1(L, L.s, C_W1, C_B1, C_G1, C_BE1, 2, X4, nullptr, tid);                       // Q(s, a') with the new critic: layer 1 -> X4
        __syncthreads();
        LPROBE(12);
        {
            float z[2][4];
            critic_l2(X4, L.ap, z);
            float pq[4][3] = {};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = ty * 2 + u;
                const float wo = L.P[C_WO + f];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const bool on = z[u][v] > 0.f;
                    pq[v][0] = __builtin_fmaf(wo, on ? z[u][v] : 0.f, pq[v][0]);
                    pq[v][1] = __builtin_fmaf(on ? wo : 0.f, L.P[C_T2 + f * 2], pq[v][1]);     // dQ/da'_0
                    pq[v][2] = __builtin_fmaf(on ? wo : 0.f, L.P[C_T2 + f * 2 + 1], pq[v][2]); // dQ/da'_1
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int c = 0; c < 3; ++c) X0[(ty * 64 + tx * 4 + v) * 3 + c] = pq[v][c];   // X0 (h1 feature-major) is free
        }
        __syncthreads();
        LPROBE(13);
        if (tid < kTile) {
            float q = L.P[C_BO], d0 = 0.f, d1 = 0.f;
            for (int w = 0; w < 16; ++w) { q += X0[(w * 64 + tid) * 3]; d0 += X0[(w * 64 + tid) * 3 + 1]; d1 += X0[(w * 64 + tid) * 3 + 2]; }
            const float lq = wave_sum(-q * invB);
            if (tid == 0) L.loss[1] += lq;
            const float t0 = L.th[tid * 2], t1 = L.th[tid * 2 + 1];                          // loss = -mean Q: dL/dz3 = -dQ/da' bound (1 - tanh^2) / B
            const float g0 = -d0 * invB * A.bound0 * (1.0f - t0 * t0), g1 = -d1 * invB * A.bound1 * (1.0f - t1 * t1);
            L.dz3[tid * 2] = g0;
            L.dz3[tid * 2 + 1] = g1;
            const float s0 = wave_sum(g0), s1 = wave_sum(g1);                               // output bias gradients
            if (tid < 2) { const float v = tid ? s1 : s0; G_[A_B3 + tid] = first ? v : G_[A_B3 + tid] + v; }
        }
        __syncthreads();
        LPROBE(14);
        {   // output layer gradients, delta of hidden layer 2 (in place of z-hat 2, X3), gamma / beta / bias 2 gradients
            const int k = tid & 63, part = tid >> 6;
            float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // dg2, dbe2, db2, dW3[0][k], dW3[1][k]
            const float w30 = L.P[A_W3 + k], w31 = L.P[A_W3 + 64 + k], g_r = L.P[A_G2 + k] * L.rstd[1][k];
            for (int i = part * 16; i < part * 16 + 16; ++i) {
                const float h = X2[i * 64 + k], d0 = L.dz3[i * 2], d1 = L.dz3[i * 2 + 1];
                const float dn = h > 0.f ? __builtin_fmaf(d0, w30, d1 * w31) : 0.f;
                acc[0] = __builtin_fmaf(dn, X3[i * 64 + k], acc[0]);
                acc[1] += dn;
                const float dz = dn * g_r;
                acc[2] += dz;
                acc[3] = __builtin_fmaf(d0, h, acc[3]);
                acc[4] = __builtin_fmaf(d1, h, acc[4]);
                X3[i * 64 + k] = dz;
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) X4[(part * 64 + k) * 5 + q] = acc[q];               // X4: its reader (critic_l2) passed two barriers
        }
        __syncthreads();
        LPROBE(15);
        for (int o = tid; o < 64 * 5; o += kThreads) {
            const int k = o / 5, q = o - 5 * k;
            const float v = X4[(0 * 64 + k) * 5 + q] + X4[(1 * 64 + k) * 5 + q] + X4[(2 * 64 + k) * 5 + q] + X4[(3 * 64 + k) * 5 + q];
            const int p = q == 0 ? A_G2 + k : q == 1 ? A_BE2 + k : q == 2 ? A_B2 + k : A_W3 + (q - 3) * 64 + k;
            G_[p] = first ? v : G_[p] + v;
        }
        {   // dW2[f][k] = sum_i dz2[i][f] h1[i][k]
            float g[4][4];
            lgemm<4, 4, 64, false>(X3, 64, X1, 64, ty * 4, tx * 4, g);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int p = A_W2 + (ty * 4 + u) * 64 + tx * 4 + v;
                    G_[p] = first ? g[u][v] : G_[p] + g[u][v];
                }
        }
        __syncthreads();
        LPROBE(16);                                                                    // X2 (h2) was read above by other threads
        {   // d h1[i][k] = sum_f dz2[i][f] W2[f][k]  ->  dL/dn of layer 1 (sample-major, X2)
            float e[4][4];
            lgemm<4, 4, 64, true>(X3, 64, &L.P[A_W2], 64, ty * 4, tx * 4, e);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int o = (ty * 4 + u) * 64 + tx * 4 + v;
                    X2[o] = X1[o] > 0.f ? e[u][v] : 0.f;
                }
        }
        __syncthreads();
        LPROBE(17);
        layer1_backward(L, X2, L.s, A_W1, A_B1, A_G1, A_BE1, 0, X4, G_, first, tid);
    }
    __threadfence_block();
    __syncthreads();
    LPROBE(7);   // actor forward, critic forward, backward through both
    if constexpr (MODE == kModeActorHalf) {
        if (tid == 0) A.loss_partial[blockIdx.x * 2 + 1] = L.loss[1];
        if (!A.reduce_in_kernel) return;
        if (!mw_arrive_last(A, L, 1, tid)) return;
        mw_reduce<A_W1, C_W1>(A, L, 1, tid);
        if (tid == 0) A.counter[1] = 0u;
    }
    adam_soft<A_W1, C_W1>(A, A.actor_lr, L.bc[2], L.bc[3], tid, (MODE == kModeAll && A.pack_blob != nullptr) ? L.P : nullptr);
    if (tid < 2) A.steps[tid] += 1;
    if constexpr (MODE == kModeActorHalf) { if (A.losses != nullptr && tid == 0) A.losses[1] = L.loss[1]; }
    else { if (A.losses != nullptr && tid < 2) A.losses[tid] = L.loss[tid]; }
    __threadfence_block();
    __syncthreads();    // the next update of this launch stages the parameters this one wrote
    LPROBE(8);   // actor Adam + soft update
  }
}

// Many tiles (batch > 256): the sum over the workgroups' rows and the Adam + soft-update step as a launch of their own, four
// parameters per lane over as many workgroups as the half has parameters / 1024 -- the sum over 64 rows of 30 KB is 2 MB of L2
// reads, tens of microseconds for ONE workgroup and a few for the device.  Same order of additions (row 0, 1, 2, ...), same Adam
// arithmetic as adam_soft: bit-identical to the in-kernel form.  HALF 0 = critic (C_W1 .. kParams), 1 = actor (A_W1 .. C_W1).
template <int HALF>
__global__ __launch_bounds__(kThreads) void mr_ddpg_mw_step_kernel(const Args A, int tiles) {
    constexpr int P0 = HALF == 0 ? C_W1 : A_W1, P1 = HALF == 0 ? kParams : C_W1;
    const int tid = threadIdx.x;
    const int p = P0 + ((int)blockIdx.x * kThreads + tid) * 4;
    const int t = A.steps[HALF] + 1;
    const float bc1 = (float)(1.0 - ipow((double)A.beta1, t)), bc2s = (float)sqrt(1.0 - ipow((double)A.beta2, t));
    const float lr = HALF == 0 ? A.critic_lr : A.actor_lr;
    if (p < P1) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < tiles; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(A.partial + (size_t)g * kParams + p);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(A.grad + p) = acc;
        const float c1 = 1.0f - A.beta1, c2 = 1.0f - A.beta2, step = lr / bc1, omt = 1.0f - A.tau;
        const float4 m4 = *reinterpret_cast<const float4*>(A.adam_m + p), v4 = *reinterpret_cast<const float4*>(A.adam_v + p),
                     o4 = *reinterpret_cast<const float4*>(A.online + p), t4 = *reinterpret_cast<const float4*>(A.target + p);
        const float g[4] = {acc.x, acc.y, acc.z, acc.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w},
                    oo[4] = {o4.x, o4.y, o4.z, o4.w}, tt[4] = {t4.x, t4.y, t4.z, t4.w};
        float mo[4], vo[4], th[4], tg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mo[q] = __builtin_fmaf(A.beta1, mm[q], c1 * g[q]);
            vo[q] = __builtin_fmaf(A.beta2, vv[q], c2 * g[q] * g[q]);
            th[q] = oo[q] - step * (mo[q] / (sqrtf(vo[q]) / bc2s + A.adam_eps));
            tg[q] = __builtin_fmaf(A.tau, th[q], omt * tt[q]);
        }
        *reinterpret_cast<float4*>(A.adam_m + p) = make_float4(mo[0], mo[1], mo[2], mo[3]);
        *reinterpret_cast<float4*>(A.adam_v + p) = make_float4(vo[0], vo[1], vo[2], vo[3]);
        *reinterpret_cast<float4*>(A.online + p) = make_float4(th[0], th[1], th[2], th[3]);
        *reinterpret_cast<float4*>(A.target + p) = make_float4(tg[0], tg[1], tg[2], tg[3]);
    }
    if (blockIdx.x == 0 && tid == 0) {
        float l = 0.f;
        for (int g = 0; g < tiles; ++g) l += A.loss_partial[g * 2 + HALF];
        if (A.losses != nullptr) A.losses[HALF] = l;
    }
    // the step counters advance once per update, after BOTH halves have read them: the actor half's step launch is the last
    // kernel of the update, and every workgroup of it has read steps[] above before any can get here -- not guaranteed across
    // workgroups, so the increment is left to the launch that follows (mr_ddpg_mw_count_kernel)
}
__global__ void mr_ddpg_mw_count_kernel(int32_t* steps) {
    if (blockIdx.x == 0 && threadIdx.x < 2) steps[threadIdx.x] += 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Behaviour-policy upload on the device: the learner's online actor (parameter vector above, offsets A_*) -> the packed block the
// env kernels read (mrsim_actor.h layout).  Same arithmetic as the host pair mrsim_actor_fold_bn_host + mrsim_actor_pack_host
// (batch norm folded in double and rounded once; bf16 terms by round-to-nearest-even): the block is bit-identical (tested).
// One workgroup of 256 lanes; runs as a kernel of its own (mrsim_actor_pack_device) or as the tail of the update kernel
// (MrsimDdpgLearner.actor_blob: the policy upload of RL/MR_ddpg.py's loop without a launch -- 9.3 us as a launch, profiles/r05).
// ---------------------------------------------------------------------------------------------------------------------
struct PackArgs {
    const float* params;      // learner layout (actor part); global, or the update kernel's staged image in LDS
    const float* mean;        // moving means of the actor's two batch-norm layers, [l * bn_stride + f]
    const float* var;         // moving variances, likewise
    int bn_stride;
    float* blob;              // [kActBlobFloats]
    float eps, bound0, bound1;
    float scale[5];
};
__device__ __forceinline__ uint16_t bf16_rne_bits(float x) {
    const uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
constexpr int kPackW2Stride = 65;   // rows of the folded W2 in LDS: lanes read one COLUMN element of 32 different rows at a time -- a
                                    // stride of 64 floats puts all of them in one bank (6 us of the 9.3 us this kernel took)
struct PackLds {
    float w1f[64 * 5], b1f[64], w2f[64 * kPackW2Stride], b2f[64];
    double g2s[64];
};
static_assert(sizeof(PackLds) <= sizeof(Lds), "the pack tail reuses the update kernel's LDS image");
__device__ __forceinline__ void actor_pack_body(const PackArgs& A, PackLds& S, int tid) {
    constexpr int H = 64;
    // fold: y = gamma (W x + b - mean) / sqrt(var + eps) + beta == W' x + b'   (double, rounded once; the per-feature gains are
    // formed once -- a double divide and square root per ELEMENT of W2 made this kernel 14 us long)
    if (tid < H) {
        const int f = tid;
        const double g1 = (double)A.params[A_G1 + f] / sqrt((double)A.var[f] + (double)A.eps);
        for (int k = 0; k < 5; ++k) S.w1f[f * 5 + k] = (float)((double)A.params[A_W1 + f * 5 + k] * g1);
        S.b1f[f] = (float)(((double)A.params[A_B1 + f] - (double)A.mean[f]) * g1 + (double)A.params[A_BE1 + f]);
        const double g2 = (double)A.params[A_G2 + f] / sqrt((double)A.var[A.bn_stride + f] + (double)A.eps);
        S.b2f[f] = (float)(((double)A.params[A_B2 + f] - (double)A.mean[A.bn_stride + f]) * g2 + (double)A.params[A_BE2 + f]);
        S.g2s[f] = g2;
    }
    __syncthreads();
    for (int o = tid * 4; o < H * H; o += 256 * 4) {        // (one row of W2 per 16 lanes: the gain is per row)
        const float4 w = *reinterpret_cast<const float4*>(A.params + A_W2 + o);
        const double g = S.g2s[o >> 6];
        float* dst = &S.w2f[(o >> 6) * kPackW2Stride + (o & 63)];
        dst[0] = (float)((double)w.x * g); dst[1] = (float)((double)w.y * g); dst[2] = (float)((double)w.z * g); dst[3] = (float)((double)w.w * g);
    }
    __syncthreads();
    // every float of the block is written below (the host packer zero-fills first: the only bytes it leaves zero are the layer-1
    // slots past the fifth input and the four pad floats of the tail)
    {   // layer-2 A operands, f32: [rt 2][q4 8][lane 64][4] -- one 16-byte store per (rt, q4, lane): 1024 stores, four per lane
        for (int it = tid; it < 2 * 8 * 64; it += 256) {
            const int lane = it & 63, q4 = (it >> 6) & 7, rt = it >> 9, f = 32 * rt + (lane & 31), h = lane >> 5;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = S.w2f[f * kPackW2Stride + act_kperm(4 * q4 + j, h)];
            *reinterpret_cast<float4*>(A.blob + kActA2 + (size_t)it * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    {   // layer-2 A operands as three bf16 terms: [rt 2][s 4][part 3][lane 64][8] -- three 16-byte stores per (rt, s, lane)
        uint16_t* bf = reinterpret_cast<uint16_t*>(A.blob + kActA2bf);
        for (int it = tid; it < 2 * 4 * 64; it += 256) {
            const int lane = it & 63, sx = (it >> 6) & 3, rt = it >> 8, f = 32 * rt + (lane & 31), h = lane >> 5;
            uint32_t pk[3][4];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                float r = S.w2f[f * kPackW2Stride + act_kperm(8 * sx + jj, h)];
#pragma unroll
                for (int part = 0; part < 3; ++part) {
                    const uint16_t t = bf16_rne_bits(r);
                    if (jj & 1) pk[part][jj >> 1] |= (uint32_t)t << 16; else pk[part][jj >> 1] = (uint32_t)t;
                    r -= __uint_as_float((uint32_t)t << 16);
                }
            }
#pragma unroll
            for (int part = 0; part < 3; ++part)
                *reinterpret_cast<uint4*>(bf + ((((size_t)(rt * 4 + sx) * 3 + part) * 64 + lane) * 8)) =
                    make_uint4(pk[part][0], pk[part][1], pk[part][2], pk[part][3]);
        }
    }
    if (tid < 128) {
        const int rt = tid >> 6, lane = tid & 63, f = 32 * rt + (lane & 31), h = lane >> 5;
        for (int sx = 0; sx < 3; ++sx) {
            const int k = 2 * sx + h;
            A.blob[kActA1 + (rt * 3 + sx) * 64 + lane] = k < 5 ? S.w1f[f * 5 + k] * A.scale[k] : 0.0f;
        }
    } else if (tid < 192) {
        const int o = tid - 128, h = o >> 5, q = o & 31;
        A.blob[kActC1 + h * 32 + q] = S.b1f[act_kperm(q, h)];
        A.blob[kActC2 + h * 32 + q] = S.b2f[act_kperm(q, h)];
        for (int oo = 0; oo < 2; ++oo) A.blob[kActW3 + (h * 2 + oo) * 32 + q] = A.params[A_W3 + oo * H + act_kperm(q, h)];
    } else if (tid == 192) {
        A.blob[kActTail + 0] = A.params[A_B3 + 0]; A.blob[kActTail + 1] = A.params[A_B3 + 1];
        A.blob[kActTail + 2] = A.bound0; A.blob[kActTail + 3] = A.bound1;
#pragma unroll
        for (int j = 4; j < 8; ++j) A.blob[kActTail + j] = 0.0f;
    }
}
__global__ __launch_bounds__(256) void mr_actor_pack_kernel(const PackArgs A) {
    __shared__ PackLds S;
    actor_pack_body(A, S, threadIdx.x);
}

__global__ __launch_bounds__(kThreads) void mr_ddpg_update_kernel(const Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ddpg_update_body<kModeAll>(A, *reinterpret_cast<Lds*>(lds_raw));
    if (A.pack_blob != nullptr) {   // the behaviour policy's block from the parameters the last update just wrote: read from the
        // staged image (the actor's Adam step left its new values there too; the online network's batch-norm constants are the
        // ones staged last) -- no trip to L2; the fold's scratch lives in the activation buffers, which are free now
        Lds& L = *reinterpret_cast<Lds*>(lds_raw);
        static_assert(sizeof(PackLds) <= sizeof(L.X), "pack scratch inside the activation buffers");
        const PackArgs K{L.P, &L.mean[0][0], &L.var[0][0], 64, A.pack_blob, A.bn_eps, A.bound0, A.bound1,
                         {A.pack_scale[0], A.pack_scale[1], A.pack_scale[2], A.pack_scale[3], A.pack_scale[4]}};
        actor_pack_body(K, *reinterpret_cast<PackLds*>(&L.X[0][0]), threadIdx.x);
    }
}
// the multi-workgroup form: grid = batch / 64 workgroups, the two halves of ONE update (see ddpg_update_body)
__global__ __launch_bounds__(kThreads) void mr_ddpg_mw_critic_kernel(const Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ddpg_update_body<kModeCriticHalf>(A, *reinterpret_cast<Lds*>(lds_raw));
}
__global__ __launch_bounds__(kThreads) void mr_ddpg_mw_actor_kernel(const Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ddpg_update_body<kModeActorHalf>(A, *reinterpret_cast<Lds*>(lds_raw));
}

// ---------------------------------------------------------------------------------------------------------------------
// Replay ring feed (RL/MR_ddpg.py:279-281 replay_buffer.add for a sampled subset of one collected launch group): n transitions
// (t, env) drawn by Philox from a resident [T][N][.] rollout, written to ring rows head .. head + n - 1 (mod capacity):
//   s = obs before step t (prev_obs for t = 0, else obs_T[t-1]) * obs_scale, a = actions_T[t], r = rew_T[t], done = done_T[t],
//   s2 = obs_T[t] * obs_scale   (after an auto-reset that is the next episode's reset observation; its target is r alone)
// ---------------------------------------------------------------------------------------------------------------------
struct PushArgs {
    const float* obs_T; const float* act_T; const float* rew_T; const uint8_t* done_T; const float* prev_obs;   // [T][N][5] ... [N][5]
    float* s; float* a; float* r; float* d; float* s2;                                                           // the ring
    long long N; int32_t T, n, capacity, head;
    float scale[5];
    uint32_t seed_lo, seed_hi, ctr_lo, ctr_hi;
};
// One wave per workgroup: every lane gathers from a different page of the [T][N][.] buffers (hundreds of MB), and 64 workgroups
// spread the address translations over 64 compute units' TLBs instead of 16.
constexpr int kPushThreads = 64;
__global__ __launch_bounds__(kPushThreads) void mr_replay_push_kernel(const PushArgs A) {
    const int q = blockIdx.x * kPushThreads + threadIdx.x;
    if (q >= A.n) return;
    uint32_t o[4];
    philox4x32_10((uint32_t)q, 0x52494E47u /* "RING" */, A.ctr_lo, A.ctr_hi, A.seed_lo, A.seed_hi, o);
    const int t = (int)(((unsigned long long)o[0] * (unsigned long long)(uint32_t)A.T) >> 32);
    const long long e = (long long)(((unsigned long long)o[1] * (unsigned long long)A.N) >> 32);
    const int row = (A.head + q) % A.capacity;
    const float* so = t == 0 ? A.prev_obs + e * 5 : A.obs_T + ((long long)(t - 1) * A.N + e) * 5;
    const float* s2o = A.obs_T + ((long long)t * A.N + e) * 5;
#pragma unroll
    for (int j = 0; j < 5; ++j) { A.s[(long long)row * 5 + j] = so[j] * A.scale[j]; A.s2[(long long)row * 5 + j] = s2o[j] * A.scale[j]; }
    const long long te = (long long)t * A.N + e;
    A.a[(long long)row * 2] = A.act_T[te * 2]; A.a[(long long)row * 2 + 1] = A.act_T[te * 2 + 1];
    A.r[row] = A.rew_T[te];
    A.d[row] = A.done_T[te] ? 1.0f : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------------
// The per-step bookkeeping of the reference-shaped loop (RL/MR_ddpg.py:278-282,307-311) as ONE launch: replay_buffer.add of the
// step's n transitions (s = the observation the action was computed from, s2 = the terminal observation where the episode ended --
// with auto-reset the env's own observation is already the next episode's reset row), `state = next_state`, and the sum / count of
// the returns of the episodes that ended at this step.  mr_rl_amd/ddpg.py did this with ~20 small PyTorch kernels per step.
// ---------------------------------------------------------------------------------------------------------------------
struct AddStepArgs {
    const float* obs_prev; const float* act; const float* rew; const uint8_t* done; const float* obs_next; const float* final_obs;
    const float* final_ret;
    float* s; float* a; float* r; float* d; float* s2;     // the ring
    float* obs_prev_out;                                    // [n][5]: receives obs_next (may be obs_prev itself)
    float* ended;                                           // [2] += {sum of final_ret over done envs, their number}, or null
    long long n; int32_t capacity, head, skip;              // the first `skip` envs are not stored (n > capacity: the last ones stay)
    float scale[5];
};
__global__ __launch_bounds__(256) void mr_replay_add_step_kernel(const AddStepArgs A) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float ret = 0.f, cnt = 0.f;
    if (i < A.n) {
        const bool dn = A.done[i] != 0;
        float so[5], s2o[5], nx[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) { so[j] = A.obs_prev[i * 5 + j]; nx[j] = A.obs_next[i * 5 + j]; }
#pragma unroll
        for (int j = 0; j < 5; ++j) s2o[j] = (dn && A.final_obs != nullptr) ? A.final_obs[i * 5 + j] : nx[j];
        if (i >= A.skip) {
            const long long row = (long long)((A.head + (i - A.skip)) % A.capacity);
#pragma unroll
            for (int j = 0; j < 5; ++j) { A.s[row * 5 + j] = so[j] * A.scale[j]; A.s2[row * 5 + j] = s2o[j] * A.scale[j]; }
            A.a[row * 2] = A.act[i * 2]; A.a[row * 2 + 1] = A.act[i * 2 + 1];
            A.r[row] = A.rew[i];
            A.d[row] = dn ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) A.obs_prev_out[i * 5 + j] = nx[j];
        if (dn && A.final_ret != nullptr) { ret = A.final_ret[i]; cnt = 1.f; }
    }
    if (A.ended != nullptr) {       // (every lane of the wave is here: full-wave reduction, one atomic pair per wave that saw an end)
        ret = wave_sum(ret); cnt = wave_sum(cnt);
        if ((threadIdx.x & 63) == 0 && cnt > 0.f) { atomicAdd(A.ended, ret); atomicAdd(A.ended + 1, cnt); }
    }
}

}  // namespace learner
}  // namespace mrsim
