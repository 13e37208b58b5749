/*
 * mrsim.h -- C ABI of libmrsim.so: the MI355X (gfx950) implementation of the
 * MR_env.step() / MR_simulator hot path of SuhailSama/MR_RL for N environments.
 *
 * The reference has no FFI / plugin boundary on this path: it is a pure Python
 * class API (class MR_Env(gym.Env), MR_env.py:21).  These entry points are what a
 * ctypes binding placed in the reference's MR_env.py would bind (INTEGRATION.md
 * shows that stub); each one names the reference interface it replaces.
 *
 * Conventions
 *   - plain C, no torch types; every pointer is a DEVICE pointer owned by the
 *     caller (e.g. torch tensors' data_ptr()) unless the name ends in _host;
 *   - the library allocates no persistent device memory and keeps no globals;
 *   - `stream` is a hipStream_t (0 = the null stream); calls are asynchronous;
 *   - every entry point returns MRSIM_OK (0) or a negative MRSIM_E* code, no
 *     exceptions cross the boundary; mrsim_strerror() names a code;
 *   - the RNG is stateless: (seed, global env id, step_idx) are arguments, so
 *     a shard [env_id0, env_id0+n) produces the same trajectories on any rank.
 *   - there is NO CPU fallback: without a HIP device every compute entry point
 *     returns MRSIM_ENODEVICE.
 *
 * seed / step_idx contract.  All randomness (noise at every RHS evaluation, sampled start
 * positions, the exploration policy) is a pure function of (seed, GLOBAL env id, step_idx) through
 * Philox4x32-10 (definition: DESIGN.md section 5, oracle/mrsim_oracle.c).  The caller owns the step
 * counter: pass a fresh step_idx to every mrsim_reset / mrsim_step call (mrsim_rollout consumes
 * step_idx0 .. step_idx0 + T - 1); passing the same (seed, step_idx) again reproduces the same draws.
 * MrsimParams.step_base lets that counter live in device memory for hipGraph replay.
 *
 * A reset CONSUMES a step index: the first mrsim_step / mrsim_rollout after mrsim_reset(step_idx = s) must use s + 1 or later.
 * The draws of an auto-reset (start position, constructor noise) sit at the step index at which the episode that ENDS took its
 * first step -- a block of its own per episode, known from the episode's start.  An explicit reset at s followed by a step at the
 * same s would make that step the first of an episode whose auto-reset block is (env, s) again: the same start position and
 * constructor noise twice in a row.  mr_rl_amd.MRVecEnv advances its counter in reset(); a C caller must do the same
 * (tests/test_rng.py pins that the blocks of an explicit reset and of the following episodes' auto-resets are distinct).
 */
#ifndef MRSIM_H
#define MRSIM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRSIM_ABI_VERSION 5

enum {
    MRSIM_OK = 0,
    MRSIM_EINVAL = -1,    /* bad argument (null pointer, n < 0, unknown enum)            */
    MRSIM_ENODEVICE = -2, /* no HIP device / wrong architecture                           */
    MRSIM_ELAUNCH = -3,   /* hipLaunchKernel / HIP runtime error                          */
    MRSIM_EALIGN = -4,    /* a buffer is not 16-byte aligned                              */
    MRSIM_ERANGE = -5,    /* n or env ids exceed 2^32                                     */
    MRSIM_ETIMEOUT = -6   /* mrsim_host_wait_word: the word did not take the value in time */
};

enum { MRSIM_INT_RK45 = 0, MRSIM_INT_EULER = 1, MRSIM_INT_RK4 = 2 };
enum { MRSIM_REW_CONSTANT10 = 0, MRSIM_REW_GOAL = 1 };
enum { MRSIM_OBS_AOS = 0 /* [N][5] */, MRSIM_OBS_SOA = 1 /* [5][N] */ };
/* FAST: hardware v_log/v_sqrt/v_sin/v_cos (normals within ~1e-6 of SPEC).  SPEC: the operation-by-
 * operation fp32 definition shared with the CPU oracle -- normals are bit-identical to the oracle's. */
enum { MRSIM_NOISE_FAST = 0, MRSIM_NOISE_SPEC = 1 };
/* Where the normals of an RK45 env step enter (ABI 4; the default changed in ABI 5).  The reference draws from NumPy's global
 * MT19937 stream, which no counter-based generator reproduces: at sigma > 0 either law matches the reference in DISTRIBUTION
 * (and the CPU oracle element for element on equal seeds), neither number for number.
 * COLLAPSED (default since ABI 5): the stage normals of one rk_step attempt reach a result only through the B-weighted sum
 *   (position) and the E-weighted sum (error estimate); those two are jointly Gaussian with the tableau's fixed covariance
 *   and are drawn directly (2 normals per noise component instead of 4; f_new and the constructor keep their own draws).
 *   Every quantity a step returns or carries has the law it has under PER_STAGE and in the reference: pinned against
 *   statistics of the reference's own Simulator.step (tests/golden/ref_increments.npz: 1e6 env steps per model law far from
 *   the origin, 1e6 where the first attempt's error_norm is about 1, 4e5 per start where every step is split into 20-50
 *   attempts; increments at |std ratio - 1| < 0.005 + KS, attempts per step by chi-square; CPU oracle and kernels) and
 *   against the per-stage oracle over 1e6 steps x 8 configurations (tests/test_noise_law_cpu.py).  RK45 integrator only
 *   (the fixed-step modes ignore it); 2 Philox calls + 2 Box-Muller pairs per env step instead of 3 + 5.
 * PER_STAGE: a fresh N(0, sigma) at every RHS evaluation, in the reference's order (MR_simulator.py:73-83; 8 evaluations
 *   per env step: stages K1..K5, f_new, and the two of the next RK45 constructor) -- the layout the oracle's tape replays
 *   of the reference's own np.random.normal draws use (tests/golden/ref_noise.npz); about 1.25 x the instructions. */
enum { MRSIM_LAW_PER_STAGE = 0, MRSIM_LAW_COLLAPSED = 1 };

/* Tunables of the reference path, with the place each one lives in the reference. */
typedef struct MrsimParams {
    double time_span;      /* Simulator.time_span = 0.030            MR_simulator.py:12        */
    double rtol;           /* time_span/number_iterations = 3e-4      MR_simulator.py:13,91     */
    double atol;           /* 1e-4                                    MR_simulator.py:91        */
    double a0;             /* Simulator.a0 (reset kwarg)              MR_env.py:168,180         */
    double sigma;          /* Simulator.noise_var (a std-dev)         MR_env.py:167,179         */
    double min_dist2goal;  /* 30                                      MR_env.py:63              */
    double obs_low[5];     /* observation_space                       MR_env.py:37-39           */
    double obs_high[5];
    double init_low[2];    /* init_space                              MR_env.py:40-42           */
    double init_high[2];
    double act_low[2];     /* range of the on-device random policy (DDPG actor range,          */
    double act_high[2];    /*   RL/MR_ddpg.py:136-137,345: [-20,20] x [-2pi,2pi])               */
    int32_t mismatched;    /* Simulator.is_mismatched                 MR_env.py:169,183         */
    int32_t integrator;    /* MRSIM_INT_*; RK45 = the reference (SciPy RK45 semantics)          */
    int32_t substeps;      /* Euler / RK4 sub-steps per env step (BASELINE configs 2, 3)        */
    int32_t reward_mode;   /* CONSTANT10 = MR_env.py:89; GOAL = calculate_reward MR_env.py:118  */
    int32_t max_timesteps; /* 50                                      MR_env.py:62              */
    int32_t auto_reset;    /* same-step auto reset (extension; 0 = reference behaviour)         */
    int32_t goal_K;        /* goal/trajectory table [K][T][2] f32; NULL table = goal (0,0)      */
    int32_t goal_T;        /*   = MR_Env.init_goal, MR_env.py:57                                */
    int32_t obs_layout;    /* MRSIM_OBS_*                                                       */
    int32_t noise_math;    /* MRSIM_NOISE_*: how Box-Muller is evaluated (same uniforms either way)    */
    int32_t auto_reset_fresh_env; /* which env object an auto-reset stands for (ABI 3).  0 (default): the SAME   */
                           /*   MR_Env re-used, `state = env.reset(...)` at the top of every episode             */
                           /*   (RL/MR_ddpg.py:270): reset_start_pos builds the RK45 object BEFORE reset() sets  */
                           /*   is_mismatched (MR_env.py:181-183), i.e. under the law the previous episode left  */
                           /*   behind = params.mismatched.  1: a fresh MR_Env per episode (utils.run_sim,       */
                           /*   utils.py:46): the constructor runs under the nominal law.  Same thing unless     */
                           /*   mismatched != 0.                                                                 */
    int32_t noise_law;     /* MRSIM_LAW_* (mrsim_default_params: COLLAPSED; 0 = PER_STAGE)                       */
    const uint64_t* step_base; /* optional DEVICE word added to every step_idx argument.  Kernel      */
                           /*   arguments are frozen inside a captured hipGraph; keeping the base in */
                           /*   HBM (advanced by mrsim_advance_step_base) lets each replay draw new  */
                           /*   noise.  NULL = 0.                                                    */
} MrsimParams;

/* Per-env persistent state in HBM, caller-owned.  16-byte records so that every
 * lane moves one dwordx4 per array.
 *   pos[n]  = {x, y} fp64           Simulator.last_state == integrator.y  (MR_simulator.py:45)
 *   aux[n]  = {f0x, f0y, hq, cnt}   the live RK45 object's carried state: integrator.f
 *             (f32, f32, f32, i32)  (first stage of the NEXT step -- the stale-stage
 *                                   quirk), hq = integrator.h_abs / time_span,
 *                                   cnt = MR_Env.counter (MR_env.py:61,80)
 *   ep_ret[n] fp32                  running episode return (extension)                 */
typedef struct MrsimState {
    double* pos;
    float* aux;
    float* ep_ret;
} MrsimState;

/* ---------------------------------------------------------------------------------------------------------
 * The DDPG actor as an on-device policy source (ABI 3) -- RL/MR_ddpg.py:80-160 ActorNetwork, :59-78 OUNoise.
 *
 * The consumer of this path is the collection loop of RL/MR_ddpg.py:270-311:
 *     action = actor.predict(state) + actor_noise();  next_state, reward, done, _ = env.step(action)
 * With the actor in host Python between two kernel launches that loop runs two orders of magnitude below the env
 * kernels, so the policy is a policy SOURCE of the kernels: mrsim_step / mrsim_rollout evaluate the network on the
 * observation they hold in registers (MrsimStepIO.actor / MrsimRolloutIO.actor), and mrsim_actor_forward is the same
 * arithmetic as a kernel of its own (the gym-loop form: obs[n][5] -> actions[n][2]).  All three produce equal bits.
 *
 * Network (create_actor_network, :120-137): 5 -> fully_connected 64 -> batch_normalization -> relu -> fully_connected 64
 * -> batch_normalization -> relu -> fully_connected 2 (tanh), scaled_out = out * action_bound.  predict() runs the
 * batch normalisation on its moving statistics (inference mode), a per-feature affine map: fold it into the preceding
 * layer with mrsim_actor_fold_bn_host, then pack the six arrays with mrsim_actor_pack_host into the parameter block the
 * kernels read (layout private to the library: weights pre-permuted for the f32 MFMA operand maps).  The two 64-wide
 * layers run on the matrix cores (v_mfma_f32_32x32x2_f32, exact f32: bit-for-bit an fmaf chain in the documented order),
 * the output layer and tanh (specified fp32 arithmetic) on the vector unit.
 *
 * Exploration noise (OUNoise.__call__, :69-73, mu = 0): x += -theta x dt + sigma sqrt(dt) N(0,1) per action component,
 * one process per env, state in ou_state[n][2]; its two normals are words 0,1 of RNG call DYN(0,0) of the step -- the
 * words the uniform exploration policy would use (mrsim_random_policy): the two policy sources are alternatives.
 * --------------------------------------------------------------------------------------------------------- */
#define MRSIM_ACTOR_HIDDEN 64
#define MRSIM_ACTOR_BLOB_FLOATS 10888  /* size of the packed parameter block (f32 section 4744 + bf16x3 section 6144) */
/* Arithmetic of the two hidden layers (MrsimActor.math).  F32: exact f32 products on v_mfma_f32_32x32x2_f32, bit-for-bit an
 * fmaf chain in the documented order.  BF16X3: every f32 operand as the sum of three bf16 terms, the six products above
 * 2^-24 on v_mfma_f32_32x32x16_bf16 with f32 accumulation -- f32-class accuracy (within 5e-6 of the action bound of the F32
 * result in the tests) on the matrix cores proper, which run beside the vector unit; about 1.5 x the collection rate. */
/* BF16: plain bf16 operands (weights and activations rounded once), f32 accumulation, in all THREE layers (the output layer runs
 * on the matrix cores as well) -- ordinary bf16 inference: the action within 7e-5 of its bound of the F32 result with the
 * reference's output-layer init, 2-6e-2 at output gains of 20-40 x; tanh on the hardware's exp2 / reciprocal (absolute error ~1e-7). */
enum { MRSIM_ACTOR_F32 = 0, MRSIM_ACTOR_BF16X3 = 1, MRSIM_ACTOR_BF16 = 2 };

/* Weights and observations must be finite.  (The kernels' ReLU is an integer maximum on the float's bits: for a NaN it returns 0
 * when the sign bit is set and the NaN otherwise, unlike fmaxf.) */
typedef struct MrsimActorWeights {   /* HOST pointers, row-major float32: the network in inference form */
    const float* w1;        /* [64][5]   first fully_connected (+ folded batch norm)   RL/MR_ddpg.py:122-123 */
    const float* b1;        /* [64]                                                                          */
    const float* w2;        /* [64][64]  second fully_connected (+ folded batch norm)  :125-126              */
    const float* b2;        /* [64]                                                                          */
    const float* w3;        /* [2][64]   output layer, tanh                            :130-134              */
    const float* b3;        /* [2]                                                                           */
    float obs_scale[5];     /* input scaling folded into w1 (1 = the reference: raw observations)            */
    float action_bound[2];  /* scaled_out = tanh(.) * action_bound   :136-137, :345 (env.action_space.high)  */
} MrsimActorWeights;

/* y = gamma (W x + b - mean) / sqrt(var + eps) + beta  ==  W' x + b'  (tflearn batch_normalization at inference;
 * w [rows][cols], everything else [rows]).  Host arithmetic in double, rounded once.  No device needed. */
int mrsim_actor_fold_bn_host(int32_t rows, int32_t cols, const float* w, const float* b, const float* gamma,
                             const float* beta, const float* mean, const float* var, float eps, float* w_out,
                             float* b_out);
/* Pack the inference-form network into blob_host[MRSIM_ACTOR_BLOB_FLOATS]; copy that block to device memory
 * (16-byte aligned) and pass it as MrsimActor.blob.  No device needed. */
int mrsim_actor_pack_host(const MrsimActorWeights* w, float* blob_host);

typedef struct MrsimActor {
    const float* blob;        /* DEVICE: packed parameters; NULL = no actor                                   */
    float* ou_state;          /* DEVICE [n][2] OUNoise.x_prev per env, updated in place; NULL = no exploration */
                              /*   noise (actor.predict alone)                                                */
    float ou_theta;           /* 0.15   OUNoise defaults, RL/MR_ddpg.py:60                                    */
    float ou_sigma;           /* 0.3                                                                          */
    float ou_dt;              /* 1e-2                                                                         */
    int32_t ou_reset_on_done; /* 0 = the reference (the process is never reset, :270-311); 1 = x_prev := 0 at  */
                              /*   the first step of every episode (MR_Env.counter == 0)                      */
    int32_t math;             /* MRSIM_ACTOR_F32 (default) | MRSIM_ACTOR_BF16X3 | MRSIM_ACTOR_BF16            */
    int32_t reserved0;        /* 0                                                                            */
} MrsimActor;

/* Optional sink of a step launch whose policy is the in-kernel actor (ABI 5): the step writes its n transitions straight into a
 * replay ring -- RL/MR_ddpg.py:278-282 `replay_buffer.add(state, action, reward, terminal, next_state)` without a launch of its
 * own.  Env i of the launch goes to row (head + i - skip) mod capacity, skip = max(0, n - capacity) (the last `capacity` envs stay);
 * s = the observation the actor was evaluated on x obs_scale, a = the applied action, r, done, s2 = the observation after the step
 * (the TERMINAL one where the episode ended, also under auto_reset) x obs_scale -- the rows mrsim_replay_add_step writes from the
 * step's outputs, bit for bit.  ended2 (optional) [2] += {sum of the returns of the episodes that ended (auto_reset), their number}. */
typedef struct MrsimReplaySink {
    float* s;            /* DEVICE [capacity][5] */
    float* a;            /* DEVICE [capacity][2] */
    float* r;            /* DEVICE [capacity]    */
    float* done;         /* DEVICE [capacity]: 1.0 / 0.0 */
    float* s2;           /* DEVICE [capacity][5] */
    float* ended2;       /* DEVICE [2], optional */
    int32_t capacity;
    int32_t head;
    float obs_scale[5];
    int32_t reserved0;   /* 0 */
} MrsimReplaySink;

/* Inputs / outputs of one step.  Optional pointers may be NULL. */
typedef struct MrsimStepIO {
    const float* actions;    /* [n][2] {f_t, alpha_t}  (MR_env.py:81-82).  NULL: draw the     */
                             /*   random policy in-kernel (uniform in act_low..act_high)      */
    float* actions_out;      /* optional [n][2]: the actions actually applied                 */
    const float* goal_table; /* optional [K][T][2]                                            */
    float* obs;              /* [n][5] or [5][n]: x, y, goal_x, goal_y, dist  (MR_env.py:100) */
    float* rew;              /* [n]                                           (MR_env.py:89)  */
    uint8_t* done;           /* [n]                                           (MR_env.py:136) */
    float* state_prime;      /* optional [n][2]: Simulator.state_prime (MR_simulator.py:87)   */
    float* final_obs;        /* optional, layout of obs: terminal obs where done && auto_reset */
    float* final_ret;        /* optional [n]: episode return where done                       */
    int32_t* final_len;      /* optional [n]: episode length where done                       */
    int32_t* status;         /* optional [1]: OR-ed per-env flags; bit0 = RK45 attempt guard  */
                             /*   tripped (the reference would raise "failed solver").  With  */
                             /*   n == 1 the word is updated by a plain load / store (it may  */
                             /*   live in mrsim_host_alloc memory), otherwise by an atomic    */
    const MrsimActor* actor; /* optional (HOST pointer; ABI 3): the policy source is the      */
                             /*   in-kernel actor on the env's current observation; actions   */
                             /*   must then be NULL and the integrator RK45                   */
    int32_t* attempts;       /* optional [n] (ABI 5): rk_step attempts this env step took =   */
                             /*   (integrator.nfev after - before) / 6 of the RK45 object     */
                             /*   that integrates it (MR_simulator.py:42-43); fixed-step      */
                             /*   modes: substeps                                             */
    const MrsimReplaySink* replay; /* optional (HOST pointer; ABI 5): needs `actor`             */
    int32_t* done_word;      /* optional (ABI 5; launches of ONE workgroup only, n <= 256):    */
    int32_t done_value;      /*   after every output of the launch has been written and made   */
    int32_t reserved0;       /*   visible at system scope, done_value is stored there (release) */
                             /*   -- a word of mrsim_host_alloc memory the host polls with     */
                             /*   mrsim_host_wait_word instead of waiting for the stream: a    */
                             /*   kernel's completion signal arrives microseconds after its    */
                             /*   last store (the one-env facade's step is that wait)           */
} MrsimStepIO;

/* ---------------------------------------------------------------------------------------------------------
 * The DDPG learner update as ONE launch (ABI 4) -- RL/MR_ddpg.py:288-305: for a batch of transitions
 *     y = r + gamma Q'(s2, mu'(s2)) (1 - done);  critic: Adam step on mean (y - Q(s, a))^2  (CriticNetwork.train);
 *     actor: Adam step along dQ/da at a = mu(s) against the UPDATED critic (critic.action_gradients -> ActorNetwork.train);
 *     both target networks <- tau online + (1 - tau) target  (update_target_network)
 * for the networks of :120-137 / :207-223.  Batch normalisation is the fixed affine map it is in the script (tflearn's
 * training mode is never switched on): moving statistics as constants, trainable gamma / beta.  Adam follows
 * torch.optim.Adam's formula (the parity target of the tests is the PyTorch twin mr_rl_amd/ddpg.py; TF1's AdamOptimizer
 * differs only in where epsilon enters; the reference itself is unpinned, TF1 / tflearn being absent).
 * Parameter vector layout (floats; online, target, Adam m, Adam v and the gradient scratch all use it): row-major tensors
 *   actor : W1[64][5] @0, b1 @320, gamma1 @384, beta1 @448, W2[64][64] @512, b2 @4608, gamma2 @4672, beta2 @4736,
 *           W3[2][64] @4800, b3[2] @4928;      (2 unused floats)
 *   critic: W1[64][5] @4932, b1 @5252, gamma1 @5316, beta1 @5380, T1[32][64] @5444, T2[32][2] @7492, bt2[32] @7556,
 *           Wo[32] @7588, bo @7620;            padded to MRSIM_DDPG_PARAMS floats.
 * --------------------------------------------------------------------------------------------------------- */
#define MRSIM_DDPG_PARAMS 7680
#define MRSIM_DDPG_MAX_BATCH 4096
typedef struct MrsimDdpgLearner {
    float* online;           /* DEVICE [MRSIM_DDPG_PARAMS], 16-byte aligned: the online actor + critic, updated in place  */
    float* target;           /* DEVICE: the target networks, soft-updated in place                                        */
    float* adam_m;           /* DEVICE: Adam first moments                                                                */
    float* adam_v;           /* DEVICE: Adam second moments                                                               */
    float* grad_scratch;     /* DEVICE: gradient scratch (holds the update's gradients on return)                         */
    int32_t* steps;          /* DEVICE [2]: Adam step counts of the critic and the actor; the kernel increments them      */
    const float* bn_stats;   /* DEVICE [2][3][2][64]: {online, target} x {actor bn1, actor bn2, critic bn1} x {moving    */
                             /*   mean, moving variance} -- constants of the update                                       */
    float bn_eps;            /* 1e-5 (tflearn batch_normalization epsilon)                                                */
    float gamma;             /* 0.99   RL/MR_ddpg.py:339                                                                  */
    float tau;               /* 0.001  :338                                                                               */
    float actor_lr;          /* 1e-3   :341                                                                               */
    float critic_lr;         /* 1e-2   :342                                                                               */
    float beta1, beta2, adam_eps; /* 0.9, 0.999, 1e-8 (Adam defaults)                                                     */
    float action_bound[2];   /* :345 env.action_space.high                                                                */
    float* batch_scratch;    /* DEVICE, 16-byte aligned, ZERO-INITIALISED once by the caller, or NULL (ABI 5): work space of the  */
                             /*   multi-workgroup form -- with it, a batch of more than 64 transitions runs as batch / 64         */
                             /*   workgroups on as many compute units (two to five launches per update: the critic's step needs  */
                             /*   the whole batch's gradient, the actor's gradient the updated critic) instead of one workgroup  */
                             /*   looping over the tiles; partial gradients are summed in tile order: bit-identical to that loop */
    int64_t batch_scratch_floats; /* its size in floats: >= MRSIM_DDPG_BATCH_SCRATCH_FLOATS(batch)                                 */
    float* actor_blob;       /* DEVICE [MRSIM_ACTOR_BLOB_FLOATS], 16-byte aligned, or NULL (ABI 5): after the LAST update of the call the */
                             /*   online actor is folded and packed into this block exactly as mrsim_actor_pack_device does (same bits)   */
                             /*   -- the behaviour policy follows the learner (RL/MR_ddpg.py:277 actor.predict uses the weights of :302)  */
                             /*   without a launch of its own: the tail of the update kernel (a launch after the multi-workgroup form)    */
    float actor_obs_scale[5];/* the input scaling folded into that block's first layer                                                    */
    int32_t reserved0;       /* 0                                                                                                          */
} MrsimDdpgLearner;
#define MRSIM_DDPG_BATCH_SCRATCH_FLOATS(batch) ((int64_t)((batch) / 64) * (MRSIM_DDPG_PARAMS + 4) + (batch) + 64)
/* One update on `batch` transitions (a multiple of 64, <= MRSIM_DDPG_MAX_BATCH; the reference uses 64; batches above 64 spread over
 * batch / 64 compute units when learner->batch_scratch is given, see there).  s / s2 [.][5], a [.][2],
 * r [.], done [.] are DEVICE arrays (e.g. the replay ring).  Which rows: idx (DEVICE [batch] int32) if given; else, with
 * ring_count > 0, drawn IN the kernel from [0, ring_count) by Philox4x32-10 keyed by (seed, draw_counter) -- without
 * repetition, the law of random.sample (RL/MR_ddpg.py:37-44), for batch <= 256 and batch <= ring_count (a partial Fisher-Yates
 * shuffle while the ring holds fewer than two batches, redrawn duplicates beyond), with repetition otherwise;
 * else rows 0 .. batch-1.  n_updates >= 1 consecutive updates run in this ONE launch (update i draws with draw_counter + i; with idx
 * or fixed rows every update sees the same batch): the learner then occupies one compute unit for the whole burst instead of
 * queueing n launches behind a device full of env kernels.  idx_out: optional DEVICE [batch] int32, the rows of the last update.
 * losses_out: optional DEVICE [2] {critic loss, actor loss} of the last update.  Everything is enqueued on `stream`; no host
 * synchronisation. */
int mrsim_ddpg_update(const MrsimDdpgLearner* learner, int32_t batch, int32_t n_updates, const float* s, const float* a,
                      const float* r, const float* done, const float* s2, const int32_t* idx, int32_t ring_count, uint64_t seed,
                      uint64_t draw_counter, int32_t* idx_out, float* losses_out, void* stream);

/* Replay-ring feed for the learner (RL/MR_ddpg.py:279-281 replay_buffer.add, for a sampled subset of one collected launch group):
 * n transitions (t, env) drawn by Philox keyed by (seed, draw_counter) from the resident [T][N][.] outputs of mrsim_rollout are
 * written to ring rows head .. head + n - 1 (mod capacity).  state = the observation the action was computed from (prev_obs [N][5]
 * for t = 0, obs_T[t-1] otherwise) x obs_scale, next state = obs_T[t] x obs_scale.  [N][5] observation layout.  All DEVICE. */
int mrsim_replay_push(int64_t n_envs, int32_t T, const float* obs_T, const float* actions_T, const float* rew_T,
                      const uint8_t* done_T, const float* prev_obs, const float* obs_scale5_host, int32_t n, float* ring_s,
                      float* ring_a, float* ring_r, float* ring_done, float* ring_s2, int32_t capacity, int32_t head,
                      uint64_t seed, uint64_t draw_counter, void* stream);
/* The per-step bookkeeping of the reference's loop as ONE launch (ABI 5) -- RL/MR_ddpg.py:278-282 `replay_buffer.add(state, action,
 * reward, terminal, next_state)` for the n transitions of one lockstep env step, :307 `state = next_state`, :309-311 the return of
 * the episodes that ended: ring rows head .. head + n - 1 (mod capacity; the last `capacity` envs when n > capacity) receive
 * s = obs_prev x obs_scale, a = actions, r = rew, done, s2 = (done and final_obs given ? final_obs : obs_next) x obs_scale -- with
 * MrsimParams.auto_reset the env's own observation of a finished env is already the next episode's reset row, the transition's s2
 * is MrsimStepIO.final_obs; obs_prev_out[n][5] := obs_next (may be obs_prev itself); ended2[2] += {sum of final_ret over the envs
 * that are done, their number} (optional).  [n][5] observation rows.  All DEVICE except obs_scale5_host. */
int mrsim_replay_add_step(int64_t n, const float* obs_prev, const float* actions, const float* rew, const uint8_t* done,
                          const float* obs_next, const float* final_obs, const float* final_ret, const float* obs_scale5_host,
                          float* ring_s, float* ring_a, float* ring_r, float* ring_done, float* ring_s2, int32_t capacity, int32_t head,
                          float* obs_prev_out, float* ended2, void* stream);
/* The learner's online actor (MrsimDdpgLearner.online) -> the packed block of MrsimActor.blob, on the device: batch norm folded
 * with the given moving statistics (bn_stats: the learner's, its online actor's two layers), weights permuted and split exactly
 * as mrsim_actor_fold_bn_host + mrsim_actor_pack_host do (bit-identical block).  One launch on `stream`; order it before the
 * launches that read the block. */
int mrsim_actor_pack_device(const float* learner_online, const float* bn_stats, float bn_eps, const float* obs_scale5_host,
                            const float* action_bound2_host, float* blob, void* stream);

int mrsim_abi_version(void);
const char* mrsim_strerror(int code);

/* MR_Env.__init__ constants + MR_Env.reset defaults (MR_env.py:34-45,56-63,164-170). */
int mrsim_default_params(MrsimParams* p);

/* MR_Env.reset(init, noise_var, a0, is_mismatched) for the envs whose mask byte is non-zero
 * (mask NULL = all) -- MR_env.py:164-201 -> Simulator.reset_start_pos MR_simulator.py:21-34.
 * init_xy: optional [n][2] fp64 start positions; NULL = init_space.sample() (MR_env.py:173).
 * ctor_mismatched: law used by the RK45 constructor's two RHS evaluations inside reset
 *   (0 on a fresh env: MR_env.py:181-183 sets is_mismatched AFTER reset_start_pos).
 * obs: reset observation, layout per params (may be NULL). */
int mrsim_reset(const MrsimParams* p, int64_t n, uint32_t env_id0, const MrsimState* st,
                const uint8_t* mask, const double* init_xy, const float* goal_table, float* obs,
                int32_t ctor_mismatched, uint64_t seed, uint64_t step_idx, void* stream);

/* MR_Env.step(action) for n envs -- MR_env.py:70-98 -> Simulator.step MR_simulator.py:36-52
 * (+ scipy RK45), convert_state :100-116, end :136-152, reward :89 / :118-134. */
int mrsim_step(const MrsimParams* p, int64_t n, uint32_t env_id0, const MrsimState* st,
               const MrsimStepIO* io, uint64_t seed, uint64_t step_idx, void* stream);

/* Uniform random policy (the DDPG warm-up / exploration workload, RL/MR_ddpg.py:277):
 * actions[n][2] ~ U[act_low, act_high). */
int mrsim_random_policy(const MrsimParams* p, int64_t n, uint32_t env_id0, float* actions,
                        uint64_t seed, uint64_t step_idx, void* stream);

/* The same policy for T consecutive steps in one launch: actions_T[t][n][2], row t bit-identical
 * to mrsim_random_policy(..., step_idx0 + t).  The exploration policy does not read the state, so
 * a whole episode's actions can be drawn ahead of its steps (MR_ddpg.py:277 draws them one by
 * one; the values are the same).  T <= 65535. */
int mrsim_random_policy_steps(const MrsimParams* p, int64_t n, uint32_t env_id0, float* actions_T,
                              int32_t T, uint64_t seed, uint64_t step_idx0, void* stream);

/* Inputs / outputs of a fused rollout.  Optional pointers may be NULL. */
typedef struct MrsimRolloutIO {
    int32_t T;               /* steps in this launch; step_idx0 .. step_idx0+T-1 are consumed          */
    int32_t shared_actions;  /* non-zero: actions is [T][2], one table for all envs (utils.run_sim)    */
    const void* actions;     /* [T][n][2] per-env, [T][2] shared (fp32, or fp64 when actions_f64), or NULL =   */
                             /*   in-kernel random policy                                                      */
    const float* goal_table; /* optional [K][T'][2]                                                    */
    double* traj_xy;         /* optional [T][n][2] fp64: position after each step, before any auto-    */
                             /*   reset == what utils.run_sim records from env.last_pos (utils.py:53)  */
    float* state_prime_T;    /* optional [T][n][2]: env.state_prime after each step (utils.py:54)      */
    float* obs_T;            /* optional [T][n][5] or [T][5][n] (params.obs_layout)                    */
    float* rew_T;            /* optional [T][n]                                                        */
    uint8_t* done_T;         /* optional [T][n]                                                        */
    float* actions_out_T;    /* optional [T][n][2]: the actions applied                                */
    float* final_ret;        /* optional [n]: return of the latest episode that ended in the launch    */
    int32_t* final_len;      /* optional [n]: its length                                               */
    int32_t* status;         /* optional [1], as MrsimStepIO.status                                    */
    int64_t row_stride;      /* envs per row of every [T][.] buffer above; 0 = n.  A value > n lets several     */
                             /*   launches (sub-shards of one env set, e.g. on different streams) write their  */
                             /*   columns of the same [T][row_stride][.] buffers: pass each launch the buffer  */
                             /*   pointers advanced to its first env.  final_ret / final_len stay [n].         */
    int32_t carry_f64;       /* 0 (default): the RK45 object's carried state (integrator.f, h_abs) is rounded  */
                             /*   to its fp32 HBM format after every step, so the launch is bit-identical to T */
                             /*   calls of mrsim_step.  1: it stays in fp64 registers until the launch ends    */
                             /*   (closer to the reference, which carries fp64; not bit-identical to steps).   */
    int32_t actions_f64;     /* non-zero: `actions` holds fp64 values ([T][2] or [T][n][2] doubles) -- the     */
                             /*   reference's action tables (main.py:14-50) are float64 linspace tables        */
    const MrsimActor* actor; /* optional (HOST pointer; ABI 3): every step's action is actor.predict(obs) +    */
                             /*   actor_noise() evaluated in-kernel on the observation of the previous step:   */
                             /*   the collection loop of RL/MR_ddpg.py:270-311 in one launch.  actions must be */
                             /*   NULL and the integrator RK45.  Bit-identical to T x (mrsim_actor_forward ->  */
                             /*   mrsim_step) with carry_f64 = 0.                                              */
} MrsimRolloutIO;

/* Fused open-loop rollout, the batched utils.run_sim (utils.py:43-61) and the DDPG rollout workload:
 * T steps of all n envs in one launch, env state kept in registers, every requested per-step output
 * written to [T][n][...] buffers.  Honors auto_reset.  Bit-identical to T calls of mrsim_step (carry_f64 = 0). */
int mrsim_rollout(const MrsimParams* p, int64_t n, uint32_t env_id0, const MrsimState* st,
                  const MrsimRolloutIO* io, uint64_t seed, uint64_t step_idx0, void* stream);

/* actions[n][2] = actor.predict(obs) + actor_noise() (RL/MR_ddpg.py:277) as a kernel of its own -- the gym-loop form:
 * feed `actions` to mrsim_step with the SAME (seed, step_idx).  obs: [n][5] or [5][n] per p->obs_layout (what
 * mrsim_reset / mrsim_step wrote).  st: the env state, read only when actor->ou_reset_on_done (MR_Env.counter); may be
 * NULL otherwise. */
int mrsim_actor_forward(const MrsimParams* p, int64_t n, uint32_t env_id0, const MrsimActor* actor,
                        const MrsimState* st, const float* obs, float* actions, uint64_t seed, uint64_t step_idx,
                        void* stream);

/* Velocity post-processing that every consumer of run_sim applies to the recorded positions
 * (Learning_module.py:46-59,72-93; main.py:102-109):
 *     p  = uniform_filter1d(p, N, mode="nearest")           (running mean, window [t - N/2, t + N - N/2 - 1])
 *     v  = np.gradient(p, time)                             (second-order interior, first-order edges)
 *     v  = uniform_filter1d(v, N/2, mode="nearest")
 *     D  = mean(v[N:-N])                                    (estimateDisturbance's drift; 0 if T <= 2N)
 * for n trajectories at once, one lane per trajectory streaming along t.  traj_xy: [T][n][2] fp64 (the
 * layout mrsim_rollout writes); time: [T] fp64; v_xy: [T][n][2] fp64 out; scratch_xy: [T][n][2] fp64 work
 * buffer; drift_xy: optional [n][2] fp64 out.  n_filter = N (the reference uses int(1/0.035/2) = 14). */
int mrsim_velocity(int64_t n, int32_t T, int32_t n_filter, const double* traj_xy, const double* time,
                   double* v_xy, double* scratch_xy, double* drift_xy, void* stream);

/* *step_base += delta on `stream` (a one-lane kernel; graph-capturable). */
int mrsim_advance_step_base(uint64_t* step_base, uint64_t delta, void* stream);

/* Streams confined to a subset of the device's compute units (hipExtStreamCreateWithCUMask), for running the learner beside
 * the collection launches: one fused-rollout launch group of 262 144 envs is exactly one resident round of workgroups, so a kernel
 * enqueued on any ordinary stream meanwhile (the learner's one-workgroup update, the replay push, the parameter upload) only finds a
 * unit at a launch boundary and then displaces collection blocks for its whole duration.  With the collection streams masked off
 * a few units and the learner's stream confined to those, neither waits for the other (mr_rl_amd/partition.py; RL/MR_ddpg.py:270-311
 * is one loop -- this is what lets its two halves overlap on one GPU).
 * mrsim_device_cu_layout: compute units of `device` and the number of XCCs mask bits are dealt over -- bit i of the mask is a
 *   unit of XCC i % xccs (measured on MI355X: tools/cumask_probe.hip; workgroups are dealt round robin over the XCCs, so a mask must
 *   keep at least one unit in EVERY XCC or a workgroup can be dealt where the queue has none -- mrsim_stream_create_cu_mask refuses
 *   such masks with MRSIM_EINVAL).
 * mrsim_stream_create_cu_mask: mask = n_words 32-bit words, bit set = unit usable; *stream_out receives a hipStream_t.
 * mrsim_stream_destroy: synchronises and destroys such a stream. */
int mrsim_device_cu_layout(int32_t device, int32_t* compute_units, int32_t* xccs);
int mrsim_stream_create_cu_mask(int32_t device, const uint32_t* mask, int32_t n_words, void** stream_out);
int mrsim_stream_destroy(void* stream);

/* Pinned, device-mapped host memory (hipHostMalloc, mapped + coherent) for callers that drive FEW envs from the host -- the
 * single-env MR_Env facade (MR_env.py:70-98 called once per Python loop iteration, utils.py:51-54, RL/MR_ddpg.py:278): the env
 * state, the action and every output of mrsim_step live in ONE such block, the kernel reads and writes it over the bus, and a step
 * is one launch + one mrsim_stream_synchronize with no copy call on either side of it.  *dev_ptr_out is the address to hand to the
 * entry points above (equal to *host_ptr_out under unified addressing; returned separately so that nothing is assumed).
 * For many envs keep state and outputs in device memory: every access to such a block crosses PCIe. */
int mrsim_host_alloc(int64_t bytes, void** host_ptr_out, void** dev_ptr_out);
int mrsim_host_free(void* host_ptr);
/* hipStreamSynchronize(stream) for callers that bind no HIP runtime of their own (ctypes). */
int mrsim_stream_synchronize(void* stream);
/* Spin (on the calling host thread, no HIP call) until *host_word == value: the host side of MrsimStepIO.done_word.  host_word is
 * the HOST address of a word in mrsim_host_alloc memory.  Returns MRSIM_OK, or MRSIM_ETIMEOUT after timeout_us microseconds (the
 * launch is then still in flight or has failed: mrsim_stream_synchronize tells which).  Everything the launch wrote before the
 * word is visible to the caller on return (acquire). */
int mrsim_host_wait_word(const int32_t* host_word, int32_t value, int64_t timeout_us);

/* Number of HIP devices visible (0 without a GPU); fills name_host (may be NULL). */
int mrsim_device_count(void);
int mrsim_device_name(int device, char* name_host, int32_t len);

#ifdef __cplusplus
}
#endif
#endif /* MRSIM_H */
