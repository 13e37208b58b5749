/*
 * mrsim_oracle.h -- CPU ORACLE for the MR_env.step()/MR_simulator hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, fp64, one-env-at-a-time
 * restatement of the reference algorithm; it is the checker the HIP path is
 * compared against.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (mr_rl_amd/) never does.
 *
 * Parity status: PINNED for sigma = 0 and for the noise plumbing (draw order,
 * stage weights, error control) by golden vectors generated from the reference
 * itself (tests/golden/make_golden.py -> ref_sim.npz, ref_noise.npz,
 * ref_env.npz).  The Philox/Box-Muller generator that replaces NumPy's global
 * MT19937 stream is this build's own definition (the reference's stream cannot
 * be reproduced on a GPU); it is checked against Random123 known-answer
 * vectors and distribution tests, not against the reference.
 *
 * Reference files restated (citations are /root/reference/<file>:<line>):
 *   MR_simulator.py:8-94   Simulator (RHS `simulate`, `step`, `reset_start_pos`)
 *   MR_env.py:34-45,56-63  spaces and constants
 *   MR_env.py:70-201       step / convert_state / calculate_reward / end / reset
 * Third-party arithmetic on the path, restated from its published algorithm
 * (SciPy is a dependency of the reference, version unpinned there; 1.15.3 here):
 *   scipy/integrate/_ivp/rk.py      rk_step, RungeKutta.__init__/_step_impl, RK45 tableau
 *   scipy/integrate/_ivp/common.py  select_initial_step, norm
 *   scipy/integrate/_ivp/base.py    OdeSolver.step
 */
#ifndef MRSIM_ORACLE_H
#define MRSIM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_INT_RK45 = 0, ORC_INT_EULER = 1, ORC_INT_RK4 = 2 };
enum { ORC_REW_CONSTANT10 = 0, ORC_REW_GOAL = 1 };
enum { ORC_NOISE_NONE = 0, ORC_NOISE_PHILOX = 1, ORC_NOISE_TAPE = 2 };
/* where the Philox-sourced normals of an RK45 step enter (mrsim_oracle.c, COL_*): PER_STAGE = one draw per RHS evaluation as
 * MR_simulator.py:73-83 does (the parity mode); COLLAPSED = the B- and E-weighted stage sums drawn directly from their joint
 * Gaussian -- equal in law, fewer draws (build extension; RK45 + Philox only, a tape replay and sigma = 0 ignore it) */
enum { ORC_LAW_PER_STAGE = 0, ORC_LAW_COLLAPSED = 1 };

typedef struct {
    double time_span;      /* 0.030                      MR_simulator.py:12            */
    double rtol;           /* time_span/100 = 3e-4       MR_simulator.py:13,91         */
    double atol;           /* 1e-4                       MR_simulator.py:91            */
    double a0;             /* Simulator.a0               MR_env.py:180                 */
    double sigma;          /* Simulator.noise_var (used as a std-dev) MR_simulator.py:73 */
    double min_dist2goal;  /* 30                         MR_env.py:63                  */
    double obs_low[5];     /* observation_space          MR_env.py:37-39               */
    double obs_high[5];
    double init_low[2];    /* init_space                 MR_env.py:40-42               */
    double init_high[2];
    int32_t mismatched;    /* Simulator.is_mismatched    MR_env.py:183                 */
    int32_t integrator;    /* ORC_INT_*  (RK45 = what the reference does)              */
    int32_t substeps;      /* fixed-step modes only                                     */
    int32_t reward_mode;   /* ORC_REW_*  (constant 10 = MR_env.py:89)                  */
    int32_t max_timesteps; /* 50                         MR_env.py:62                  */
    int32_t auto_reset;    /* build extension: re-draw init and reset in the same step  */
    int32_t goal_K;        /* goal table [K][T][2] (float); K = T = 1, (0,0) = reference */
    int32_t goal_T;
    int32_t auto_reset_fresh_env; /* 0: an auto-reset is reset() on the SAME env object (RL/MR_ddpg.py:270): the RK45   */
                           /*    constructor runs under the law the previous episode left (MR_env.py:181-183);     */
                           /*    1: a fresh MR_Env per episode (nominal-law constructor)                           */
    int32_t noise_law;     /* ORC_LAW_*                                                  */
} OrcParams;

/* One environment = one MR_Env + its Simulator + its live RK45 object. */
typedef struct {
    double y[2];           /* integrator.y == Simulator.last_state                      */
    double t;              /* integrator.t                                              */
    double f[2];           /* integrator.f   (stage K[0] of the next rk_step)           */
    double h_abs;          /* integrator.h_abs                                          */
    double state_prime[2]; /* Simulator.state_prime = last RHS value (MR_simulator.py:87) */
    double ep_ret;         /* sum of rewards this episode (build extension)             */
    double err_norm0;      /* diagnostic: error_norm of the last step's FIRST rk_step attempt                  */
    double err_margin;     /* diagnostic: min over the last step's rk_step attempts of |error_norm - 1|, the   */
                           /*   distance of the accept / reject decision from its discontinuity (tests use it  */
                           /*   to tell a genuine decision flip from a kernel bug)                             */
    int32_t counter;       /* MR_Env.counter                                            */
    int32_t n_rhs;         /* RHS evaluations in the last step (diagnostic)             */
    int32_t n_attempts;    /* rk_step attempts in the last step (diagnostic)            */
    int32_t status;        /* 0 ok, <0 the reference would have raised                  */
} OrcEnv;

/* Noise source for one call. */
typedef struct {
    int32_t kind;          /* ORC_NOISE_*                                               */
    uint64_t seed;         /* PHILOX: key                                               */
    uint64_t step_idx;     /* PHILOX: global step index (counter words 1,2)             */
    const double* tape;    /* TAPE: values exactly as numpy.random.normal returned them */
    int64_t tape_len;
    int64_t tape_pos;      /* in/out                                                    */
} OrcNoise;

void orc_default_params(OrcParams* p);
/* {cB, cE1, cE2} of the collapsed law recomputed from the tableau (pins the literals in mrsim_oracle.c / mrsim_device.h) */
void orc_collapsed_constants(double out[3]);

/* RNG definition (build's own; shared by spec with the HIP kernel) */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_box_muller(uint32_t ua, uint32_t ub, float* z0, float* z1);
/* 4 standard normals of call `c0` for (seed, env_id, step_idx) */
void orc_normals4(uint64_t seed, uint32_t env_id, uint64_t step_idx, uint32_t c0, float z[4]);
void orc_fill_normals(uint64_t seed, uint32_t env_id, uint64_t step_idx, uint32_t c0_start, int64_t ncalls,
                      float* out);
/* 2 uniforms in (0,1) (fp64) of call c0 */
void orc_uniform2(uint64_t seed, uint32_t env_id, uint64_t step_idx, uint32_t c0, double u[2]);

/* Simulator-level (MR_simulator.py) */
void orc_sim_reset(const OrcParams* p, OrcEnv* e, double x0, double y0, int ctor_mismatched,
                   OrcNoise* nz, uint32_t env_id, int ctor_stream);
int  orc_sim_step(const OrcParams* p, OrcEnv* e, double f_t, double alpha_t, OrcNoise* nz, uint32_t env_id);

/* Env-level (MR_env.py): one step of one env.  obs[5] fp64 as the reference returns it. */
int  orc_env_step(const OrcParams* p, OrcEnv* e, const float* goal_table, double f_t, double alpha_t,
                  OrcNoise* nz, uint32_t env_id, double obs[5], double* rew, uint8_t* done,
                  double final_obs[5], double* final_ret, int32_t* final_len);
void orc_env_reset(const OrcParams* p, OrcEnv* e, const float* goal_table, double x0, double y0,
                   int ctor_mismatched, OrcNoise* nz, uint32_t env_id, double obs[5]);
void orc_convert_state(const double state[2], const double goal[2], double obs[5]);
double orc_calculate_reward(const OrcParams* p, const double obs[5], int32_t counter);
int  orc_end(const OrcParams* p, const double obs[5], int32_t counter);
void orc_goal_at(const OrcParams* p, const float* goal_table, uint32_t env_id, int32_t counter, double goal[2]);
void orc_sample_init(const OrcParams* p, uint64_t seed, uint32_t env_id, uint64_t step_idx, double xy[2]);
void orc_random_action(int integrator, uint64_t seed, uint32_t env_id, uint64_t step_idx, const double lo[2],
                       const double hi[2], float act[2]);

/* Batched drivers over n envs (global ids env_id0..env_id0+n-1); PHILOX noise or none.
 * threads <= 1: serial.  Used by tests and as bench.py's cpu_baseline ("port"). */
int orc_vec_reset(const OrcParams* p, int64_t n, uint32_t env_id0, OrcEnv* envs, const float* goal_table,
                  const double* init_xy /* [n][2] or NULL = sample */, uint64_t seed, uint64_t step_idx,
                  double* obs /* [n][5] */, int threads);
int orc_vec_step(const OrcParams* p, int64_t n, uint32_t env_id0, OrcEnv* envs, const float* goal_table,
                 const float* actions /* [n][2] */, uint64_t seed, uint64_t step_idx,
                 double* obs /* [n][5] */, double* rew, uint8_t* done,
                 double* final_obs /* [n][5] or NULL */, double* final_ret, int32_t* final_len, int threads);
int orc_vec_random_policy(int integrator, int64_t n, uint32_t env_id0, uint64_t seed, uint64_t step_idx,
                          const double lo[2], const double hi[2], float* actions, int threads);
/* ---- DDPG actor + OU noise as the policy (RL/MR_ddpg.py:80-160 ActorNetwork.predict, :59-78 OUNoise, :277).
 * Inference form: 5 -> 64 (+ folded batch norm) -> relu -> 64 (+ folded batch norm) -> relu -> 2 tanh, * bound.
 * Parity status: UNPINNED against the reference (TensorFlow 1.x / tflearn are absent); tests pin this restatement
 * against the PyTorch twin's fp32 forward (mr_rl_amd/ddpg.py: Actor, eval mode).  The fp32 summation order is part of
 * the definition (mr_rl_amd/csrc/mrsim_actor.h), so the HIP kernels agree with this function bit for bit. */
typedef struct {
    float w1[64 * 5];      /* [64][5]  (obs scaling already folded in)                   */
    float b1[64];
    float w2[64 * 64];     /* [64][64]                                                   */
    float b2[64];
    float w3[2 * 64];      /* [2][64]                                                    */
    float b3[2];
    float bound[2];        /* action_bound                                               */
    float ou_theta_dt;     /* theta * dt           (products formed in double, rounded)  */
    float ou_sigma_sqrt_dt;/* sigma * sqrt(dt)                                           */
    int32_t ou_enabled;    /* 0: actor.predict alone                                     */
    int32_t ou_reset_on_done; /* 1: x_prev := 0 at the first step of an episode (counter == 0) */
    int32_t math;          /* 0: f32 fmaf chains (bitwise = the kernel's f32 MFMA); 2: plain bf16 operands; 1: bf16 x 3 */
                           /*    arithmetic -- the same operand splits as the kernel, products exact, the 16 products of  */
                           /*    one MFMA summed exactly and added to the f32 accumulator with one rounding (the          */
                           /*    hardware's internal order is not documented: compared with a 2e-6 tolerance)             */
    int32_t reserved0;
    float w2_split[64 * 64 * 3]; /* [f][k][term]: the bf16 terms of w2, filled by orc_actor_prepare                   */
    float w1_split[64 * 5 * 3];  /* [f][k][term]: the bf16 terms of w1 (scaled), filled by orc_actor_prepare         */
} OrcActor;
void orc_actor_prepare(OrcActor* a);
float orc_spec_tanhf(float x);
void orc_actor_forward(const OrcActor* a, const float obs[5], float act[2]);
/* action = actor.predict(obs) + actor_noise(); ou[2] is OUNoise.x_prev, updated in place.  The OU pair's normals are
 * words 0,1 of DYN(0,0) (RK45) / POLICY(0,0) (fixed-step) of (seed, env_id, step_idx), spec Box-Muller. */
void orc_actor_policy(const OrcActor* a, int integrator, const float obs[5], int32_t counter, float ou[2], uint64_t seed,
                      uint32_t env_id, uint64_t step_idx, float act[2]);
int orc_vec_actor_policy(const OrcActor* a, int integrator, int64_t n, uint32_t env_id0, const float* obs /* [n][5] */,
                         const int32_t* counter /* [n] or NULL */, float* ou /* [n][2] or NULL */, uint64_t seed,
                         uint64_t step_idx, float* actions /* [n][2] */, int threads);
int orc_sizeof_actor(void);

int orc_num_threads(void);
int orc_sizeof_env(void);
int orc_sizeof_params(void);

#ifdef __cplusplus
}
#endif
#endif
