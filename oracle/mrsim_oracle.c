/*
 * mrsim_oracle.c -- CPU ORACLE (test infrastructure, never shipped or measured
 * as the product).  See mrsim_oracle.h for scope, parity status and the list of
 * reference files restated.  Citations: /root/reference/<file>:<line>; SciPy
 * citations name the function in scipy/integrate/_ivp/{rk,common,base}.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 * -ffp-contract=off matters: the Box-Muller below is specified with explicit
 * fmaf() so that it is bit-identical to the HIP kernel's.
 */
#include "mrsim_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* defaults: MR_env.py:34-45,62-63 and MR_simulator.py:12-13,90-91           */
/* ------------------------------------------------------------------------ */
void orc_default_params(OrcParams* p) {
    memset(p, 0, sizeof(*p));
    p->time_span = 0.030;                 /* MR_simulator.py:12 */
    p->rtol = p->time_span / 100;         /* MR_simulator.py:13,91 */
    p->atol = 1e-4;                       /* MR_simulator.py:91 */
    p->a0 = 1.0;                          /* MR_env.py:168 (reset default) */
    p->sigma = 1.0;                       /* MR_env.py:167 (reset default noise_var=1) */
    p->min_dist2goal = 30.0;              /* MR_env.py:63 */
    const double lo[5] = {-5000, -5000, -5000, -5000, 0};      /* MR_env.py:38 */
    const double hi[5] = {5000, 5000, 5000, 5000, 80000};      /* MR_env.py:39 */
    memcpy(p->obs_low, lo, sizeof lo);
    memcpy(p->obs_high, hi, sizeof hi);
    p->init_low[0] = p->init_low[1] = 100.0;                   /* MR_env.py:41 */
    p->init_high[0] = p->init_high[1] = 120.0;                 /* MR_env.py:42 */
    p->mismatched = 0;
    p->integrator = ORC_INT_RK45;
    p->substeps = 1;
    p->reward_mode = ORC_REW_CONSTANT10;  /* MR_env.py:89 */
    p->max_timesteps = 50;                /* MR_env.py:62 */
    p->auto_reset = 0;
    p->goal_K = 1;
    p->goal_T = 1;
}

int orc_sizeof_env(void) { return (int)sizeof(OrcEnv); }
int orc_sizeof_params(void) { return (int)sizeof(OrcParams); }
int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* RNG definition (this build's; replaces numpy's global MT19937 stream).    */
/* Philox4x32-10 (Salmon et al., SC'11; Random123 constants).                */
/* counter = {c0, step_idx lo, step_idx hi, global env id}, key = seed.      */
/* c0 = stream<<28 | block<<4 | call.                                        */
/* ------------------------------------------------------------------------ */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

enum { STREAM_DYN = 0, STREAM_CTOR = 1, STREAM_RESET_POS = 2, STREAM_RESET_CTOR = 3, STREAM_POLICY = 4 };
#define C0(stream, block, call) (((uint32_t)(stream) << 28) | ((uint32_t)(block) << 4) | (uint32_t)(call))

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ln(u) for u in [2^-33, 1], fp32, explicit fma; Cephes logf polynomial. */
static float spec_logf(float u) {
    uint32_t b = f2u(u);
    int e = (int)(b >> 23) - 127;
    float m = u2f((b & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m *= 0.5f; e += 1; }
    float t = m - 1.0f;
    float z = t * t;
    float p = 7.0376836292E-2f;
    p = fmaf(p, t, -1.1514610310E-1f);
    p = fmaf(p, t, 1.1676998740E-1f);
    p = fmaf(p, t, -1.2420140846E-1f);
    p = fmaf(p, t, 1.4249322787E-1f);
    p = fmaf(p, t, -1.6668057665E-1f);
    p = fmaf(p, t, 2.0000714765E-1f);
    p = fmaf(p, t, -2.4999993993E-1f);
    p = fmaf(p, t, 3.3333331174E-1f);
    float y = (t * z) * p;
    y = fmaf(-0.5f, z, y);
    float lm = t + y;
    return fmaf((float)e, 0.693147180559945f, lm);
}

/* Box-Muller, fully specified in fp32 so that CPU and GPU agree bit for bit:
 *   u     = fma(float(ua), 2^-32, 2^-33)            in (0,1]
 *   r     = sqrt(-2 ln u)                            (IEEE sqrt)
 *   theta = 2 pi (ub + 0.5) / 2^32, split in octants; the in-octant angle
 *           phi in (0, pi/4] goes through Cephes sinf/cosf polynomials.   */
void orc_box_muller(uint32_t ua, uint32_t ub, float* z0, float* z1) {
    float u = fmaf((float)ua, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    float r = sqrtf(-2.0f * spec_logf(u));
    uint32_t oct = ub >> 29;
    uint32_t rem = ub & 0x1FFFFFFFu;
    if (oct & 1u) rem = 0x1FFFFFFFu - rem;
    float x = fmaf((float)rem, 1.862645149230957e-09f, 9.313225746154785e-10f); /* 2^-29, 2^-30 */
    float phi = x * 0.78539816339744831f;
    float zz = phi * phi;
    float ps = -1.9515295891E-4f;
    ps = fmaf(ps, zz, 8.3321608736E-3f);
    ps = fmaf(ps, zz, -1.6666654611E-1f);
    float s = fmaf(ps * zz, phi, phi);
    float pc = 2.443315711809948E-005f;
    pc = fmaf(pc, zz, -1.388731625493765E-003f);
    pc = fmaf(pc, zz, 4.166664568298827E-002f);
    float c = fmaf(pc * zz, zz, fmaf(-0.5f, zz, 1.0f));
    uint32_t swap = ((oct + 1u) >> 1) & 1u;
    uint32_t cneg = ((oct + 2u) >> 2) & 1u;
    uint32_t sneg = oct >> 2;
    float cc = swap ? s : c;
    float ss = swap ? c : s;
    if (cneg) cc = -cc;
    if (sneg) ss = -ss;
    *z0 = r * cc;
    *z1 = r * ss;
}

static void philox_call(uint64_t seed, uint32_t env_id, uint64_t step_idx, uint32_t c0, uint32_t out[4]) {
    uint32_t ctr[4] = {c0, (uint32_t)step_idx, (uint32_t)(step_idx >> 32), env_id};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_philox4x32_10(ctr, key, out);
}

void orc_normals4(uint64_t seed, uint32_t env_id, uint64_t step_idx, uint32_t c0, float z[4]) {
    uint32_t r[4];
    philox_call(seed, env_id, step_idx, c0, r);
    orc_box_muller(r[0], r[1], &z[0], &z[1]);
    orc_box_muller(r[2], r[3], &z[2], &z[3]);
}

/* bulk draw for distribution tests: calls c0_start .. c0_start+ncalls-1 */
void orc_fill_normals(uint64_t seed, uint32_t env_id, uint64_t step_idx, uint32_t c0_start, int64_t ncalls,
                      float* out) {
    for (int64_t i = 0; i < ncalls; ++i) orc_normals4(seed, env_id, step_idx, c0_start + (uint32_t)i, out + 4 * i);
}

void orc_uniform2(uint64_t seed, uint32_t env_id, uint64_t step_idx, uint32_t c0, double u[2]) {
    uint32_t r[4];
    philox_call(seed, env_id, step_idx, c0, r);
    u[0] = ((double)r[0] + 0.5) * 2.3283064365386963e-10; /* exact in fp64 */
    u[1] = ((double)r[1] + 0.5) * 2.3283064365386963e-10;
}

/* Sequential stream of N(0,1) draws inside one block (one rk_step attempt,
 * one constructor, one fixed substep): draw q comes from call q>>2, lane q&3. */
typedef struct {
    OrcNoise* nz;
    uint32_t env_id;
    uint32_t base; /* C0(stream, block, 0) */
    int pos;
    int cached_call; /* call whose 4 normals are in z[], -1 = none */
    float z[4];
} NStream;

static void ns_open(NStream* s, OrcNoise* nz, uint32_t env_id, int stream, uint32_t block) {
    s->nz = nz; s->env_id = env_id; s->base = C0(stream, block, 0); s->pos = 0; s->cached_call = -1;
}

/* jump to draw index `pos` of the block (PHILOX source only; a tape is always consumed in order) */
static void ns_seek(NStream* s, int pos) { s->pos = pos; }

/* numpy.random.normal(0, scale, 1)[0] == 0 + scale * z   (MR_simulator.py:56,79-83) */
static double ns_normal(NStream* s, double scale) {
    OrcNoise* nz = s->nz;
    if (nz->kind == ORC_NOISE_TAPE) {
        if (nz->tape_pos >= nz->tape_len) { nz->tape_pos++; return NAN; }
        return nz->tape[nz->tape_pos++]; /* already loc + scale*z */
    }
    if (nz->kind == ORC_NOISE_NONE || scale == 0.0) { s->pos++; return 0.0; }
    if ((s->pos >> 2) != s->cached_call) {
        s->cached_call = s->pos >> 2;
        orc_normals4(nz->seed, s->env_id, nz->step_idx, s->base | (uint32_t)s->cached_call, s->z);
    }
    double z = (double)s->z[s->pos & 3];
    s->pos++;
    return scale * z;
}

/* ------------------------------------------------------------------------ */
/* Simulator.simulate -- the ODE right-hand side.  MR_simulator.py:58-88      */
/* Ignores t and the state; side effect: state_prime = fx (:87).             */
/* ------------------------------------------------------------------------ */
static void simulate(const OrcParams* p, int mismatched, OrcEnv* e, const double act[2], NStream* ns,
                     double fx[2]) {
    const double f_t = act[0], alpha_t = act[1];
    const double sigma = p->sigma;        /* :73 */
    double a0 = p->a0;                    /* :76 */
    double dx1, dx2;
    if (mismatched) {
        /* a0_linear(alpha_t, f_t, sigma/4): a0 + (f/4)*0.8 + N(0, sigma/4)   :55-56,78 */
        a0 = p->a0 + (f_t / 4) * 0.8 + ns_normal(ns, sigma / 4);
        dx1 = a0 * f_t * cos(alpha_t + 0.1) + ns_normal(ns, sigma) + 0.2;   /* :79 */
        dx2 = a0 * f_t * sin(alpha_t - 0.15) + ns_normal(ns, sigma) - 0.1;  /* :80 */
    } else {
        dx1 = a0 * f_t * cos(alpha_t) + ns_normal(ns, sigma);               /* :82 */
        dx2 = a0 * f_t * sin(alpha_t) + ns_normal(ns, sigma);               /* :83 */
    }
    fx[0] = dx1; fx[1] = dx2;
    e->state_prime[0] = dx1; e->state_prime[1] = dx2;                       /* :87 */
    e->n_rhs++;
}

/* scipy common.norm: RMS norm of a 2-vector */
static double rms2(double a, double b) { return sqrt(a * a + b * b) / sqrt(2.0); }

/* RK45 tableau (scipy rk.py class RK45).  A and C never matter here because
 * the RHS ignores (t, y); only B (solution weights) and E (error weights) do. */
static const double RK_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
static const double RK_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525,
                               1.0 / 40};
#define RK_SAFETY 0.9
#define RK_MIN_FACTOR 0.2
#define RK_MAX_FACTOR 10.0
#define RK_ERR_EXP (-1.0 / 5.0) /* -1/(error_estimator_order+1), order 4 */

/* RungeKutta.__init__ (rk.py) as run by Simulator.scipy_runge_kutta (MR_simulator.py:90-91):
 *   self.f = fun(t0, y0); self.h_abs = select_initial_step(...)  (common.py)          */
/* Where the Philox-sourced draws of an RK45 env step live (layout chosen so that the GPU needs three
 * Philox calls per step in the common case; a tape replay ignores it and consumes draws in order).
 *   NOMINAL law (2 draws per RHS evaluation), block = DYN(attempt a), draw index = 4*call + lane:
 *     call 0: [K1 (never reaches a result; for a = 0 its two WORDS carry the exploration policy) | K2]
 *     call 1: [K3 | K4]        call 2: [K5 | F0]        call 3: [K6 | F1]
 *   F0, F1 = the two RHS evaluations of the constructor that runs after the step's LAST attempt; they are
 *   taken from that attempt's block.  K6 (= f_new) and F1 rarely influence a result (K6: only the error
 *   estimate and the next sub-step's K0; F1: only d2 and Simulator.state_prime).
 *   MISMATCHED law (3 draws (z_a, z_x, z_y) per evaluation), same idea over six calls of DYN(attempt a):
 *     draws 0..1: K1 (dead; policy words as above)   2..4: K2   5..7: K3   8..10: K4   11..13: K5
 *     14..16: F0   17..19: K6   20..22: F1   (23 unused)
 *   so that calls 0..4 are needed eagerly, the pair (K6x, K6y) and call 5 (F1) almost never.
 *   Reset constructors.  Nominal law: F0 = draws 2,3 of the RESET_POS call whose words 0,1 are the init position
 *   (orc_sample_init), F1 = draws 0,1 of RESET_CTOR(0) -- an auto-reset then costs the GPU one Philox call, F1 being
 *   needed as rarely as after a step.  Mismatched law: sequential in RESET_CTOR(0). */
#define NOM_POS_K6 12
#define NOM_POS_F0 10
#define NOM_POS_F1 14
#define MIS_POS_K2 2
#define MIS_POS_F0 14
#define MIS_POS_K6 17
#define MIS_POS_F1 20

/* ---- COLLAPSED noise law (OrcParams.noise_law = ORC_LAW_COLLAPSED; build extension, Philox source only).
 * Per rk_step attempt the per-stage normals N_2..N_5 reach a result only through two weighted sums,
 *   S_B = sum_{2..5} B_i N_i (the position)   and   S_E = sum_{2..5} E_i N_i (the error estimate),
 * which per noise component (x, y and, under the mismatched law, the a0 draw) are jointly Gaussian with the
 * fixed covariance of the tableau: var S_B = cB^2, var S_E = cE^2, corr = rho.  The collapsed law draws them
 * directly:  S_B = cB z1,  S_E = cE1 z1 + cE2 z2  (cE1 = rho cE, cE2 = sqrt(1 - rho^2) cE), z1, z2 iid N(0,1)
 * per component.  K6 (= f_new), F0 and F1 (the constructor's two RHS evaluations) keep their own draws.  Equal in
 * law to MR_simulator.py:73-83's per-evaluation noise for everything a step returns or carries (positions,
 * error_norm, accept/reject, step-size factors, integrator.f, h_abs, state_prime); NOT draw for draw: 6 (nominal)
 * / 11 (mismatched) eager normals per attempt instead of 10 / 16.  A tape replay (the reference pin) ignores it.
 *   NOMINAL, block DYN(attempt a), draw = 4*call + lane, Box-Muller pairs = consecutive even/odd draws:
 *     0,1 policy words (dead as normals)   2,3 z1 (x, y)   4,5 F0   6,7 z2   8,9 K6   10,11 F1
 *   MISMATCHED (round 5).  Two more steps of the same idea, both exact in law:
 *   (i) the noise of one RHS evaluation, (g z_a + sigma z_x, g' z_a + sigma z_y) with three normals (MR_simulator.py:55-56,
 *       77-80), is a 2-vector with covariance sigma^2 I + g g^T.  Its symmetric square root is sigma I + c g g^T with
 *       c = 1 / (sigma + sqrt(sigma^2 + |g|^2)), so N = M(u) = sigma u + c g (g . u) for a standard normal PAIR u has exactly
 *       that law: two normals per evaluation instead of three;
 *   (ii) f_new's noise N_6 reaches the attempt's accept / reject decision only through E6 N_6 inside the error estimate, and
 *       anything else (the next sub-step's K0) only when another sub-step follows.  S_E' = S_E + E6 N_6 is drawn directly,
 *       S_E' = M(cE1 u1 + cE2' u2'), cE2' = sqrt(cE2^2 + E6^2); when N_6 itself is needed it is drawn from its conditional
 *       law given S_E':  N_6 = M((E6 / cE2') u2' + (cE2 / cE2') w),  w a fresh pair.
 *   The block then has the nominal law's layout and needs two Philox calls in the common case instead of three:
 *     0,1 policy   2,3 u1   4,5 F0   6,7 u2'   8,9 w   10,11 F1                                                          */
#define COL_CB  0.8641431770614779      /* sqrt(sum_{2..5} B_i^2)                         */
#define COL_CE1 (-0.05097452091652899)  /* sum_{2..5} B_i E_i / cB                        */
#define COL_CE2 0.05594888714408681     /* sqrt(sum_{2..5} E_i^2 - cE1^2)                 */
#define COL_CE2P 0.06128032288313894    /* sqrt(cE2^2 + E6^2): f_new's noise folded into the error sum (mismatched model) */
#define CNOM_POS_Z1 2
#define CNOM_POS_F0 4
#define CNOM_POS_Z2 6
#define CNOM_POS_K6 8
#define CNOM_POS_F1 10
#define CMIS_POS_U1 2
#define CMIS_POS_F0 4
#define CMIS_POS_U2 6
#define CMIS_POS_W 8
#define CMIS_POS_F1 10

void orc_collapsed_constants(double out[3]) {
    /* from the tableau, for the test that pins the literals above */
    double sb = 0, se = 0, be = 0;
    for (int i = 2; i < 6; ++i) { sb += RK_B[i] * RK_B[i]; se += RK_E[i] * RK_E[i]; be += RK_B[i] * RK_E[i]; }
    const double cb = sqrt(sb), ce1 = be / cb;
    out[0] = cb; out[1] = ce1; out[2] = sqrt(se - ce1 * ce1);
}

/* noise-free part V of simulate() and, under the mismatched law, the gain g of the a0 draw:
 * K = V + (g z_a + sigma z_x, g' z_a + sigma z_y)   (MR_simulator.py:55-56,77-83) */
static void rhs_mean(const OrcParams* p, int mismatched, const double act[2], double V[2], double G[2]) {
    const double f_t = act[0], alpha_t = act[1];
    if (mismatched) {
        const double a0 = p->a0 + (f_t / 4) * 0.8;
        V[0] = a0 * f_t * cos(alpha_t + 0.1) + 0.2;
        V[1] = a0 * f_t * sin(alpha_t - 0.15) - 0.1;
        G[0] = (p->sigma / 4) * f_t * cos(alpha_t + 0.1);
        G[1] = (p->sigma / 4) * f_t * sin(alpha_t - 0.15);
    } else {
        V[0] = p->a0 * f_t * cos(alpha_t);
        V[1] = p->a0 * f_t * sin(alpha_t);
        G[0] = G[1] = 0.0;
    }
}

/* N = M(u) = sigma u + c g (g . u): the mismatched model's per-evaluation noise from a standard normal pair (see COL_* above) */
static void noise_2d(const OrcParams* p, const double G[2], const double u[2], double N[2]) {
    const double c = 1.0 / (p->sigma + sqrt(p->sigma * p->sigma + G[0] * G[0] + G[1] * G[1]));
    const double t = c * (G[0] * u[0] + G[1] * u[1]);
    N[0] = p->sigma * u[0] + G[0] * t;
    N[1] = p->sigma * u[1] + G[1] * t;
}

/* one RHS evaluation of the mismatched model under the collapsed law: V + M(u), u = the next two draws */
static void simulate_2d(const OrcParams* p, OrcEnv* e, const double act[2], NStream* ns, double fx[2]) {
    double V[2], G[2], u[2], N[2];
    rhs_mean(p, 1, act, V, G);
    u[0] = ns_normal(ns, 1.0); u[1] = ns_normal(ns, 1.0);
    noise_2d(p, G, u, N);
    fx[0] = V[0] + N[0]; fx[1] = V[1] + N[1];
    e->state_prime[0] = fx[0]; e->state_prime[1] = fx[1];
    e->n_rhs++;
}

/* law2d: the constructor's two evaluations draw pairs through M (mismatched model under the collapsed law, after a step) */
static void rk45_construct_ex(const OrcParams* p, int mismatched, OrcEnv* e, const double act[2], OrcNoise* nz,
                              uint32_t env_id, int stream, uint32_t block, int pos_f0, int pos_f1, int stream_f1, int law2d) {
    NStream ns;
    ns_open(&ns, nz, env_id, stream, block);
    const double t0 = e->t, t_bound = e->t + p->time_span;
    double f0[2], f1[2];
    if (pos_f0 >= 0) ns_seek(&ns, pos_f0);
    if (law2d) simulate_2d(p, e, act, &ns, f0);
    else simulate(p, mismatched, e, act, &ns, f0);
    e->f[0] = f0[0]; e->f[1] = f0[1];
    /* select_initial_step(fun, t0, y0, t_bound, max_step=inf, f0, direction=1, order=4, rtol, atol) */
    const double interval_length = fabs(t_bound - t0);
    const double sc0 = p->atol + fabs(e->y[0]) * p->rtol;
    const double sc1 = p->atol + fabs(e->y[1]) * p->rtol;
    const double d0 = rms2(e->y[0] / sc0, e->y[1] / sc1);
    const double d1 = rms2(f0[0] / sc0, f0[1] / sc1);
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    h0 = fmin(h0, interval_length);
    /* y1 = y0 + h0*f0 is formed and passed to fun, which ignores it */
    if (stream_f1 >= 0) ns_open(&ns, nz, env_id, stream_f1, block); /* F1 lives in another stream's block */
    if (pos_f1 >= 0) ns_seek(&ns, pos_f1);
    if (law2d) simulate_2d(p, e, act, &ns, f1);
    else simulate(p, mismatched, e, act, &ns, f1);
    const double d2 = rms2((f1[0] - f0[0]) / sc0, (f1[1] - f0[1]) / sc1) / h0;
    double h1;
    if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
    else h1 = pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
    e->h_abs = fmin(fmin(100 * h0, h1), interval_length); /* max_step = inf */
}
static void rk45_construct(const OrcParams* p, int mismatched, OrcEnv* e, const double act[2], OrcNoise* nz,
                           uint32_t env_id, int stream, uint32_t block, int pos_f0, int pos_f1, int stream_f1) {
    rk45_construct_ex(p, mismatched, e, act, nz, env_id, stream, block, pos_f0, pos_f1, stream_f1, 0);
}

/* Simulator.reset_start_pos.  MR_simulator.py:21-34: state, zero action, fresh RK45 at t0 = 0. */
void orc_sim_reset(const OrcParams* p, OrcEnv* e, double x0, double y0, int ctor_mismatched, OrcNoise* nz,
                   uint32_t env_id, int ctor_stream) {
    e->y[0] = x0; e->y[1] = y0;           /* :28-29 */
    e->t = 0.0;                           /* scipy_runge_kutta default t0 = 0, :90 */
    const double zero[2] = {0.0, 0.0};    /* :30 */
    e->n_rhs = 0; e->n_attempts = 0; e->status = 0;
    const int cs = ctor_stream ? ctor_stream : STREAM_RESET_CTOR;
    if (!ctor_mismatched && cs == STREAM_RESET_CTOR)
        rk45_construct(p, 0, e, zero, nz, env_id, STREAM_RESET_POS, 0, 2, 0, STREAM_RESET_CTOR);
    else
        rk45_construct(p, ctor_mismatched, e, zero, nz, env_id, cs, 0, -1, -1, -1);
}

/* Simulator.step.  MR_simulator.py:36-52 */
int orc_sim_step(const OrcParams* p, OrcEnv* e, double f_t, double alpha_t, OrcNoise* nz, uint32_t env_id) {
    const double act[2] = {f_t, alpha_t}; /* :41 */
    const int mis = p->mismatched;
    e->n_rhs = 0; e->n_attempts = 0;
    e->err_margin = INFINITY;
    e->err_norm0 = 0.0;
    NStream ns;

    if (p->integrator != ORC_INT_RK45) {
        /* Build extension (BASELINE configs 2/3), not in the reference: fixed-step
         * Euler / classical RK4 over time_span with `substeps` sub-steps.  Noise enters
         * the derivative at every RHS evaluation, exactly as in `simulate`. */
        const int S = p->substeps > 0 ? p->substeps : 1;
        const double h = p->time_span / S;
        for (int s = 0; s < S; ++s) {
            ns_open(&ns, nz, env_id, STREAM_DYN, (uint32_t)s);
            double k1[2], k2[2], k3[2], k4[2];
            if (p->integrator == ORC_INT_EULER) {
                simulate(p, mis, e, act, &ns, k1);
                e->y[0] = e->y[0] + h * k1[0];
                e->y[1] = e->y[1] + h * k1[1];
            } else {
                simulate(p, mis, e, act, &ns, k1);
                simulate(p, mis, e, act, &ns, k2);
                simulate(p, mis, e, act, &ns, k3);
                simulate(p, mis, e, act, &ns, k4);
                e->y[0] = e->y[0] + (h / 6) * (k1[0] + 2 * k2[0] + 2 * k3[0] + k4[0]);
                e->y[1] = e->y[1] + (h / 6) * (k1[1] + 2 * k2[1] + 2 * k3[1] + k4[1]);
            }
            e->n_attempts++;
        }
        e->t = e->t + p->time_span;
        e->f[0] = e->state_prime[0]; e->f[1] = e->state_prime[1];
        e->h_abs = p->time_span;
        return 0;
    }

    /* while not integrator.status == 'finished': integrator.step()   :42-43
     * OdeSolver.step (base.py) + RungeKutta._step_impl (rk.py), direction = +1, max_step = inf */
    const double t_bound = e->t + p->time_span; /* set when the integrator was built, :49 */
    uint32_t attempt = 0;
    /* the collapsed law re-defines where the Philox-sourced normals enter; a tape is consumed draw for draw, sigma = 0 draws nothing */
    const int collapsed = p->noise_law == ORC_LAW_COLLAPSED && nz->kind == ORC_NOISE_PHILOX && p->sigma != 0.0;
    for (;;) {
        if (e->t == t_bound) break; /* OdeSolver.step corner case */
        const double t = e->t;
        const double y0 = e->y[0], y1 = e->y[1];
        const double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        double h_abs = e->h_abs;
        if (h_abs < min_step) h_abs = min_step;
        int step_accepted = 0, step_rejected = 0;
        double t_new = t, yn0 = y0, yn1 = y1, fn[2] = {0, 0};
        while (!step_accepted) {
            if (h_abs < min_step) { e->status = -1; return -1; } /* TOO_SMALL_STEP -> status 'failed' */
            double h = h_abs;
            t_new = t + h;
            if (t_new - t_bound > 0) t_new = t_bound;
            h = t_new - t;
            h_abs = fabs(h);
            /* rk_step: K[0] = f; K[s] = fun(...), s = 1..5; y_new = y + h*dot(K[:-1].T, B); K[6] = f_new */
            double K[7][2];
            double s0 = 0, s1 = 0, e0 = 0, e1 = 0;
            K[0][0] = e->f[0]; K[0][1] = e->f[1];
            ns_open(&ns, nz, env_id, STREAM_DYN, attempt);
            if (collapsed) {
                /* the stage sums of rk_step drawn directly (see COL_* above): with K_i = V + N_i, sum B = 1, sum E = 0,
                 * B1 = E1 = 0:  dot(K[:-1].T, B) = B0 K0 + (1 - B0) V + S_B,  dot(K.T, E) = E0 (K0 - V) + S_E + E6 N_6 */
                double V[2], G[2], z1[3], z2[3], z6[3];
                rhs_mean(p, mis, act, V, G);
              if (mis) {
                /* mismatched model: pairs through M, f_new's noise folded into the error sum (COL_* above) */
                double u1[2], u2[2], w[2], v[2], SB[2], SE[2], N6[2];
                ns_seek(&ns, CMIS_POS_U1); u1[0] = ns_normal(&ns, 1.0); u1[1] = ns_normal(&ns, 1.0);
                ns_seek(&ns, CMIS_POS_U2); u2[0] = ns_normal(&ns, 1.0); u2[1] = ns_normal(&ns, 1.0);
                ns_seek(&ns, CMIS_POS_W);  w[0] = ns_normal(&ns, 1.0);  w[1] = ns_normal(&ns, 1.0);
                v[0] = COL_CB * u1[0]; v[1] = COL_CB * u1[1];
                noise_2d(p, G, v, SB);
                v[0] = COL_CE1 * u1[0] + COL_CE2P * u2[0]; v[1] = COL_CE1 * u1[1] + COL_CE2P * u2[1];
                noise_2d(p, G, v, SE);                                              /* S_E + E6 N_6 */
                v[0] = (RK_E[6] / COL_CE2P) * u2[0] + (COL_CE2 / COL_CE2P) * w[0];
                v[1] = (RK_E[6] / COL_CE2P) * u2[1] + (COL_CE2 / COL_CE2P) * w[1];
                noise_2d(p, G, v, N6);                                              /* N_6 given S_E' (only a following sub-step sees it) */
                for (int c = 0; c < 2; ++c) {
                    fn[c] = V[c] + N6[c];
                    const double sB = RK_B[0] * K[0][c] + (1.0 - RK_B[0]) * V[c] + SB[c];
                    const double sE = RK_E[0] * (K[0][c] - V[c]) + SE[c];
                    if (c == 0) { s0 = sB; e0 = sE; } else { s1 = sB; e1 = sE; }
                }
              } else {
                ns_seek(&ns, CNOM_POS_Z1);
                z1[0] = 0.0; z1[1] = ns_normal(&ns, 1.0); z1[2] = ns_normal(&ns, 1.0);
                ns_seek(&ns, CNOM_POS_Z2);
                z2[0] = 0.0; z2[1] = ns_normal(&ns, 1.0); z2[2] = ns_normal(&ns, 1.0);
                ns_seek(&ns, CNOM_POS_K6); z6[0] = 0.0; z6[1] = ns_normal(&ns, 1.0); z6[2] = ns_normal(&ns, 1.0);
                for (int c = 0; c < 2; ++c) {
                    const double sb = G[c] * (COL_CB * z1[0]) + p->sigma * (COL_CB * z1[1 + c]);
                    const double se = G[c] * (COL_CE1 * z1[0] + COL_CE2 * z2[0]) +
                                      p->sigma * (COL_CE1 * z1[1 + c] + COL_CE2 * z2[1 + c]);
                    const double n6 = G[c] * z6[0] + p->sigma * z6[1 + c];
                    fn[c] = V[c] + n6;                                           /* K[6] = f_new */
                    const double sB = RK_B[0] * K[0][c] + (1.0 - RK_B[0]) * V[c] + sb;
                    const double sE = RK_E[0] * (K[0][c] - V[c]) + se + RK_E[6] * n6;
                    if (c == 0) { s0 = sB; e0 = sE; } else { s1 = sB; e1 = sE; }
                }
              }
                e->state_prime[0] = fn[0]; e->state_prime[1] = fn[1];            /* last RHS evaluation of rk_step */
                e->n_rhs += 6;
            } else {
                for (int s = 1; s < 6; ++s) {
                    simulate(p, mis, e, act, &ns, K[s]);
                    if (mis && s == 1) ns_seek(&ns, MIS_POS_K2); /* K1 has weight 0 in B and E */
                }
                for (int i = 0; i < 6; ++i) { s0 += K[i][0] * RK_B[i]; s1 += K[i][1] * RK_B[i]; }
            }
            yn0 = y0 + h * s0;
            yn1 = y1 + h * s1;
            if (!collapsed) {
                ns_seek(&ns, mis ? MIS_POS_K6 : NOM_POS_K6);
                simulate(p, mis, e, act, &ns, fn);
                K[6][0] = fn[0]; K[6][1] = fn[1];
                for (int i = 0; i < 7; ++i) { e0 += K[i][0] * RK_E[i]; e1 += K[i][1] * RK_E[i]; }
            }
            attempt++; e->n_attempts++;
            /* scale = atol + max(|y|,|y_new|)*rtol; error_norm = norm(dot(K.T,E)*h/scale) */
            const double sc0 = p->atol + fmax(fabs(y0), fabs(yn0)) * p->rtol;
            const double sc1 = p->atol + fmax(fabs(y1), fabs(yn1)) * p->rtol;
            const double error_norm = rms2(e0 * h / sc0, e1 * h / sc1);
            if (fabs(error_norm - 1.0) < e->err_margin) e->err_margin = fabs(error_norm - 1.0);
            if (e->n_attempts == 1) e->err_norm0 = error_norm;
            if (error_norm < 1) {
                double factor = (error_norm == 0) ? RK_MAX_FACTOR
                                                  : fmin(RK_MAX_FACTOR, RK_SAFETY * pow(error_norm, RK_ERR_EXP));
                if (step_rejected) factor = fmin(1.0, factor);
                h_abs *= factor;
                step_accepted = 1;
            } else {
                h_abs *= fmax(RK_MIN_FACTOR, RK_SAFETY * pow(error_norm, RK_ERR_EXP));
                step_rejected = 1;
            }
            if (attempt > (1u << 20)) { e->status = -2; return -2; } /* oracle guard */
        }
        e->t = t_new; e->y[0] = yn0; e->y[1] = yn1; e->h_abs = h_abs;
        e->f[0] = fn[0]; e->f[1] = fn[1];
        if (e->t - t_bound >= 0) break; /* status = 'finished' */
    }
    /* last_state = integrator.y (:45); new RK45 from (t, y) to t + time_span (:46-50) */
    if (collapsed)
        rk45_construct_ex(p, mis, e, act, nz, env_id, STREAM_DYN, attempt ? attempt - 1 : 0,
                          mis ? CMIS_POS_F0 : CNOM_POS_F0, mis ? CMIS_POS_F1 : CNOM_POS_F1, -1, mis);
    else
    rk45_construct(p, mis, e, act, nz, env_id, STREAM_DYN, attempt ? attempt - 1 : 0,
                   mis ? MIS_POS_F0 : NOM_POS_F0, mis ? MIS_POS_F1 : NOM_POS_F1, -1);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* MR_Env                                                                     */
/* ------------------------------------------------------------------------ */
/* convert_state.  MR_env.py:100-116 */
void orc_convert_state(const double state[2], const double goal[2], double obs[5]) {
    const double dx = goal[0] - state[0], dy = goal[1] - state[1];
    obs[0] = state[0]; obs[1] = state[1]; obs[2] = goal[0]; obs[3] = goal[1];
    obs[4] = sqrt(dx * dx + dy * dy); /* np.linalg.norm(goal - cur) */
}

/* spaces.Box.contains, numeric meaning: low <= obs <= high (gym is absent here) */
static int obs_in_bounds(const OrcParams* p, const double obs[5]) {
    for (int i = 0; i < 5; ++i)
        if (!(obs[i] >= p->obs_low[i] && obs[i] <= p->obs_high[i])) return 0;
    return 1;
}

/* calculate_reward.  MR_env.py:118-134 (defined, call site commented out at :89) */
double orc_calculate_reward(const OrcParams* p, const double obs[5], int32_t counter) {
    const double d = obs[4];
    if (d < p->min_dist2goal) return 100.0;
    else if (!obs_in_bounds(p, obs) || counter > p->max_timesteps) return -100.0;
    else return -0.1;
}

/* end.  MR_env.py:136-152 */
int orc_end(const OrcParams* p, const double obs[5], int32_t counter) {
    const double d = obs[4];
    if (!obs_in_bounds(p, obs) || counter > p->max_timesteps) return 1;
    else if (d < p->min_dist2goal) return 1;
    else return 0;
}

/* Goal for an env at a given episode step.  Reference: fixed init_goal = (0,0)
 * (MR_env.py:57,157-162) == table K = T = 1 {(0,0)}.  Extension: reference-trajectory
 * table [K][T][2] (fp32), trajectory = env_id mod K, row = min(counter, T-1). */
void orc_goal_at(const OrcParams* p, const float* goal_table, uint32_t env_id, int32_t counter, double goal[2]) {
    if (!goal_table) { goal[0] = goal[1] = 0.0; return; }
    int K = p->goal_K > 0 ? p->goal_K : 1, T = p->goal_T > 0 ? p->goal_T : 1;
    int k = (int)(env_id % (uint32_t)K);
    int r = counter < 0 ? 0 : (counter >= T ? T - 1 : counter);
    goal[0] = (double)goal_table[((int64_t)k * T + r) * 2 + 0];
    goal[1] = (double)goal_table[((int64_t)k * T + r) * 2 + 1];
}

/* init_space.sample(): uniform float32 in [low, high)  (MR_env.py:40-42,173) */
void orc_sample_init(const OrcParams* p, uint64_t seed, uint32_t env_id, uint64_t step_idx, double xy[2]) {
    double u[2];
    orc_uniform2(seed, env_id, step_idx, C0(STREAM_RESET_POS, 0, 0), u);
    xy[0] = (double)(float)(p->init_low[0] + (p->init_high[0] - p->init_low[0]) * u[0]);
    xy[1] = (double)(float)(p->init_low[1] + (p->init_high[1] - p->init_low[1]) * u[1]);
}

/* Exploration policy (build extension).  With the RK45 integrator the two uniform words are words 0,1
 * of DYN block 0 call 0 -- the words whose normals would feed stage K1 of the first rk_step attempt,
 * which never reaches a result (B[1] = E[1] = 0) -- so the fused GPU kernels need one Philox call
 * less per step.  Fixed-step modes consume those words for noise and give the policy its own call. */
void orc_random_action(int integrator, uint64_t seed, uint32_t env_id, uint64_t step_idx, const double lo[2],
                       const double hi[2], float act[2]) {
    /* fp32 on purpose (actions are float32 at the ABI): u = fma(float(w), 2^-32, 2^-33), a = fma(span, u, lo) */
    uint32_t r[4];
    philox_call(seed, env_id, step_idx, integrator == ORC_INT_RK45 ? C0(STREAM_DYN, 0, 0) : C0(STREAM_POLICY, 0, 0), r);
    for (int j = 0; j < 2; ++j) {
        const float u = fmaf((float)r[j], 2.3283064365386963e-10f, 1.1641532182693481e-10f);
        act[j] = fmaf((float)(hi[j] - lo[j]), u, (float)lo[j]);
    }
}

/* ------------------------------------------------------------------------ */
/* DDPG actor + OU noise as the policy (RL/MR_ddpg.py:120-137,145-148,69-73,277)      */
/* fp32, summation order of mr_rl_amd/csrc/mrsim_actor.h                     */
/* ------------------------------------------------------------------------ */
static int act_kperm(int q, int h) { return 32 * (q / 16) + 8 * ((q % 16) / 4) + 4 * h + (q % 4); }

/* exp(x), Cephes expf with rint() reduction, explicit fma, exact ldexp */
static float spec_expf(float x) {
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    const float z = r * r;
    const float e = fmaf(p, z, r) + 1.0f;
    return ldexpf(e, (int)n);
}

/* tflearn activation 'tanh' (RL/MR_ddpg.py:133) in specified fp32 arithmetic: Cephes tanhf */
float orc_spec_tanhf(float x) {
    const float ax = fabsf(x);
    float r;
    if (ax >= 9.0f) {
        r = 1.0f;
    } else if (ax >= 0.625f) {
        const float e = spec_expf(ax + ax);
        r = 1.0f - 2.0f / (e + 1.0f);
    } else {
        const float z = x * x;
        float p = -5.70498872745E-3f;
        p = fmaf(p, z, 2.06390887954E-2f);
        p = fmaf(p, z, -5.37397155531E-2f);
        p = fmaf(p, z, 1.33314422036E-1f);
        p = fmaf(p, z, -3.33332819422E-1f);
        return fmaf(p * z, x, x);
    }
    return x < 0.0f ? -r : r;
}

/* bf16 helpers of the bf16 x 3 arithmetic: round-to-nearest-even as v_cvt_pk_bf16_f32 does, x = x1 + x2 + x3 */
static float bf16_round(float x) {
    uint32_t u = f2u(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return x;
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    return u2f(u);
}
static void bf16_split3(float x, float t[3]) {
    t[0] = bf16_round(x);
    const float r1 = x - t[0];
    t[1] = bf16_round(r1);
    const float r2 = r1 - t[1];
    t[2] = bf16_round(r2);
}

/* fills OrcActor.w2_split from w2 (call after setting the weights, before math = 1 evaluations) */
void orc_actor_prepare(OrcActor* a) {
    for (int i = 0; i < 64 * 64; ++i) bf16_split3(a->w2[i], a->w2_split + 3 * i);
    for (int i = 0; i < 64 * 5; ++i) bf16_split3(a->w1[i], a->w1_split + 3 * i);
}

/* ActorNetwork.predict (RL/MR_ddpg.py:145-148) -> scaled_out (:136-137) */
void orc_actor_forward(const OrcActor* a, const float obs[5], float act[2]) {
    float h1[64], h2[64];
    if (a->math == 2) {
        /* plain bf16: layer 1 is ONE v_mfma_f32_32x32x16_bf16 per tile -- the five inputs as three bf16 terms each (their sum is
         * the input, exactly) against the bf16-rounded weight, 15 exact products and the bias summed and rounded once */
        double x[5];
        for (int k = 0; k < 5; ++k) {
            float t[3];
            bf16_split3(obs[k], t);
            x[k] = ((double)t[0] + (double)t[1]) + (double)t[2];
        }
        for (int f = 0; f < 64; ++f) {
            double sum = 0.0;
            for (int k = 0; k < 5; ++k) sum += (double)bf16_round(a->w1[f * 5 + k]) * x[k];
            const float acc = (float)((double)a->b1[f] + sum);
            h1[f] = acc > 0.0f ? acc : 0.0f;
        }
    } else if (a->math == 1) {
        /* bf16 x 3: weights and inputs as three bf16 terms, the six products above 2^-24 in TWO MFMAs per tile, the 2^-16
         * terms (w1 x3 + w2 x2 + w3 x1) first; each instruction's products summed exactly and rounded once */
        float xs[5][3];
        for (int k = 0; k < 5; ++k) bf16_split3(obs[k], xs[k]);
        for (int f = 0; f < 64; ++f) {
            const float* ws = a->w1_split + (size_t)f * 5 * 3;
            double small = 0.0, big = 0.0;
            for (int k = 0; k < 5; ++k) {
                small += (double)ws[k * 3 + 0] * xs[k][2] + (double)ws[k * 3 + 1] * xs[k][1] + (double)ws[k * 3 + 2] * xs[k][0];
                big += (double)ws[k * 3 + 0] * xs[k][0] + (double)ws[k * 3 + 0] * xs[k][1] + (double)ws[k * 3 + 1] * xs[k][0];
            }
            float acc = (float)((double)a->b1[f] + small);
            acc = (float)((double)acc + big);
            h1[f] = acc > 0.0f ? acc : 0.0f;
        }
    } else
    for (int f = 0; f < 64; ++f) {                              /* fully_connected 64 + batch norm (folded), relu  :122-124 */
        float acc = a->b1[f];
        for (int k = 0; k < 5; ++k) acc = fmaf(a->w1[f * 5 + k], obs[k], acc);
        h1[f] = acc > 0.0f ? acc : 0.0f;
    }
    if (a->math == 0) {
        for (int f = 0; f < 64; ++f) {                          /* fully_connected 64 + batch norm (folded), relu  :125-127 */
            float acc = a->b2[f];
            for (int q = 0; q < 32; ++q)
                for (int h = 0; h < 2; ++h) acc = fmaf(a->w2[f * 64 + act_kperm(q, h)], h1[act_kperm(q, h)], acc);
            h2[f] = acc > 0.0f ? acc : 0.0f;
        }
    } else {
        /* bf16 x 3: k-step s of v_mfma_f32_32x32x16_bf16 sums the 16 features kperm(8 s + jj, h), h = 0..1, jj = 0..7; six
         * MFMAs per k-step in the kernel's order (a3 b1, a2 b2, a1 b3, a2 b1, a1 b2, a1 b1) */
        static const int term6[6][2] = {{2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};
        static const int term1[1][2] = {{0, 0}};                /* math = 2: plain bf16 operands, one MFMA per k-step */
        const int (*term)[2] = a->math == 1 ? term6 : term1;
        const int nterm = a->math == 1 ? 6 : 1;
        float hs[64][3];
        for (int k = 0; k < 64; ++k) bf16_split3(h1[k], hs[k]);
        for (int f = 0; f < 64; ++f) {
            float acc = a->b2[f];
            const float* ws = a->w2_split + (size_t)f * 64 * 3;   /* orc_actor_prepare: the three terms of W2[f][k] */
            for (int s = 0; s < 4; ++s)
                for (int m = 0; m < nterm; ++m) {
                    double sum = 0.0;
                    for (int h = 0; h < 2; ++h)
                        for (int jj = 0; jj < 8; ++jj) {
                            const int k = act_kperm(8 * s + jj, h);
                            sum += (double)ws[k * 3 + term[m][0]] * (double)hs[k][term[m][1]];
                        }
                    acc = (float)((double)acc + sum);
                }
            h2[f] = acc > 0.0f ? acc : 0.0f;
        }
    }
    if (a->math == 2) {
        /* plain bf16: the output layer runs on the bf16 matrix cores as well -- W3 and relu(h2) rounded to bf16 once, k-step s of
         * v_mfma_f32_32x32x16_bf16 sums the 16 features kperm(8 s + jj, h) exactly and rounds once into the f32 accumulator */
        for (int o = 0; o < 2; ++o) {
            float acc = 0.0f;
            for (int s = 0; s < 4; ++s) {
                double sum = 0.0;
                for (int h = 0; h < 2; ++h)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int k = act_kperm(8 * s + jj, h);
                        sum += (double)bf16_round(a->w3[o * 64 + k]) * (double)bf16_round(h2[k]);
                    }
                acc = (float)((double)acc + sum);
            }
            const float pre = (acc + 0.0f) + a->b3[o];
            act[o] = orc_spec_tanhf(pre) * a->bound[o];
        }
        return;
    }
    for (int o = 0; o < 2; ++o) {                               /* fully_connected 2, tanh, * action_bound  :130-137 */
        float p[2] = {0.0f, 0.0f};
        for (int h = 0; h < 2; ++h)
            for (int q = 0; q < 32; ++q) p[h] = fmaf(a->w3[o * 64 + act_kperm(q, h)], h2[act_kperm(q, h)], p[h]);
        const float pre = (p[0] + p[1]) + a->b3[o];
        act[o] = orc_spec_tanhf(pre) * a->bound[o];
    }
}

/* action = actor.predict(state) + actor_noise()  (RL/MR_ddpg.py:277); OUNoise.__call__ :69-73 with mu = 0 */
void orc_actor_policy(const OrcActor* a, int integrator, const float obs[5], int32_t counter, float ou[2], uint64_t seed,
                      uint32_t env_id, uint64_t step_idx, float act[2]) {
    orc_actor_forward(a, obs, act);
    if (a->ou_enabled && ou) {
        if (a->ou_reset_on_done && counter == 0) ou[0] = ou[1] = 0.0f;
        uint32_t r[4];
        float z[2];
        philox_call(seed, env_id, step_idx, integrator == ORC_INT_RK45 ? C0(STREAM_DYN, 0, 0) : C0(STREAM_POLICY, 0, 0), r);
        orc_box_muller(r[0], r[1], &z[0], &z[1]);
        for (int j = 0; j < 2; ++j) {
            ou[j] = fmaf(a->ou_sigma_sqrt_dt, z[j], fmaf(-a->ou_theta_dt, ou[j], ou[j]));
            act[j] = act[j] + ou[j];
        }
    }
}

int orc_vec_actor_policy(const OrcActor* a, int integrator, int64_t n, uint32_t env_id0, const float* obs,
                         const int32_t* counter, float* ou, uint64_t seed, uint64_t step_idx, float* actions, int threads) {
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < n; ++i)
        orc_actor_policy(a, integrator, obs + 5 * i, counter ? counter[i] : 1, ou ? ou + 2 * i : 0, seed,
                         env_id0 + (uint32_t)i, step_idx, actions + 2 * i);
    return 0;
}

int orc_sizeof_actor(void) { return (int)sizeof(OrcActor); }

/* reset.  MR_env.py:164-201 (prints and the unused second sample() omitted) */
void orc_env_reset(const OrcParams* p, OrcEnv* e, const float* goal_table, double x0, double y0,
                   int ctor_mismatched, OrcNoise* nz, uint32_t env_id, double obs[5]) {
    orc_sim_reset(p, e, x0, y0, ctor_mismatched, nz, env_id, STREAM_RESET_CTOR); /* :179-181 */
    e->counter = 0;                                                           /* :186 */
    e->ep_ret = 0.0;
    double goal[2];
    orc_goal_at(p, goal_table, env_id, 0, goal);
    if (obs) orc_convert_state(e->y, goal, obs);                              /* :201 */
}

/* step.  MR_env.py:70-98 */
int orc_env_step(const OrcParams* p, OrcEnv* e, const float* goal_table, double f_t, double alpha_t,
                 OrcNoise* nz, uint32_t env_id, double obs[5], double* rew, uint8_t* done,
                 double final_obs[5], double* final_ret, int32_t* final_len) {
    e->counter += 1;                                                          /* :80 */
    int rc = orc_sim_step(p, e, f_t, alpha_t, nz, env_id);                    /* :83 */
    if (rc) return rc;
    double goal[2];
    orc_goal_at(p, goal_table, env_id, e->counter, goal);
    orc_convert_state(e->y, goal, obs);                                       /* :87 */
    int d = orc_end(p, obs, e->counter);                                      /* :88 */
    double r = (p->reward_mode == ORC_REW_GOAL) ? orc_calculate_reward(p, obs, e->counter) : 10.0; /* :89 */
    e->ep_ret += r;
    *rew = r; *done = (uint8_t)d;
    if (d && p->auto_reset) {
        /* build extension: same-step auto-reset.  Terminal obs/return/length go to the
         * final_* outputs, the env restarts as MR_Env.reset(init=None) would on the re-used
         * env object (auto_reset_fresh_env = 0) or on a fresh one (= 1; MR_env.py:181-183) and the RETURNED obs is
         * the reset observation. */
        if (final_obs) memcpy(final_obs, obs, 5 * sizeof(double));
        if (final_ret) *final_ret = e->ep_ret;
        if (final_len) *final_len = e->counter;
        /* RNG position of an auto-reset's draws (start position, constructor noise): the step index at which the episode
         * that just ended took its FIRST step, step_idx - (length - 1) in 64-bit modular arithmetic.  Every episode has its
         * own first step, so every reset has its own counter block -- and the block is known from the moment the episode
         * starts, which lets the GPU rollout prepare the resets of a wave's envs together (mrsim_kernels.hip: reset cache). */
        const uint64_t step_now = nz->step_idx;
        nz->step_idx = step_now - (uint64_t)(uint32_t)(e->counter - 1);
        double xy[2];
        orc_sample_init(p, nz->seed, env_id, nz->step_idx, xy);
        /* MR_env.py:181-183: reset_start_pos runs BEFORE is_mismatched is assigned, so the env object an episode loop
         * re-uses builds its RK45 under the previous episode's law; a fresh object under the nominal one */
        orc_env_reset(p, e, goal_table, xy[0], xy[1], (p->mismatched && !p->auto_reset_fresh_env) ? 1 : 0, nz, env_id, obs);
        nz->step_idx = step_now;
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* batched drivers                                                           */
/* ------------------------------------------------------------------------ */
int orc_vec_reset(const OrcParams* p, int64_t n, uint32_t env_id0, OrcEnv* envs, const float* goal_table,
                  const double* init_xy, uint64_t seed, uint64_t step_idx, double* obs, int threads) {
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < n; ++i) {
        OrcNoise nz = {p->sigma != 0.0 ? ORC_NOISE_PHILOX : ORC_NOISE_NONE, seed, step_idx, 0, 0, 0};
        uint32_t id = env_id0 + (uint32_t)i;
        double xy[2];
        if (init_xy) { xy[0] = init_xy[2 * i]; xy[1] = init_xy[2 * i + 1]; }
        else orc_sample_init(p, seed, id, step_idx, xy);
        orc_env_reset(p, &envs[i], goal_table, xy[0], xy[1], 0, &nz, id, obs ? obs + 5 * i : 0);
    }
    return 0;
}

int orc_vec_step(const OrcParams* p, int64_t n, uint32_t env_id0, OrcEnv* envs, const float* goal_table,
                 const float* actions, uint64_t seed, uint64_t step_idx, double* obs, double* rew,
                 uint8_t* done, double* final_obs, double* final_ret, int32_t* final_len, int threads) {
    int bad = 0;
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1) reduction(| : bad)
    for (int64_t i = 0; i < n; ++i) {
        OrcNoise nz = {p->sigma != 0.0 ? ORC_NOISE_PHILOX : ORC_NOISE_NONE, seed, step_idx, 0, 0, 0};
        uint32_t id = env_id0 + (uint32_t)i;
        int rc = orc_env_step(p, &envs[i], goal_table, (double)actions[2 * i], (double)actions[2 * i + 1], &nz,
                              id, obs + 5 * i, rew + i, done + i, final_obs ? final_obs + 5 * i : 0,
                              final_ret ? final_ret + i : 0, final_len ? final_len + i : 0);
        if (rc) bad |= 1;
    }
    return bad ? -1 : 0;
}

int orc_vec_random_policy(int integrator, int64_t n, uint32_t env_id0, uint64_t seed, uint64_t step_idx,
                          const double lo[2], const double hi[2], float* actions, int threads) {
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < n; ++i)
        orc_random_action(integrator, seed, env_id0 + (uint32_t)i, step_idx, lo, hi, actions + 2 * i);
    return 0;
}
