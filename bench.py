#!/usr/bin/env python3
"""bench.py -- env-steps/s of the MR_env.step() hot path on MI355X (BASELINE.json metric).

Workload (BASELINE config 4): N = 262 144 envs per GPU, integrator = reference
(SciPy-RK45 semantics), sigma = 1 (MR_Env.reset default), uniform random policy in the DDPG actor
range drawn on device, reward + done on device, same-step auto-reset (every episode is 51 steps),
seed 7.  One "step" = one MR_Env.step() of all N envs.

--mode rollout (default): the fused kernel advances all envs --rollout-len (= 51, one episode) steps
    per launch with the env state in registers and writes every step's transition (obs[5], action[2],
    reward, done = 33 B per env-step) to [T, N, ...] buffers in HBM -- the DDPG warm-up/rollout
    workload.  K steps = K / 51 launches.
--mode step: one launch per env.step() (the drop-in gym loop): [policy kernel -> actions in HBM] +
    [step kernel], captured as a hipGraph of --graph-len steps.

--gpus N > 1: this process starts N rank processes itself (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in their environment, rendezvous on 127.0.0.1) BEFORE anything touches the GPU, relays rank 0's
JSON line and exits non-zero if any rank does.  Under a launcher that already set WORLD_SIZE (the driver's
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) it is a rank.  Every rank owns
a contiguous shard of N x 262144 envs (weak scaling), there is no data-path collective, and the returns of every
episode are all-gathered over RCCL inside the timed region, --gather-interval (8) episodes per collective.

Prints ONE JSON line on rank 0.  Besides the contract's fields:
  roofline      the dominant kernel (the fused rollout) against HBM: `achieved` = the bytes one launch really moves (`traffic`: the
                committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS build of libmrsim.so -- refused when the library
                hash differs -- else the minimal-traffic model, labelled) / the kernel's average launch duration, from HIP events
                attached to the dispatches of a one-stream region of this very run; `peak` 8000 GB/s.  SURVEY 8(d)'s algorithmic
                97 B/env-step (state round-trips HBM every step) is what the one-launch-per-step form moves, not the fused
                kernel: kept as the labelled `algorithmic_equiv`, never as `frac`.  `valu`: the kernel's vector-issue floor from
                the committed SQ counters.
  config.noise_law / other_noise_law   the headline runs --noise-law collapsed (same distribution as the reference's per-evaluation
                noise, element-wise parity against the oracle's restatement of the same law); the per-stage law
                (the library's default) is measured beside it in the same run, with its own roofline object.
  sustained     the same workload over its own >= 10 200-step region, whatever --steps was (a 20-step timed region is
                ONE launch and is launch-latency-bound); kernel durations come from HIP events attached to that
                region's dispatches.
  actor_in_loop the collection loop with the reference's DDPG actor + OU noise evaluated inside the rollout kernel (MFMA roofline).
  learner       the DDPG update: updates/s eager / one hipGraph / libmrsim's fused kernel, and env-steps/s of the whole
                collect-and-learn loop at a stated update : transition ratio.
  cpu_baseline  the oracle (C restatement) on this box's host cores.
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 97   # fp64 positions -> reads 40 + writes 57 per env-step
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0    # same table: 6.29 TB/s measured (float4 copy, 79 %)
SIMDS = 1024                   # 256 CUs x 4 SIMDs
PREROLL_EPISODES = 900         # untimed episodes issued right in front of every secondary timed region: after ANY host-side
                               # pause of a millisecond or more (allocating buffers, creating or reading a thousand events)
                               # the GPU has dropped its clock and needs ~40 ms of load to bring it back (rocprofv3 kernel
                               # trace of this script, profiles/r02/rollout_kernel_by_region.json: 150 us per launch right
                               # after a pause, 126 us from 300 launches on).  900, not 300: behind the driver's 20-step region
                               # (drained queue, 2 ms pause, short burst) 300 episodes left the sustained legs 10 % low --
                               # 101-104 G against 113-115 G behind 900 or 2000 and behind the default 1000-episode region
                               # (profiles/r03/NOTES.md)

_T0 = time.perf_counter()


def trace(msg):
    """phase markers on stderr (MRSIM_BENCH_TRACE=1): where a profiled run is when something goes wrong"""
    if os.environ.get("MRSIM_BENCH_TRACE"):
        print("[bench +%.3fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def under_pmc():
    """rocprofv3 --pmc exports ROCPROF_COUNTER_COLLECTION to the profiled process.  hipGraph capture / replay under
    counter collection crashes the profiler (ROCm 7.2: SIGSEGV in its collection thread -- the root cause of round 1's
    stale pmc_traffic.json, found with tools/pmc_bisect.sh), so graph legs are skipped there and the JSON says so."""
    return bool(os.environ.get("ROCPROF_COUNTER_COLLECTION"))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=51000)   # 1000 episodes
    ap.add_argument("--warmup", type=int, default=10200)   # 200 episodes = 35 ms: the GPU clocks have settled (tools/clock_ramp_probe.py)
    ap.add_argument("--envs-per-gpu", type=int, default=262144)
    ap.add_argument("--mode", choices=["rollout", "step"], default="rollout")
    ap.add_argument("--rollout-len", type=int, default=51)
    ap.add_argument("--carry", choices=["f32", "f64"], default="f64",
                    help="rollout mode, carried RK45 state (integrator.f, h_abs) inside a launch: f64 registers (default; "
                         "what the reference carries) or rounded to its fp32 HBM format every step (bit-identical to "
                         "the step path)")
    ap.add_argument("--streams", type=int, default=2,
                    help="rollout mode: the envs of a GPU run as this many sub-shard launches on as many HIP streams "
                         "(mr_rl_amd.collector; 1 = one launch per episode).  Kernel durations for `roofline` always come "
                         "from a one-stream region")
    ap.add_argument("--gather-interval", type=int, default=8,
                    help="rollout mode, N > 1 ranks: episodes per RCCL all-gather of returns -- the returns of EVERY episode are "
                         "gathered, E episodes ([E, n_local] per rank) per collective (the reference logs per 100 episodes, "
                         "RL/MR_ddpg.py:317-320); 1 = one collective per episode, which is bound by the ~100 us of Python / RCCL "
                         "enqueue per episode, not by the GPU")
    ap.add_argument("--policy", choices=["kernel", "overlap", "episode", "fused"], default="kernel",
                    help="step mode only: policy kernel -> HBM -> step kernel; the same with step t+1's policy kernel on a "
                         "second captured stream beside step t; one policy launch per episode (51 rows); or drawn inside the step kernel")
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph", help="step mode only")
    ap.add_argument("--graph-len", type=int, default=51)
    ap.add_argument("--obs-layout", choices=["aos", "soa"], default="aos")
    ap.add_argument("--noise-math", choices=["fast", "spec"], default="fast")
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--noise-law", choices=["collapsed", "per_stage"], default=None,
                    help="where the normals of an RK45 step enter (include/mrsim.h, MRSIM_LAW_*).  Default: the LIBRARY's default "
                         "(mrsim_default_params / MRConfig: collapsed = the weighted stage sums drawn directly from their joint "
                         "Gaussian; the law of MR_simulator.py:73-83's per-evaluation noise, pinned against the reference's own "
                         "samples -- tests/increments.py -- and element-wise against the oracle's restatement).  per_stage = one draw "
                         "per RHS evaluation in the reference's order.  The line reports the other law beside the headline "
                         "(`other_noise_law`)")
    ap.add_argument("--no-other-law", action="store_true", help="skip the leg that measures the other noise law")
    ap.add_argument("--mismatched", action="store_true", help="non-default: the reference's is_mismatched=True law")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--kernel-samples", type=int, default=0, help="0 = auto")
    ap.add_argument("--workload", choices=["ddpg", "mixed"], default="ddpg",
                    help="ddpg = BASELINE config 4 (default); mixed = config 5's trajectory set: goal = reference-"
                         "trajectory table[env_id mod 3] (straight line / figure-eight / random), tracking reward")
    ap.add_argument("--settle-episodes", type=int, default=400,
                    help="untimed episodes before the W warm-up steps so that the GPU clocks have settled whatever W "
                         "is (a fixed count, not a time: every rank must issue the same collectives)")
    ap.add_argument("--queue-depth", type=int, default=32,
                    help="untimed settle phase: episodes enqueued between two device-wide synchronizes (bounded launch queue; any --steps)")
    ap.add_argument("--sustained-steps", type=int, default=10200, help="length of the `sustained` leg (0 = skip)")
    ap.add_argument("--no-step-path", action="store_true", help="skip the extra one-launch-per-step measurement")
    ap.add_argument("--no-mixed-set", action="store_true", help="skip the extra mixed-trajectory-set measurement")
    ap.add_argument("--no-power", action="store_true", help="skip the 1.5 s package-power leg (hwmon sysfs)")
    ap.add_argument("--no-actor-leg", action="store_true", help="skip the actor-in-the-loop collection measurement")
    ap.add_argument("--no-learner-leg", action="store_true", help="skip the DDPG learner measurement (updates/s, end-to-end training rate)")
    ap.add_argument("--no-facade-leg", action="store_true", help="skip the single-env MR_Env.step() drop-in measurement")
    ap.add_argument("--no-consumers-leg", action="store_true", help="skip the run_sim + velocity post-processing / recorder export measurement")
    ap.add_argument("--no-streaming-point", action="store_true", help="skip the N = 2 097 152 single-GPU streaming point (SURVEY H4)")
    ap.add_argument("--no-partition-row", action="store_true",
                    help="learner leg without its compute-unit-partition row (round 4 needed this under rocprofv3 --kernel-trace: "
                         "CU-masked streams left alive at interpreter exit crash the profiler's exit path; they are destroyed in "
                         "order now -- profiles/r05/partition_kt.txt -- and the flag is no longer used by tools/profile_round.sh)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real runs; gloo only to rehearse")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--master-port", type=int, default=0, help="--gpus N launcher: rendezvous port (0 = pick a free one)")
    ap.add_argument("--rendezvous-timeout", type=float, default=180.0,
                    help="seconds init_process_group may wait for the other ranks before this rank gives up (exit code 3)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------------
# --gpus N launcher (no GPU / torch.cuda call on this path: a process that has touched the GPU must never be re-executed)
# ----------------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_commands(n_ranks, argv, port, python=None, script=None, base_env=None):
    """argv / environment of the N rank processes `bench.py --gpus N` starts: [(cmd, env), ...] -- rank r gets
    RANK = LOCAL_RANK = r, WORLD_SIZE = LOCAL_WORLD_SIZE = N and a rendezvous on 127.0.0.1:port."""
    cmd = [python or sys.executable, script or os.path.abspath(__file__)] + list(argv)
    out = []
    for r in range(n_ranks):
        env = dict(os.environ if base_env is None else base_env)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                    "MRSIM_BENCH_LAUNCHER": "bench.py"})
        # as torch.distributed.run does: N ranks x all host cores of OpenMP / intra-op threads oversubscribe the box
        # (measured with two gloo ranks: 230 ms per host-staged collective without this, ~2 ms with it)
        env.setdefault("OMP_NUM_THREADS", "1")
        out.append((cmd, env))
    return out


def launch_ranks(args, argv):
    """Parent of `bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment): start the N ranks as fresh child
    processes, relay rank 0's stdout (the JSON line), wait for all, return the first non-zero exit code."""
    import tempfile
    ensure_built()
    port = args.master_port or _free_port()
    procs = []
    out0 = tempfile.TemporaryFile()
    for r, (cmd, env) in enumerate(rank_commands(args.gpus, argv, port)):
        procs.append(subprocess.Popen(cmd, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = set(range(len(procs)))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print("[bench launcher] rank %d exited with code %d; stopping the other ranks" % (r, code),
                          file=sys.stderr, flush=True)
                    for q in pending:
                        procs[q].terminate()  # exactly the children started above
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    out0.seek(0)
    # the contract is ONE JSON line on stdout: libraries of the child may have written there too (gloo announces its
    # connections on stdout), that goes to stderr
    for line in out0.read().decode(errors="replace").splitlines():
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return rc


def ensure_built():
    lib_so = os.path.join(ROOT, "mr_rl_amd", "libmrsim.so")
    if os.path.exists(lib_so):
        return
    # git-ignored build product missing in this checkout: build it (no fallback path)
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        import __graft_entry__
        __graft_entry__.build()
    else:
        t_wait = time.time()
        while not os.path.exists(lib_so) and time.time() - t_wait < 600:
            time.sleep(1.0)
        time.sleep(2.0)


# ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline(cfg, seed, target_seconds):
    """The oracle (C restatement, kind "port") timed on this box's host cores on a bounded sample of
    the same workload: n_cpu envs, same config, random policy, auto-reset."""
    from oracle import oracle as O
    from tests.util import orc_params_from_cfg
    cores = len(os.sched_getaffinity(0))
    threads = max(1, min(cores, 64))
    n = 16384 * threads if threads <= 16 else 262144
    orc = O.VecOracle(n, orc_params_from_cfg(cfg), seed=seed, threads=threads)
    orc.reset(0)
    lo, hi = cfg.policy_low, cfg.policy_high
    a = orc.random_policy(1, lo, hi)
    orc.step(a, 1)  # warm
    t0 = time.perf_counter()
    steps, k = 0, 2
    while True:
        a = orc.random_policy(k, lo, hi)
        orc.step(a, k)
        k += 1
        steps += 1
        el = time.perf_counter() - t0
        if el >= target_seconds or steps >= 2000:
            break
    out = {"value": n * steps / el, "unit": "env-steps/s", "cores": threads, "kind": "port",
           "sample": f"{n} envs x {steps} steps ({el:.1f} s) of the same workload, oracle/mrsim_oracle.c "
                     f"with OpenMP over {threads} host threads"}
    # the reference's OWN Python path beside every GPU number.  The reference's files never
    # travel to the GPU box, so this is a QUOTED record: measured in the build container by tools/ref_python_baseline.py and
    # committed under profiles/ (hardware, versions and command inside); nothing of it is executed here.
    f = newest_profile("ref_python_baseline.json")
    if f is not None:
        try:
            ref = json.load(open(f))
            ref["quoted_from"] = os.path.relpath(f, ROOT)
            out["reference_python"] = ref
        except Exception:
            pass
    return out


def newest_profile(name):
    import glob
    f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)), reverse=True)
    return f[0] if f else None


def profiled_config_matches(args, n_local):
    """PMC counters cannot be read from inside the timed process: the committed passes describe a few configurations (the default
    one; the mismatched model; N = 2 097 152 -- matched per record by committed_traffic / committed_valu); any other run gets
    null instead of somebody else's counters."""
    return (args.sigma == 1.0 and args.noise_math == "fast" and n_local in (262144, 2097152) and args.obs_layout == "aos"
            and args.workload == "ddpg" and (args.mode == "step" or args.rollout_len == 51))


def loaded_lib_sha16():
    """sha256 prefix of the libmrsim.so this process runs -- committed counters describe ONE build of the kernels"""
    import hashlib
    from mr_rl_amd import _lib
    return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]


def counters_describe_this_build(d):
    """(ok, why_not): committed PMC records carry the sha256 prefix of the library they were taken with; counters of another
    build are somebody else's kernel and are not quoted"""
    want = d.get("libmrsim_so_sha16")
    have = loaded_lib_sha16()
    if want is None or want != have:
        return False, "committed counters describe libmrsim.so %s, this run loaded %s: not quoted" % (want, have)
    return True, None


def committed_traffic(args, n_local, law=None):
    """(HBM bytes per launch, source, why_none) of the dominant kernel from the newest committed rocprofv3 PMC passes
    (profiles/rNN/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this same command and
    corrected as MI355X_MICROARCH.md prescribes) -- only for the profiled configuration and the very build that was profiled."""
    law = law or args.noise_law
    f = newest_profile("pmc_traffic.json")
    if f is None or not profiled_config_matches(args, n_local):
        return None, None, "no committed counters for this configuration"
    key = "mr_rollout_kernel" if args.mode == "rollout" else "mr_step_kernel"
    try:
        d = json.load(open(f))
        ok, why = counters_describe_this_build(d)
        if not ok:
            return None, None, why
        for name, k in d["kernels"].items():
            if not (name.startswith(key) and bool(k.get("mismatched", False)) == bool(args.mismatched) and int(k.get("N", 262144)) == n_local):
                continue
            if args.mode == "step" or (k.get("carry", "f32") == args.carry and k.get("noise_law", "per_stage") == law):
                return int(k["hbm_bytes_per_launch"]), os.path.relpath(f, ROOT), None
    except Exception as exc:
        return None, None, "unreadable %s: %s" % (f, exc)
    return None, None, "no record for this kernel in %s" % os.path.relpath(f, ROOT)


def modelled_traffic(n_local, T):
    """What one launch of the fused rollout must move when nothing is read or written twice: every transition (obs 20 +
    action 8 + reward 4 + done 1 = 33 B per env-step), the env state in and out (pos 16 + aux 16 + ep_ret 4 each way) and the
    episode return / length of envs whose episode ended (8 B; every env once per 51 steps).  The PMC passes of rounds 2-3
    measured 1.01 x this figure."""
    return n_local * (33 * T + 2 * 36 + 8)


def committed_valu(args, n_local, T, law=None):
    """VALU-issue roofline of the rollout kernel from the committed counters (profiles/rNN/pmc_valu.json, written by
    tools/collect_profiles.py): the kernel's instruction mix per wave-step (rocprofv3 --pmc SQ_INSTS_VALU_*) priced with the
    per-instruction issue costs tools/instbench measures at the launch's occupancy, against the kernel's own SQ_WAVE_CYCLES
    -- shader cycles on both sides, so the clock the chip holds under load cancels.  Only for the profiled configuration."""
    law = law or args.noise_law
    f = newest_profile("pmc_valu.json")
    if f is None or args.mode != "rollout" or not profiled_config_matches(args, n_local) or n_local != 262144:
        return None
    try:
        d = json.load(open(f))
        if not counters_describe_this_build(d)[0]:
            return None
        k = d["kernels"]["rollout_" + args.carry + ("" if law == "per_stage" else "_" + law) + ("_mismatched" if args.mismatched else "")]
        return {"valu_issue_frac": k["valu_issue_frac"], "valu_issue_frac_at_spec_rates": k.get("valu_issue_frac_at_spec_rates"),
                "insts_valu_per_wave_step": k["insts_valu_per_wave_step"],
                "issue_floor_cycles_per_wave_step": k["issue_floor_cycles_per_wave_step"],
                "wave_cycles_per_wave_step": k["wave_cycles_per_wave_step"], "waves_per_simd": d["waves_per_simd"],
                "floor_us_per_launch_at_burst_clock": k["floor_us_per_launch_at_burst_clock"],
                "costs": d["issue_costs"]["source"], "source": os.path.relpath(f, ROOT)}
    except Exception:
        return None


def measure_step_path(cfg, n_local, dev, seed, steps=10200, samples=204):
    """Secondary figure reported beside the headline: the same workload driven through the drop-in gym loop,
    one launch per MR_Env.step(), hipGraph of 51 steps: [policy kernel -> actions in HBM] + [step kernel] (`value`), and
    with the exploration policy drawn inside the step kernel (`policy_in_step_kernel`, one kernel per step).  (A third
    form, step t+1's policy kernel on a second captured stream beside step t -- capture_steps(policy="overlap") -- is
    slower: graph replay pays ~5 us per cross-stream dependency, 14.7 vs 9.1 us per step; not measured here.)"""
    import torch
    from mr_rl_amd import MRVecEnv
    ep = cfg.max_timesteps + 1
    k = (steps // ep) * ep

    def run(policy):
        env = MRVecEnv(n_local, cfg=cfg, device=dev, seed=seed)
        env.reset()
        graph = env.capture_steps(ep, policy=policy)
        for _ in range(100):  # ~45 ms: lets the GPU clocks settle
            graph.replay()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps // ep):
            graph.replay()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        env.check_status()
        return env, el

    env, el = run("kernel")
    act = torch.empty((n_local, 2), dtype=torch.float32, device=dev)
    ms = sorted(env.step_timed(env.random_policy(out=act)) for _ in range(samples))
    avg_ms = sum(ms) / len(ms)
    ach = n_local * ALGO_BYTES_PER_ENV_STEP / (avg_ms * 1e-3) / 1e9
    _, el_f = run("fused")
    _, el_e = run("episode")
    return {"mode": "step (one launch per env.step, hipGraph of 51 steps, policy kernel + step kernel)",
            "value": n_local * k / el, "unit": "env-steps/s", "steps": k, "ms_per_step": el / k * 1e3,
            "kernel": "mr_step_kernel", "avg_kernel_us": round(avg_ms * 1e3, 3),
            "roofline_frac": round(ach / HBM_PEAK_GBS, 4),
            "policy_in_step_kernel": {"value": n_local * k / el_f, "unit": "env-steps/s", "ms_per_step": el_f / k * 1e3},
            # still a separate policy kernel writing actions to HBM, but ONE launch per episode draws all 51 rows (the
            # exploration policy reads no state); an observation-dependent actor cannot do this
            "policy_kernel_once_per_episode": {"value": n_local * k / el_e, "unit": "env-steps/s",
                                               "ms_per_step": el_e / k * 1e3}}


def mixed_goal_table(cfg, seed):
    """BASELINE config 5's "mixed trajectory set": env_id mod 3 -> straight line / figure eight / random waypoints,
    one goal per episode step; switches the env to the goal reward (calculate_reward, MR_env.py:118-134)."""
    import numpy as np
    Tg = cfg.max_timesteps + 2
    k = np.arange(Tg)
    th = 2 * np.pi * k / Tg
    tab = np.zeros((3, Tg, 2), dtype=np.float32)
    tab[0, :, 0] = 110 + 0.3 * k; tab[0, :, 1] = 110 + 0.3 * k                           # straight line
    tab[1, :, 0] = 110 + 8 * np.sin(th); tab[1, :, 1] = 110 + 8 * np.sin(th) * np.cos(th)  # figure eight
    tab[2] = np.random.default_rng(seed).uniform(100, 120, (Tg, 2))                        # random waypoints
    cfg.reward_mode, cfg.min_dist2goal = "goal", 1.0
    return tab


class RolloutRegion:
    """Timed regions of the fused-rollout workload on a RolloutCollector (S sub-shard launches per episode on S HIP
    streams): `steps` env steps in launch groups of <= T cut at episode boundaries, the returns of every episode
    all-gathered (gather() is called at every episode boundary and starts one collective per --gather-interval episodes), barrier + synchronize on both sides, max over ranks.  With S = 1, rank 0 can attach a HIP event pair to
    every full-length dispatch (non-blocking, hipExtLaunchKernelGGL on the launch stream)."""

    def __init__(self, col, gatherer, ep, world, dev, dist_backend):
        self.col, self.g, self.T, self.ep = col, gatherer, col.T, ep
        self.world, self.dev, self.backend = world, dev, dist_backend
        self.done_steps = 0
        self.launches = 0

    def run(self, nsteps, pool=None, used=None):
        col, T, ep = self.col, self.T, self.ep
        left = nsteps
        while left > 0:
            chunk = min(left, T, ep - (self.done_steps % ep))
            ev = None
            if pool is not None and chunk == T and col.S == 1 and len(used) < len(pool):
                ev = [pool[len(used)]]
                used.append(ev[0])
            col.collect(events=ev, steps=chunk)
            self.launches += len(col.shards)
            self.done_steps += chunk
            left -= chunk
            # after EVERY launch group (normally = one episode): the gatherer counts them and all-gathers a block of
            # returns per --gather-interval groups
            self.g.gather()

    def schedule(self, nsteps):
        """The launch-group lengths run(nsteps) will use from here."""
        out, done, left = [], self.done_steps, nsteps
        while left > 0:
            chunk = min(left, self.T, self.ep - (done % self.ep))
            out.append(chunk)
            done += chunk
            left -= chunk
        return out

    def barrier(self):
        import torch
        import torch.distributed as dist
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize(self.dev)  # device-wide: the sub-shard streams included

    def timed(self, nsteps, pool=None, used=None):
        """-> (seconds: max over ranks, launches in the region)"""
        import torch
        import torch.distributed as dist
        # launch argument blocks (ctypes structs: one per sub-shard, buffer set, returns row and group length) are built
        # once and reused; build the ones this region needs before the clock starts, as any steady loop has them already
        # (a 20-step region would otherwise spend half its wall time building two structs it uses once)
        self.col.prime(self.schedule(nsteps))
        gc_was_on = gc.isenabled()
        gc.disable()         # no collector pause inside the region (a 20-step region is ~80 us of wall time)
        self.barrier()
        l0 = self.launches
        t0 = time.perf_counter()
        self.run(nsteps, pool, used)
        t1 = time.perf_counter()
        self.g.finish()  # outstanding async all-gathers belong to the timed region
        torch.cuda.synchronize(self.dev)
        t2 = time.perf_counter()
        el = t2 - t0         # this rank's clock stops when ITS K steps (and its collectives) are complete on its device ...
        if self.world > 1:
            self.barrier()   # ... the closing barrier + synchronize follows; the job's time is the MAX of the ranks' clocks (below),
        t3 = time.perf_counter()   # all started together by the opening barrier.  (A NCCL barrier is itself a collective of tens of
        if gc_was_on:              # microseconds: inside the clock it would be most of a 20-step region at N = 8.)
            gc.enable()
        # where this rank's wall time went: host enqueue (launch argument blocks, event records, collectives), waiting for
        # the GPU, closing barrier -- a region of one short launch group is mostly the first and the last
        self.last_phases_us = {"enqueue": round((t1 - t0) * 1e6, 1), "wait_gpu": round((t2 - t1) * 1e6, 1),
                               "closing_barrier_untimed": round((t3 - t2) * 1e6, 1)}
        self.last_per_rank_s = [el]
        if self.world > 1:
            cdev = self.dev if self.backend == "nccl" else "cpu"
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            every = torch.zeros(self.world, dtype=torch.float64, device=cdev)
            dist.all_gather_into_tensor(every, t)      # each rank's own clock around the region ...
            dist.all_reduce(t, op=dist.ReduceOp.MAX)   # ... and the contract's figure: the slowest rank
            el = float(t.item())
            self.last_per_rank_s = [float(x) for x in every.tolist()]
        return el, self.launches - l0


def make_region(args, cfg, n_local, env_id0, world, dev, seed, streams, goal_table=None, T=None):
    from mr_rl_amd.collector import BlockReturnGatherer, RolloutCollector
    ep = cfg.max_timesteps + 1
    col = RolloutCollector(n_local, cfg=cfg, device=dev, seed=seed, env_id0=env_id0, goal_table=goal_table,
                           streams=streams, T=T if T is not None else ep, carry=args.carry,
                           returns_interval=args.gather_interval)
    col.reset()
    g = BlockReturnGatherer(col, world, force_collective=bool(os.environ.get("MRSIM_BENCH_FORCE_DIST")))
    return RolloutRegion(col, g, ep, world, dev, args.dist_backend)


def hwmon_of(dev):
    """hwmon directory of torch device `dev`, matched by PCI address (the host shows the sensors of all its GPUs)"""
    import glob
    import torch
    try:
        pr = torch.cuda.get_device_properties(dev)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        return None
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        if os.path.basename(os.path.realpath(os.path.join(h, "..", ".."))) == bdf and os.path.exists(h + "/power1_input"):
            return h
    return None


def measure_power(reg, dev, total, seconds=1.5):
    """Package power (hwmon power1_input = PPT, against power1_cap) and the driver's shader-clock reading while the headline
    workload runs for `seconds` on its collector: the rollout kernel runs at the package power limit, which is what sets
    its clock (tools/power_probe.py is the stand-alone form).  The sensor averages over a few hundred
    milliseconds, hence a leg of its own; the mean of the second half of the window is reported.  None if the sensors
    are not readable."""
    import threading
    import torch
    h = hwmon_of(dev)
    if h is None:
        return None

    def rd(name):
        try:
            return int(open(os.path.join(h, name)).read())
        except (OSError, ValueError):
            return None
    if rd("power1_input") is None:
        return None
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append((time.perf_counter(), rd("power1_input"), rd("freq1_input")))
            time.sleep(0.02)
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds:
        reg.run(400 * reg.ep)
        torch.cuda.synchronize(dev)
        steps += 400 * reg.ep
    t1 = time.perf_counter()
    stop[0] = True
    th.join()
    load = [x for x in samples if t0 + 0.5 * (t1 - t0) < x[0] < t1 and x[1] is not None]
    if not load:
        return None
    cap = rd("power1_cap")
    clk = [x[2] for x in load if x[2] is not None]
    return {"package_W": round(sum(x[1] for x in load) / len(load) * 1e-6, 1), "cap_W": round(cap * 1e-6, 1) if cap else None,
            "sclk_MHz_driver_reading": round(sum(clk) / len(clk) * 1e-6) if clk else None,
            "window_s": round(t1 - t0, 3), "samples": len(load), "value_in_window": total * steps / (t1 - t0),
            "what": "hwmon power1_input (PPT) of this GPU, mean of the second half of a window in which the headline "
                    "workload runs back to back; at the cap the kernel's clock is what the power limit leaves"}


def measure_mixed_set(args, n_local, env_id0, world, dev, seed, streams, steps=10200):
    """Secondary figure (every rank takes part, same barrier / max-over-ranks protocol as the headline): the fused
    rollout on BASELINE config 5's mixed straight-line / figure-eight / random-waypoint trajectory set with the
    goal reward, returns all-gathered once per episode."""
    from mr_rl_amd import MRConfig
    cfg = MRConfig(noise_var=args.sigma, auto_reset=True, obs_layout=args.obs_layout, noise_math=args.noise_math,
                   seed=seed, is_mismatched=args.mismatched, noise_law=args.noise_law)
    tab = mixed_goal_table(cfg, seed)
    reg = make_region(args, cfg, n_local, env_id0, world, dev, seed, streams, goal_table=tab)
    ep = reg.ep
    reg.run(PREROLL_EPISODES * ep)
    k = (steps // ep) * ep
    el, launches = reg.timed(k)
    reg.col.check_status()
    return {"workload": "BASELINE config 5 trajectory set: env_id mod 3 -> straight line / figure eight / random "
                        "waypoints (goal table), goal reward, same policy / noise / outputs as the headline",
            "value": n_local * world * k / el, "unit": "env-steps/s", "steps": k, "launches": launches,
            "streams": reg.col.S, "ms_per_step": el / k * 1e3, "mean_episode_return": reg.g.last_mean()}


ACTOR_FLOP_PER_ENV_STEP = 2 * (5 * 64 + 64 * 64 + 64 * 2)   # RL/MR_ddpg.py:120-137: 5 -> 64 -> 64 -> 2
MFMA_F32_PEAK_TFLOPS = 157.3                                # MI355X_MICROARCH.md: f32-input MFMA, dense
MFMA_BF16_PEAK_TFLOPS = 2500.0                              # same table: bf16 MFMA, dense


def measure_actor_in_loop(args, n_local, dev, seed, streams, episodes=150, event_episodes=60):
    """The DDPG collection loop with an OBSERVATION-DEPENDENT policy (RL/MR_ddpg.py:270-311: action = actor.predict(state)
    + actor_noise(); env.step(action)): the reference's actor architecture, random-initialised, evaluated inside the
    fused rollout kernel (mr_rl_amd.actor.DeviceActor as the collector's policy), OU noise on, every transition written.
    Wall-clock rate over `episodes` episodes after a pre-roll, then the same on ONE stream with a HIP event pair on every
    dispatch: the kernel durations behind the `mfma` roofline object (the two 64-wide layers are f32-input MFMA)."""
    import torch
    from mr_rl_amd import MRConfig
    from mr_rl_amd._lib import EventPair
    from mr_rl_amd.actor import DeviceActor
    from mr_rl_amd.collector import RolloutCollector
    from mr_rl_amd.ddpg import Actor
    torch.manual_seed(seed)
    module = Actor().eval()
    actor = DeviceActor.from_module(module, obs_scale=[0.01] * 5, device=dev)
    cfg = MRConfig(noise_var=args.sigma, auto_reset=True, obs_layout=args.obs_layout, noise_math=args.noise_math, seed=seed,
                   is_mismatched=args.mismatched, noise_law=args.noise_law)
    ep = cfg.max_timesteps + 1

    def run(S, n, events=None, actor=actor):
        col = RolloutCollector(n_local, cfg=cfg, device=dev, seed=seed, streams=S, carry=args.carry, policy=actor)
        col.reset()
        for _ in range(40):
            col.collect(); col.ready(); col.release()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(n):
            col.collect(events=None if events is None else [events[k]]); col.ready(); col.release()
        col.join()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        col.check_status()
        return el
    el = run(streams, episodes)
    pool = [EventPair() for _ in range(event_episodes)]
    el1 = run(1, event_episodes, pool)
    ms = [p.elapsed_ms() for p in pool]
    for p in pool:
        p.close()
    avg_us, med_us = stats_us(ms)
    tflops = n_local * ep * ACTOR_FLOP_PER_ENV_STEP / (avg_us * 1e-6) / 1e12
    # the same with the 64 x 64 layer in bf16 x 3 arithmetic (three bf16 terms per f32 operand, six bf16 MFMAs, f32 accumulate)
    # and in plain bf16 (one term, one MFMA)
    waves = (n_local + 63) // 64

    def variant(math, mfma_bf16_per_wave_step, what, bound_note):
        pol = DeviceActor.from_module(module, obs_scale=[0.01] * 5, device=dev, math=math)
        el_v = run(streams, episodes, actor=pol)
        pool_v = [EventPair() for _ in range(event_episodes)]
        el1_v = run(1, event_episodes, pool_v, actor=pol)
        ms_v = [p.elapsed_ms() for p in pool_v]
        for p in pool_v:
            p.close()
        avg_v, med_v = stats_us(ms_v)
        exec_flops = waves * ep * mfma_bf16_per_wave_step * 32 * 32 * 16 * 2     # both hidden layers run on the bf16 matrix cores
        return {"what": what, "value": n_local * ep * episodes / el_v, "unit": "env-steps/s", "ms_per_step": el_v / (episodes * ep) * 1e3,
                "one_stream_with_events": {"value": n_local * ep * event_episodes / el1_v, "avg_kernel_us": round(avg_v, 2),
                                           "median_kernel_us": round(med_v, 2)},
                "roofline": {"bound": "mfma", "achieved": round(exec_flops / (avg_v * 1e-6) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(exec_flops / (avg_v * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                             "traffic": None, "note": bound_note,
                             "algorithmic_actor_TFLOPs": round(n_local * ep * ACTOR_FLOP_PER_ENV_STEP / (avg_v * 1e-6) / 1e12, 1)}}
    note = ("EXECUTED matrix flops (%d bf16 MFMAs of 32x32x16 per wave and step, layer 1 included) / kernel duration against the dense "
            "bf16 MFMA peak; the kernel is bound by vector-instruction issue at the package power cap (operand conversion, ReLU, the "
            "output layer, the env step), not by the matrix pipe")
    bf16x3 = variant("bf16x3", 104, "the same with MrsimActor.math = BF16X3: every f32 operand of the two hidden layers as three bf16 terms, "
                     "the six products above 2^-24 on v_mfma_f32_32x32x16_bf16 with f32 accumulation (within 5e-6 of the action bound of "
                     "the f32 result; tests/test_gpu_actor.py) -- the matrix cores proper, which run beside the vector unit", note % 104)
    bf16 = variant("bf16", 20, "the same with MrsimActor.math = BF16: plain bf16 operands, f32 accumulation -- ordinary bf16 inference "
                   "(bitwise equal to the oracle's bf16 emulation up to accumulation order; tests/test_gpu_actor.py bounds the action's distance from the f32 result at 6e-2 of its bound): exploration-grade collection", note % 20)
    return {"bf16x3": bf16x3, "bf16": bf16, "what": "BASELINE config 4 with the reference's DDPG actor (5-64-64-2, eval-mode batch norm folded, tanh x bound) "
                    "+ OU noise as the policy, evaluated INSIDE the fused rollout kernel on each step's observation "
                    "(RL/MR_ddpg.py:277-278 without leaving the registers); random-initialised weights, every transition written",
            "value": n_local * ep * episodes / el, "unit": "env-steps/s", "episodes": episodes, "streams": streams,
            "ms_per_step": el / (episodes * ep) * 1e3,
            "one_stream_with_events": {"value": n_local * ep * event_episodes / el1, "avg_kernel_us": round(avg_us, 2),
                                       "median_kernel_us": round(med_us, 2)},
            "roofline": {"bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tflops / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                         "flop_per_env_step": ACTOR_FLOP_PER_ENV_STEP,
                         "kernel": "mr_rollout_actor_fl_kernel<RK45,%s,%s,DDPG|carry64|actor|OU>" %
                                   ("nonoise" if args.sigma == 0 else args.noise_math, "mismatched" if args.mismatched else "nominal"),
                         "note": "algorithmic flops of the actor (2 x 4544 multiply-adds per env-step) / the kernel's average "
                                 "duration (HIP events on the dispatches) against the dense f32-input MFMA peak; 97 % of the "
                                 "flops (layers 1, 2) run on v_mfma_f32_32x32x2_f32, the env step runs on the vector unit "
                                 "beside them"}}


def measure_learner(args, n_local, dev, seed, streams):
    """The learner half of the DDPG loop (RL/MR_ddpg.py:288-305: sample 64 -> critic target -> critic step -> actor step -> two
    soft updates) and the loop as a whole.
    updates_per_s: the update on a filled 10 000-slot ring, batch 64 -- eager PyTorch (no host sync any more), the same captured
    as ONE hipGraph, and libmrsim's fused kernel (mrsim_ddpg_update: one launch, rows drawn in-kernel).
    end_to_end: DDPG(fused=True).train_collected on n_local envs -- every episode one fused launch group of the rollout kernel
    with the agent's actor (+ OU noise) in the kernel, 4096 of its transitions into the ring, U learner updates, parameters
    folded / packed / uploaded on the device, collection of episode k + 1 beside the updates of episode k -- env-steps/s
    INCLUDING the learner, at the stated update : transition ratio (the reference does 1 : 1 on one env; with N lockstep envs
    the updates are the sequential part)."""
    import torch
    from mr_rl_amd import MRConfig, MRVecEnv
    from mr_rl_amd.ddpg import DDPG
    cfg = MRConfig(noise_var=args.sigma, auto_reset=True, obs_layout=args.obs_layout, noise_math=args.noise_math, seed=seed,
                   is_mismatched=args.mismatched, noise_law=args.noise_law)
    env = MRVecEnv(n_local, cfg=cfg, device=dev, seed=seed)
    scale = [0.01] * 5

    def filled(**kw):
        ag = DDPG(env, seed=seed, obs_scale=scale, **kw)
        g = torch.Generator(device=dev).manual_seed(1)
        n = 10000
        s = torch.randn(n, 5, device=dev, generator=g)
        ag.buffer.add(s, torch.randn(n, 2, device=dev, generator=g), torch.randn(n, device=dev, generator=g),
                      (torch.rand(n, device=dev, generator=g) < 0.02).float(), s + 0.01 * torch.randn(n, 5, device=dev, generator=g))
        return ag

    def rate(f, n, chunk=256):
        # (bounded launch queue, as everywhere in this script: thousands of launches enqueued without a wait leave the runtime a
        # backlog whose retirement the later launch calls pay for -- the same leg read 14 k and 23 k updates/s on two boxes)
        f(10)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for c0 in range(0, n, chunk):
            f(min(chunk, n - c0))
            torch.cuda.synchronize(dev)
        return n / (time.perf_counter() - t0)
    ag = filled()
    ups = {"eager_pytorch": rate(lambda n: [ag.update() for _ in range(n)], 100)}
    ag = filled()
    ups["hipgraph_replay"] = rate(lambda n: ag.update_graphed(n), 1000)
    ag = filled(fused=True)
    ups["fused_kernel"] = rate(lambda n: [ag.update() for _ in range(n)], 3000)
    # the form train_collected uses: a burst of updates per launch (the kernel loops; one host call per burst)
    ups["fused_kernel_8_per_launch"] = rate(lambda n: [ag.update_graphed(8) for _ in range(n // 8)], 3200)
    # large batches: batch / 64 workgroups on as many compute units (two launches per update) against the single workgroup looping
    # over the tiles -- what lets a 262 144-env collector's data be used at more than one 64-row batch per ~40 us
    big = {}
    for B in (64, 256, 1024, 4096):
        row = {}
        for multi in ((True, False) if B > 64 else (False,)):
            agb = filled(fused=True, min_batch=B)
            agb.fused.multi_workgroup = multi
            n_upd = 400 if (multi or B <= 256) else 60
            r_ = rate(lambda n: [agb.update() for _ in range(n)], n_upd)
            row["multi_workgroup" if multi else "single_workgroup"] = {"updates_per_s": round(r_, 1), "us_per_update": round(1e6 / r_, 1),
                                                                       "transitions_per_s": round(r_ * B, 1)}
        big[str(B)] = row
    out = {"what": "the DDPG learner: updates/s on a filled ring (batch 64, the reference's) and env-steps/s of the whole loop "
                   "(collection with the actor in the kernel + learner) at a stated update : transition ratio",
           "batch": 64, "updates_per_s": {k: round(v, 1) for k, v in ups.items()},
           "us_per_update": {k: round(1e6 / v, 1) for k, v in ups.items()}, "by_batch": big, "end_to_end": []}
    # the reference's own loop shape (RL/MR_ddpg.py:262-311: act, env.step, replay add, ONE update -- per env step), DDPG.train:
    # step kernel with the actor inside + mrsim_replay_add_step + mrsim_ddpg_update (+ the policy upload), N envs in lockstep
    loop = {"what": "DDPG.train: one env step, one replay add, one learner update (64 rows) per iteration, as RL/MR_ddpg.py:262-311 does "
                    "for one env; per-step bookkeeping as PyTorch statements (rounds 2-4), as ONE launch (mrsim_replay_add_step), and inside "
                    "the step kernel (MrsimStepIO.replay: no launch of its own)",
            "rows": []}
    for n_loop in (1, 256, 4096):
        for fb in (False, "add_step", True):
            envl = MRVecEnv(n_loop, cfg=MRConfig(noise_var=0.1, reward_mode="goal", auto_reset=True, init_low=(20.0, 20.0),
                                                 init_high=(30.0, 30.0), min_dist2goal=25.0, seed=seed), device=dev, seed=seed,
                            track_actions=True)
            agl = DDPG(envl, seed=seed, obs_scale=[0.01, 0.01, 0.01, 0.01, 1.0], device_actor=True, fused=True,
                       buffer_size=max(10000, 4 * n_loop))
            agl.train(200, fused_bookkeeping=fb)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            agl.train(1500, fused_bookkeeping=fb)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            agl.close()
            loop["rows"].append({"envs": n_loop, "bookkeeping": {False: "pytorch statements", "add_step": "one launch", True: "in the step kernel"}[fb],
                                 "us_per_iteration": round(dt / 1500 * 1e6, 1), "iterations_per_s": round(1500 / dt, 1),
                                 "env_steps_per_s": round(1500 * n_loop / dt, 1)})
    out["reference_shaped_loop"] = loop
    ep = cfg.max_timesteps + 1
    # learner_cus = 1: the learner's launches on 8 compute units of their own (one per XCC), the collection on the other 248
    # (mr_rl_amd.partition) -- reported beside the shared-device rows
    # (actor arithmetic, updates per episode, episodes, learner compute units per XCC or 0 = shared, rows per update)
    rows = [("f32", 8, 60, 0, 64), ("bf16", 8, 150, 0, 64), ("bf16", 8, 150, 1, 64), ("bf16", 0, 150, 0, 64),
            ("bf16", 8, 150, 0, 4096)]      # the same loop learning from 4 096 rows per update (batch / 64 workgroups per update)
    if getattr(args, "no_partition_row", False):
        rows = [r for r in rows if r[3] == 0]
    for math, U, episodes, cus, batch in rows:
        agent = DDPG(env, seed=seed, obs_scale=scale, fused=True, min_batch=batch)
        st = {}
        rets = agent.train_collected(episodes, updates_per_episode=U, sample=4096, streams=streams, math=math, stats=st, warm_episodes=10,
                                     learner_cus=cus)
        agent.close()    # the partition's streams go last, after everything that names them (DDPG.close; profiles/r05/partition_kt.txt)
        out["end_to_end"].append({
            "actor_math": math, "updates_per_episode": U, "rows_per_update": batch, "learner_compute_units": 8 * cus if cus else "shared",
            "rows_learnt_from_per_transition_collected": U * batch / float(n_local * ep),
            "transitions_per_episode": n_local * ep,
            "update_to_transition_ratio": U / float(n_local * ep), "value": st["env_steps_timed"] / st["seconds"], "unit": "env-steps/s",
            "updates_per_s": st["updates_timed"] / st["seconds"], "episodes_timed": st["episodes_timed"],
            "mean_return_last_episode": rets[-1] if rets else None})
    return out


def trajectory_rmse(dev, carry):
    """Second half of BASELINE's metric: trajectory RMSE vs the CPU reference, on the committed golden trajectories
    the reference itself produced (tests/golden/ref_sim.npz, sigma = 0, 1000-2000 steps each), through the same
    fused kernel the timed region runs.  `value` feeds the reference float32-rounded action tables (what the golden
    files recorded the reference with); `f64_tables` feeds the reference's own float64 tables (ref_sim_f64.npz, the
    main.py:14-50 profiles unrounded) through the fp64 action-table input of the rollout."""
    import numpy as np
    from mr_rl_amd import MRConfig, MRVecEnv

    def run(fname, f64):
        g = np.load(os.path.join(ROOT, "tests", "golden", fname))
        names = sorted({k.split("/")[0] for k in g.files})
        worst = 0.0
        for name in names:
            G = {k.split("/")[1]: g[k] for k in g.files if k.startswith(name + "/")}
            env = MRVecEnv(64, cfg=MRConfig(noise_var=0.0, a0=float(G["a0"]), is_mismatched=bool(G["mismatched"])), device=dev)
            env._prev_mismatched = bool(G["mismatch_at_reset"])
            env.reset(init=np.tile(G["init"][None, :], (64, 1)), is_mismatched=bool(G["mismatched"]))
            acts = G["actions"].astype(np.float64 if f64 else np.float32)
            traj = env.rollout(len(acts), actions=acts, shared_actions=True, want=("traj",), carry=carry)["traj"][:, 0, :].cpu().numpy()
            worst = max(worst, float(np.sqrt(np.mean(np.sum((traj - G["pos"]) ** 2, axis=1)))))
        return worst, len(names)

    worst, n = run("ref_sim.npz", False)
    out = {"value": worst, "unit": "position units (max over %d golden trajectories)" % n, "target": 1e-5,
           "fixtures": "tests/golden/ref_sim.npz", "carry": carry}
    if os.path.exists(os.path.join(ROOT, "tests", "golden", "ref_sim_f64.npz")):
        w64, n64 = run("ref_sim_f64.npz", True)
        out["f64_tables"] = {"value": w64, "trajectories": n64, "fixtures": "tests/golden/ref_sim_f64.npz"}
    return out


def rollout_roofline(args, n_local, T, law, avg_us, med_us, timed_where):
    """`roofline` object of the fused rollout kernel under noise law `law`, from its average launch duration (HIP events on
    the dispatches of a one-stream region of THIS run).
    bound = "hbm": achieved = the HBM bytes one launch really moves / that duration.  The bytes come from the committed
    rocprofv3 --pmc passes of this very build (`traffic`, profiles/rNN/pmc_traffic.json; refused when the library hash differs)
    or, failing that, from the minimal-traffic model (`traffic_source` says which; the PMC passes measured 1.01 x the model).
    SURVEY 8(d)'s algorithmic 97 B per env-step (state round-trips HBM every step) is NOT what a fused launch moves -- the
    state stays in registers for all T steps -- and is kept as the labelled `algorithmic_equiv`, never as a fraction.
    `valu`: the kernel's vector-issue floor from the committed SQ counters, the other resource the kernel leans on."""
    units = n_local * T
    traffic, traffic_src, why = committed_traffic(args, n_local, law)
    valu = committed_valu(args, n_local, T, law)
    moved, moved_src = (traffic, traffic_src) if traffic else (modelled_traffic(n_local, T), "model: 33 B per env-step of transitions + "
                                                               "state in/out + episode returns (bench.py: modelled_traffic)")
    gbs = moved / (avg_us * 1e-6) / 1e9
    clock = valu["wave_cycles_per_wave_step"] * T / (avg_us * 1e3) if valu else None
    nz = "nonoise" if args.sigma == 0 else args.noise_math
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": traffic_src, "traffic_unavailable_because": why,
            "bytes_priced": moved, "bytes_priced_source": moved_src,
            "hbm_achievable_GBs": HBM_ACHIEVABLE_GBS, "frac_of_achievable": round(gbs / HBM_ACHIEVABLE_GBS, 4),
            "north_star_hbm_target": 0.60, "north_star_hbm_target_met": bool(gbs / HBM_PEAK_GBS >= 0.60),
            "noise_law": law,
            "in_kernel_env_steps_per_s": units / (avg_us * 1e-6),
            "valu": valu, "valu_issue_frac": valu["valu_issue_frac"] if valu else None,
            "valu_issue_frac_at_spec_rates": valu.get("valu_issue_frac_at_spec_rates") if valu else None,
            "in_kernel_clock_GHz": round(clock, 3) if clock else None,
            "algorithmic_equiv": {
                "what": "SURVEY 8(d)'s algorithmic bytes (state round-trips HBM every step) / kernel time; the fused kernel keeps "
                        "the state in registers and does NOT move these bytes: an equivalence figure, not a fraction of any peak",
                "bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP,
                "GBs": round(units * ALGO_BYTES_PER_ENV_STEP / (avg_us * 1e-6) / 1e9, 1)},
            "kernel": "mr_rollout_kernel<RK45,%s%s,%s,carry=%s>" % (nz, "" if law == "per_stage" else "+collapsed",
                                                                    "mismatched" if args.mismatched else "nominal", args.carry),
            "kernel_timed_over": timed_where, "avg_kernel_us": round(avg_us, 3), "median_kernel_us": round(med_us, 3),
            "env_steps_per_launch": units}


def measure_other_law(args, cfg, n_local, env_id0, world, dev, seed, streams, goal_table, T, law, episodes=400):
    """The same workload under the OTHER noise law (every rank takes part): sustained wall-clock rate through the collector on
    `streams` streams, then one launch per episode on ONE stream with a HIP event pair on every dispatch -- the kernel
    durations of that law's `roofline` object."""
    import copy
    from mr_rl_amd._lib import EventPair
    c2 = copy.copy(cfg)
    c2.noise_law = law
    ep = c2.max_timesteps + 1
    reg = make_region(args, c2, n_local, env_id0, world, dev, seed, streams, goal_table=goal_table, T=T)
    reg.run(PREROLL_EPISODES * ep)
    el, launches = reg.timed(episodes * ep)
    reg.col.check_status()
    out = {"noise_law": law, "value": n_local * world * episodes * ep / el, "unit": "env-steps/s", "steps": episodes * ep,
           "streams": reg.col.S, "launches": launches, "ms_per_step": el / (episodes * ep) * 1e3}
    del reg
    reg1 = make_region(args, c2, n_local, env_id0, world, dev, seed, 1, goal_table=goal_table, T=T)
    pool = [EventPair() for _ in range(episodes)] if int(os.environ.get("RANK", "0")) == 0 else None
    used = []
    reg1.run(PREROLL_EPISODES * ep)
    el1, _ = reg1.timed(episodes * ep, pool, used)
    reg1.col.check_status()
    ms = [e.elapsed_ms() for e in used]
    if pool is not None:
        for e in pool:
            e.close()
    out["one_stream_with_events"] = {"value": n_local * world * episodes * ep / el1}
    if ms:
        avg_us, med_us = stats_us(ms)
        out["roofline"] = rollout_roofline(args, n_local, T, law, avg_us, med_us,
                                           "the %d dispatches of a one-stream region of this leg" % len(ms))
    return out


def measure_facade(seed, episodes=40):
    """The literal drop-in call: mr_rl_amd.MR_Env.step() for ONE env, once per Python loop iteration, as utils.run_sim
    (utils.py:51-54) and the DDPG loop (RL/MR_ddpg.py:270-278) call the reference's MR_env.MR_Env.step (MR_env.py:70-98).
    One launch + one wait per step on a pinned host record (mr_rl_amd/env.py; the wait polls a word of the record the kernel stores
    after its last output).  `value` includes the reset of every
    51-step episode (the DDPG loop's shape); `phases_us` splits a step into the launch call, the wait and the Python around
    them; the reference's own rate is quoted from the committed measurement (the reference never travels to the GPU box)."""
    import ctypes as C
    from mr_rl_amd import MR_Env, _lib
    env = MR_Env(seed=seed)
    env.reset()
    a = [5.0, 1.0]
    for _ in range(500):
        env.step(a)
    n = 0
    t0 = time.perf_counter()
    for _ in range(episodes):
        env.reset()
        for _ in range(51):
            env.step(a)
            n += 1
    el = time.perf_counter() - t0
    # run_sim's shape: one reset, then steps whatever `done` says
    env.reset()
    t0 = time.perf_counter()
    for _ in range(1000):
        env.step(a)
    el_rs = time.perf_counter() - t0
    # where a step goes: the same two C calls the facade makes, timed apart
    L = env._L
    t_launch = t_wait = t_sync = 0.0
    for k in range(500):
        env._io.done_value = word = 1000 + k
        t1 = time.perf_counter()
        L.mrsim_step(env._pp, 1, env.env_id, C.byref(env._st), C.byref(env._io), env.seed_value, env.step_idx + k, None)
        t2 = time.perf_counter()
        L.mrsim_host_wait_word(env._word_host, word, 2000000)      # what the facade waits on: the record's own step word
        t3 = time.perf_counter()
        t_launch += t2 - t1; t_wait += t3 - t2
    env.step_idx += 500
    for k in range(500):       # the same with a stream wait instead (round 5's first form): what the kernel's completion signal costs
        env._io.done_value = 2000 + k
        L.mrsim_step(env._pp, 1, env.env_id, C.byref(env._st), C.byref(env._io), env.seed_value, env.step_idx + k, None)
        t2 = time.perf_counter()
        L.mrsim_stream_synchronize(None)
        t_sync += time.perf_counter() - t2
    env.step_idx += 500
    env._word = 2499
    out = {"what": "mr_rl_amd.MR_Env.step(action) for one env, one call per Python iteration (the drop-in for MR_env.py:70-98): one "
                   "launch + one wait per step on a pinned host record, no copy calls",
           "value": n / el, "unit": "MR_Env.step calls/s (one env; the reset of every 51-step episode included)",
           "us_per_step": round(el / n * 1e6, 2), "steps": n,
           "run_sim_shape": {"value": 1000 / el_rs, "us_per_step": round(el_rs / 1000 * 1e6, 2),
                             "what": "1000 consecutive steps without resets (utils.run_sim ignores done)"},
           "phases_us": {"mrsim_step_call": round(t_launch / 500 * 1e6, 2), "mrsim_host_wait_word": round(t_wait / 500 * 1e6, 2),
                         "python_around_them": round(el_rs / 1000 * 1e6 - (t_launch + t_wait) / 500 * 1e6, 2)},
           "wait_with_mrsim_stream_synchronize_us": round(t_sync / 500 * 1e6, 2)}
    env.close()
    f = newest_profile("ref_python_baseline.json")
    if f is not None:
        try:
            ref = json.load(open(f))
            out["reference_python"] = {"quoted_from": os.path.relpath(f, ROOT), "record": ref}
        except Exception:
            pass
    return out


def measure_streaming_point(args, cfg, dev, seed, n=2097152, episodes=60):
    """SURVEY H4's streaming point: N = 2 097 152 envs on ONE GPU (BASELINE config 5's total; 3.7 GB of transitions per 51-step
    launch, far beyond the 256 MB Infinity Cache): the fused rollout under both noise laws (one launch per episode on one stream,
    a HIP event pair on every dispatch) and the one-launch-per-step kernel, each against the HBM peak on the bytes it must move."""
    import copy
    import torch
    from mr_rl_amd import MRVecEnv
    from mr_rl_amd._lib import EventPair
    ep = cfg.max_timesteps + 1
    out = {"what": "N = 2 097 152 envs on one GPU (the size at which nothing fits the Infinity Cache): kernel durations from HIP events "
                   "on the dispatches, bytes = the minimal-traffic model N (33 T + 80) for the fused rollout (the PMC passes at N = 262 144 "
                   "measured 1.01 x it; profiles/rNN/pmc_traffic_2m.json holds this size's own counters when collected), SURVEY 8(d)'s 97 B "
                   "per env-step for the step kernel", "envs": n, "rollout": {}}
    for law in (args.noise_law, "per_stage" if args.noise_law == "collapsed" else "collapsed"):
        c2 = copy.copy(cfg)
        c2.noise_law = law
        reg = make_region(args, c2, n, 0, 1, dev, seed, 1, T=ep)
        pool, used = [EventPair() for _ in range(episodes)], []
        reg.run(40 * ep)
        el, _ = reg.timed(episodes * ep, pool, used)
        reg.col.check_status()
        avg_us, med_us = stats_us([e.elapsed_ms() for e in used])
        for e in pool:
            e.close()
        traffic, traffic_src, _why = committed_traffic(args, n, law)     # this size's own PMC passes, when committed for this build
        moved = traffic if traffic else modelled_traffic(n, ep)
        gbs = moved / (avg_us * 1e-6) / 1e9
        out["rollout"][law] = {"value": n * episodes * ep / el, "unit": "env-steps/s", "avg_kernel_us": round(avg_us, 2),
                               "median_kernel_us": round(med_us, 2), "bytes_per_launch": moved,
                               "bytes_source": traffic_src or "model N (33 T + 80)", "traffic": traffic,
                               "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": round(gbs / HBM_PEAK_GBS, 4)},
                               "in_kernel_env_steps_per_s": n * ep / (avg_us * 1e-6)}
        del reg
        torch.cuda.empty_cache()
    env = MRVecEnv(n, cfg=cfg, device=dev, seed=seed)
    env.reset()
    act = torch.empty((n, 2), dtype=torch.float32, device=dev)
    for _ in range(30):
        env.step(env.random_policy(out=act))
    ms = [env.step_timed(env.random_policy(out=act)) for _ in range(102)]
    avg_us, med_us = stats_us(ms)
    env.check_status()
    gbs = n * ALGO_BYTES_PER_ENV_STEP / (avg_us * 1e-6) / 1e9
    out["step_kernel"] = {"avg_kernel_us": round(avg_us, 2), "median_kernel_us": round(med_us, 2),
                          "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP,
                          "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(gbs / HBM_PEAK_GBS, 4)},
                          "in_kernel_env_steps_per_s": n / (avg_us * 1e-6)}
    del env, act
    torch.cuda.empty_cache()
    return out


def verify_ranks(args, cfg, world, rank, dev, seed, n_local, probe_envs=4096):
    """Self-verification of an N > 1 run, inside the record: (1) what the process group says its size is, (2) every rank's device
    identity (PCI address, UUID, name, host) gathered with all_gather_object, (3) a data check through the SAME backend the returns
    travel on: every rank runs one episode of the first `probe_envs` envs of ITS shard (global env ids env_id0 ..) and contributes
    {rank, env_id0, sum x, sum y, sum of returns} to an all_gather_into_tensor; rank 0 recomputes every shard's probe itself -- the
    RNG is keyed by the global env id, so the values must agree bit for bit -- and says whether they did."""
    import socket
    import torch
    import torch.distributed as dist
    from mr_rl_amd import MRVecEnv
    from mr_rl_amd.dist import shard_of

    def probe(env_id0):
        env = MRVecEnv(probe_envs, cfg=cfg, device=dev, seed=seed, env_id0=env_id0)
        env.reset()
        env.rollout(cfg.max_timesteps + 1, want=())
        v = torch.stack([env.pos[:, 0].sum(), env.pos[:, 1].sum(), env.final_ret.double().sum()])
        env.check_status()
        return v

    from mr_rl_amd.dist import verify_shards
    pr = torch.cuda.get_device_properties(dev)
    ident = {"rank": rank, "host": socket.gethostname(), "device_index": dev.index, "name": pr.name,
             "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
             "uuid": str(getattr(pr, "uuid", "")), "pid": os.getpid()}
    check = verify_shards(probe, n_local * world, rank, world, device=dev if args.dist_backend == "nccl" else "cpu")
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    else:
        idents = [ident]
    if rank != 0:
        return None
    ok = check["all_equal"]
    rows = [{"rank": r["rank"], "env_id0": r["env_id0"], "probe_sum_x": r["probe"][0], "probe_sum_y": r["probe"][1],
             "probe_sum_returns": r["probe"][2], "equals_rank0_recomputation": r["equals_rank0_recomputation"]} for r in check["per_rank"]]
    distinct = len({(d["host"], d["pci"], d["uuid"]) for d in idents})
    return {"world_size_from_process_group": dist.get_world_size() if (world > 1 or dist.is_initialized()) else 1,
            "backend": (dist.get_backend() if (world > 1 or dist.is_initialized()) else "none"),
            "devices": idents, "distinct_devices": distinct, "one_device_per_rank": bool(distinct == world or args.share_gpu),
            "probe": {"what": "%d envs of every rank's shard, one episode, sums of final positions and returns gathered with "
                              "all_gather_into_tensor over the run's backend and recomputed on rank 0 from the global env ids" % probe_envs,
                      "per_rank": rows, "all_equal": ok}}



def measure_consumers(dev, seed, n=65536, T=600):
    """The two data-format rows either side of the path, measured: the batched run_sim + velocity post-processing
    every consumer of run_sim applies (Learning_module.py:46-59: box filter -> gradient -> box filter -> drift; one fused launch over
    trajectories x time chunks: 32 algorithmic bytes per (t, env) of [T][n][2] fp64), and the batched MRExperiment export (recorder.export_all: the episodes of
    all envs of one resident rollout cut on the device).  Workload: one circle of main.py's learning set (600 steps at freq 4,
    main.py:21-33) for 65 536 envs at main.py's parameters."""
    import numpy as np
    import torch
    from mr_rl_amd import MRConfig, MRVecEnv, recorder
    from mr_rl_amd.rollout import estimate_velocity
    acts = np.stack([np.full(T, 4.0), np.linspace(-np.pi, np.pi, T)], 1)
    env = MRVecEnv(n, cfg=MRConfig(noise_var=0.5, a0=1.5, is_mismatched=True), device=dev, seed=seed)
    env.reset(init=np.zeros((n, 2)), is_mismatched=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    traj = env.rollout(T, actions=acts, shared_actions=True, want=("traj",))["traj"]
    ev[1].record()
    time_axis = np.linspace(0, (T - 1) / 30.0, T)
    estimate_velocity(traj, time_axis)            # warm
    ms = []
    for _ in range(5):
        ev[2].record()
        v, drift = estimate_velocity(traj, time_axis)
        ev[3].record()
        torch.cuda.synchronize(dev)
        ms.append(ev[2].elapsed_time(ev[3]))
    env.check_status()
    vel_ms = sorted(ms)[len(ms) // 2]
    # algorithmic bytes of the post-processing: every position read once, every velocity written once (fp64 pairs): 32 B per point
    # (the three-pass form of rounds 2-4 moved 96-128)
    gbs = 32.0 * n * T / (vel_ms * 1e-3) / 1e9
    out = {"what": "batched run_sim + velocity post-processing (8f-1, 8f-4) and batched MRExperiment export (8f-3)",
           "run_sim": {"envs": n, "steps": T, "ms": round(ev[0].elapsed_time(ev[1]), 3),
                       "env_steps_per_s": n * T / (ev[0].elapsed_time(ev[1]) * 1e-3),
                       "note": "one fused launch, shared fp64 action table, fp64 positions of every step written (16 B per env-step)"},
           "velocity": {"ms": round(vel_ms, 3), "points_per_s": n * T / (vel_ms * 1e-3),
                        "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_point": 32},
                        "mean_drift_estimate": [float(x) for x in drift.mean(dim=0).tolist()]}}
    del traj, v, drift, env
    torch.cuda.empty_cache()
    n2, T2 = 16384, 204
    env2 = MRVecEnv(n2, cfg=MRConfig(noise_var=1.0, auto_reset=True), device=dev, seed=seed)
    env2.reset()
    recorder.export_all(env2, T2)                 # warm
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    be = recorder.export_all(env2, T2)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    out["recorder_export"] = {"envs": n2, "steps": T2, "ms": round(el * 1e3, 3), "episodes_cut": int(be.ep_count.sum().item()),
                              "transitions_per_s": n2 * T2 / el}
    return out


def stats_us(ms):
    ms = sorted(ms)
    return sum(ms) / len(ms) * 1e3, ms[len(ms) // 2] * 1e3


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    ensure_built()
    if args.noise_law is None:   # the headline runs whatever law a user of the library gets without asking
        from mr_rl_amd import MRConfig as _C
        args.noise_law = _C().noise_law
        args.noise_law_is_library_default = True
    else:
        from mr_rl_amd import MRConfig as _C
        args.noise_law_is_library_default = args.noise_law == _C().noise_law
    # The contract is ONE JSON line on stdout.  Native libraries write to file descriptor 1 as well (RCCL prints a five-line
    # version banner there when its first communicator comes up, gloo announces its connections): from here on fd 1 is
    # stderr, and the JSON line goes to the saved descriptor at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from mr_rl_amd import MRConfig, MRVecEnv
    from mr_rl_amd.dist import ReturnGatherer, shard_of

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.share_gpu:
            local_rank = 0
        elif torch.cuda.device_count() <= local_rank:
            raise SystemExit("bench.py: rank %d needs cuda:%d but only %d device(s) are visible (--share-gpu only "
                             "for rehearsals)" % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        import datetime
        try:
            kw = {"timeout": datetime.timedelta(seconds=args.rendezvous_timeout)}
            if args.dist_backend == "nccl":
                kw["device_id"] = torch.device("cuda", local_rank)
            dist.init_process_group(args.dist_backend, **kw)
        except Exception as exc:  # a missing rank, a busy port, an unreachable MASTER_ADDR: say which and stop
            sys.stderr.write("bench.py: rank %d/%d could not join the %s process group at %s:%s within %.0f s: %s: %s\n"
                             % (rank, world, args.dist_backend, os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"),
                                args.rendezvous_timeout, type(exc).__name__, exc))
            sys.exit(3)
        if dist.get_world_size() != world or (args.gpus > 1 and dist.get_world_size() != args.gpus):
            sys.stderr.write("bench.py: process group has %d ranks, WORLD_SIZE=%d, --gpus %d: they must agree\n"
                             % (dist.get_world_size(), world, args.gpus))
            dist.destroy_process_group()
            sys.exit(4)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if world == 1 and os.environ.get("MRSIM_BENCH_FORCE_DIST"):
        # rehearsal of the per-episode collective's HOST cost on a one-GPU box: a one-rank RCCL group, same call path
        dist.init_process_group(args.dist_backend, init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                                **({"device_id": torch.device("cuda", 0)} if args.dist_backend == "nccl" else {}))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    pmc = under_pmc()

    n_local = args.envs_per_gpu
    total = n_local * world
    env_id0, _ = shard_of(total, rank, world)
    seed = 7
    cfg = MRConfig(noise_var=args.sigma, auto_reset=True, obs_layout=args.obs_layout, noise_math=args.noise_math,
                   seed=seed, is_mismatched=args.mismatched, noise_law=args.noise_law)
    goal_table = None
    if args.workload == "mixed":
        goal_table = mixed_goal_table(cfg, seed)
    K, W = max(args.steps, 1), max(args.warmup, 0)
    ep = cfg.max_timesteps + 1
    from mr_rl_amd._lib import EventPair
    launch = args.launch
    if pmc and launch == "graph":
        launch = "eager"
    streams = 1 if pmc else max(1, args.streams)  # under counter collection dispatches are serialised anyway

    if args.mode == "rollout":
        T = args.rollout_len
        reg = make_region(args, cfg, n_local, env_id0, world, dev, seed, streams, goal_table=goal_table, T=T)
        run, gatherer = reg.run, reg.g
        launch_desc = {"rollout_len": T, "transition_bytes_per_env_step": 33, "carry": args.carry, "streams": streams,
                       "sub_shards": [n for _, n in reg.col.shards]}
    else:
        env = MRVecEnv(n_local, cfg=cfg, device=dev, seed=seed, env_id0=env_id0, goal_table=goal_table)
        env.reset()
        gatherer = ReturnGatherer(env, world)
        G = args.graph_len
        act = torch.empty((n_local, 2), dtype=torch.float32, device=dev)
        done_steps = [0]
        n_launches = [0]

        def eager_step():
            if args.policy in ("kernel", "overlap", "episode"):  # eager remainder: same action values, drawn per step
                env.step(env.random_policy(out=act))
            else:
                env.step(None)

        graph = env.capture_steps(G, policy=args.policy) if launch == "graph" else None

        def run(nsteps, pool=None, used=None):
            left = nsteps
            while left > 0:
                chunk = min(left, ep - (done_steps[0] % ep))
                if graph is not None and chunk == G:
                    graph.replay()
                    n_launches[0] += 1
                else:
                    if graph is not None:
                        env.step_idx = 0
                    for _ in range(chunk):
                        eager_step()
                    n_launches[0] += chunk * (1 if args.policy == "fused" else 2)
                    if graph is not None:
                        env.advance_step_base(chunk)
                        env.step_idx = 0
                done_steps[0] += chunk
                left -= chunk
                if done_steps[0] % ep == 0:
                    gatherer.gather()
        launch_desc = {"policy": args.policy, "launch": launch, "graph_len": G if graph is not None else 0}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # every event pair the timed regions will use is created NOW: a thousand hipEventCreate calls between the warm-up and
    # the timed region would idle the GPU for milliseconds and the region would start on a dropped clock
    one_stream = args.mode == "rollout" and streams == 1
    ks = max(args.sustained_steps // ep, 1) * ep if (args.mode == "rollout" and args.sustained_steps > 0) else 0
    sus_pool = [EventPair() for _ in range(ks // ep + 1)] if (rank == 0 and ks) else None
    sus_used = []

    # clock settle (tools/clock_ramp_probe.py: ~35 ms of load after idle), independent of the W the caller asks for
    trace("env ready; settle phase: %d episodes" % args.settle_episodes)
    # the untimed phases keep the launch queue BOUNDED, as a loop that consumes its episodes does (every launch group of a real
    # collection loop is waited for by somebody): a device-wide synchronize every --queue-depth episodes.  Hundreds of launches queued
    # without a wait leave the HIP runtime a backlog of completed dispatches to retire, which the next launch calls then pay for
    # (25 - 240 us each instead of ~10, tools/enqueue_profile.py) -- whatever the length of the region that follows.
    qd = max(1, args.queue_depth)
    for e0 in range(0, args.settle_episodes, qd):
        run(min(qd, args.settle_episodes - e0) * ep)
        torch.cuda.synchronize(dev)
    trace("settle issued; warm-up %d steps" % W)
    run(W)
    barrier()
    trace("warm-up done; timed region")
    # ---- the contract's timed region: EXACTLY K steps between barrier + synchronize, max over ranks
    region_phases = None
    if args.mode == "rollout":
        el, launches = reg.timed(K)
        region_phases = reg.last_phases_us
        per_rank_s = reg.last_per_rank_s
    else:
        l0 = n_launches[0]
        gc.disable()
        t0 = time.perf_counter()
        run(K)
        gatherer.finish()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0    # this rank's clock; the job's time is the MAX over ranks (below)
        if world > 1:
            barrier()    # closing barrier + synchronize, outside the ranks' clocks (see RolloutRegion.timed)
        gc.enable()
        launches = n_launches[0] - l0
        per_rank_s = [el]
        if world > 1:
            cdev = dev if args.dist_backend == "nccl" else "cpu"
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            every = torch.zeros(world, dtype=torch.float64, device=cdev)
            dist.all_gather_into_tensor(every, t)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
            per_rank_s = [float(x) for x in every.tolist()]
    trace("timed region done")

    # ---- sustained legs (every rank takes part): the same workload over its own >= 10 200-step region, (a) as
    # configured (S streams), (b) one launch per episode on ONE stream with a HIP event pair on every dispatch -- the
    # kernel durations `roofline` is built from.  Each leg = PREROLL_EPISODES untimed episodes, then the timed region.
    sustained = None
    if ks:
        def leg(region, pool, used):
            region.run(PREROLL_EPISODES * ep)
            els, ls = region.timed(ks, pool, used)
            return {"value": total * ks / els, "unit": "env-steps/s", "steps": ks, "launches": ls, "streams": region.col.S,
                    "preroll_episodes": PREROLL_EPISODES, "ms_per_step": els / ks * 1e3}, els

        sustained, els_main = leg(reg, None, None)        # plain launches: the throughput
        sustained["what"] = "the headline workload over its own region, independent of --steps"
        # the same on ONE stream, one launch per episode: plain, then again with a HIP event pair on every dispatch (the durations)
        if one_stream:
            reg1, one = reg, {k: v for k, v in sustained.items() if k != "what"}
        else:
            reg1 = make_region(args, cfg, n_local, env_id0, world, dev, seed, 1, goal_table=goal_table, T=T)  # its own buffers and state
            one, _ = leg(reg1, None, None)
        ev_leg, els_one = leg(reg1, sus_pool, sus_used)
        reg1.col.check_status()
        one["what"] = ("the same on ONE stream, one launch per episode (plain launches); `with_events` = that region "
                       "again with a HIP event pair attached to every dispatch, the kernel durations behind `roofline`")
        one["with_events"] = {"value": ev_leg["value"], "ms_per_step": ev_leg["ms_per_step"]}
        sustained["one_stream"] = one
        if one_stream:      # and the two-sub-shard form (mr_rl_amd.collector: two launch chains fill each other's tails), for the record
            reg2 = make_region(args, cfg, n_local, env_id0, world, dev, seed, 2, goal_table=goal_table, T=T)
            two, _ = leg(reg2, None, None)
            reg2.col.check_status()
            two["what"] = "the same as two sub-shard launches per episode on two HIP streams (--streams 2)"
            sustained["two_streams"] = two
            del reg2
        else:
            del reg1
        trace("sustained legs done")

    (reg.col if args.mode == "rollout" else env).check_status()
    mean_ret = gatherer.last_mean()
    # events are read only now, after every timed region
    region_ms = []      # (the contract's timed region carries no events: they cost ~5 % of a one-stream region)
    sus_ms = [e.elapsed_ms() for e in sus_used]
    if sus_pool is not None:
        for e in sus_pool:
            e.close()
    if sustained is not None and sus_ms:
        avg_us, med_us = stats_us(sus_ms)
        tgt = sustained["one_stream"]["with_events"]
        tgt.update({"avg_kernel_us": round(avg_us, 3), "median_kernel_us": round(med_us, 3),
                    "kernel_time_over_wall": round(sum(sus_ms) * 1e-3 / els_one, 4),
                    "in_kernel_value": n_local * args.rollout_len / (avg_us * 1e-6)})

    mixed = None
    if args.mode == "rollout" and args.workload == "ddpg" and not args.no_mixed_set:
        mixed = measure_mixed_set(args, n_local, env_id0, world, dev, seed, streams)
        trace("mixed set done")

    other_law = None
    if args.mode == "rollout" and not args.no_other_law and args.sigma > 0 and not pmc:
        other_law = measure_other_law(args, cfg, n_local, env_id0, world, dev, seed, streams, goal_table, args.rollout_len,
                                      "per_stage" if args.noise_law == "collapsed" else "collapsed")
        trace("other-law leg done")

    # ---- roofline of the dominant kernel (rank 0): durations from HIP events attached to the dispatches
    # (hipExtLaunchKernelGGL) on the stream they run on.  Rollout mode: the dispatches of the one-stream sustained region
    # (or of the timed region when that leg is off).  Step mode (graph replays cannot carry per-dispatch events): separate
    # timed launches right after the region, same state regime.
    roof = None
    if rank == 0:
        law = "mismatched" if args.mismatched else "nominal"
        nz = "nonoise" if args.sigma == 0 else args.noise_math
        if args.mode == "rollout":
            T = args.rollout_len
            ms, timed_where = (sus_ms, "the %d dispatches of the one-stream sustained region" % len(sus_ms)) if sus_ms else \
                (region_ms, "the %d full-length dispatches of the timed region" % len(region_ms))
            if not ms:  # no sustained leg and a multi-stream / short timed region: sample afterwards instead
                ns = args.kernel_samples or 10
                env1 = MRVecEnv(n_local, cfg=cfg, device=dev, seed=seed, env_id0=env_id0, goal_table=goal_table)
                env1.reset()
                b1 = {}
                ms = [env1.rollout(T, actions=None, want=reg.col.want, out=b1, timed=True, carry=args.carry)["kernel_ms"]
                      for _ in range(ns + 20)][20:]
                timed_where = "%d launches after the timed region" % len(ms)
            avg_us, med_us = stats_us(ms)
            roof = rollout_roofline(args, n_local, T, args.noise_law, avg_us, med_us, timed_where)
        else:
            ns = args.kernel_samples or 102
            ms = [env.step_timed(env.random_policy(out=act) if args.policy != "fused" else None) for _ in range(ns)]
            avg_us, med_us = stats_us(ms)
            units = n_local
            traffic, traffic_src, _why = committed_traffic(args, n_local)
            ach = units * ALGO_BYTES_PER_ENV_STEP / (avg_us * 1e-6) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "hbm_frac": round(traffic / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                    "kernel": "mr_step_kernel<RK45,%s,%s,%s>" % (nz, law, args.obs_layout),
                    "kernel_timed_over": "%d launches right after the timed region (graph replays cannot carry events)" % len(ms),
                    "avg_kernel_us": round(avg_us, 3), "median_kernel_us": round(med_us, 3),
                    "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP, "env_steps_per_launch": units}
    trace("roofline events read")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, seed, args.cpu_seconds)
    rmse = trajectory_rmse(dev, args.carry) if rank == 0 and world == 1 else None
    trace("trajectory rmse done")
    step_path = None
    if rank == 0 and world == 1 and args.mode == "rollout" and not args.no_step_path and args.workload == "ddpg":
        if pmc:
            step_path = {"skipped": "hipGraph capture under rocprofv3 --pmc crashes the profiler (ROCm 7.2)"}
        else:
            step_path = measure_step_path(cfg, n_local, dev, seed)
        trace("step path done")

    actor_leg = None
    if rank == 0 and world == 1 and args.mode == "rollout" and args.workload == "ddpg" and not args.no_actor_leg:
        actor_leg = measure_actor_in_loop(args, n_local, dev, seed, streams)
        trace("actor-in-the-loop leg done")

    learner_leg = None
    if rank == 0 and world == 1 and args.mode == "rollout" and args.workload == "ddpg" and not args.no_learner_leg and not pmc:
        learner_leg = measure_learner(args, n_local, dev, seed, streams)
        trace("learner leg done")

    facade = None
    if rank == 0 and world == 1 and not args.no_facade_leg and not pmc:
        facade = measure_facade(seed)
        trace("facade leg done")

    streaming = None
    if rank == 0 and world == 1 and args.mode == "rollout" and args.workload == "ddpg" and not args.no_streaming_point and not pmc:
        streaming = measure_streaming_point(args, cfg, dev, seed)
        trace("streaming point done")

    consumers = None
    if rank == 0 and world == 1 and args.mode == "rollout" and args.workload == "ddpg" and not args.no_consumers_leg and not pmc:
        consumers = measure_consumers(dev, seed)
        trace("consumers leg done")

    rank_check = None
    if world > 1 or os.environ.get("MRSIM_BENCH_FORCE_DIST"):
        rank_check = verify_ranks(args, cfg, world, rank, dev, seed, n_local)
        trace("rank verification done")

    power = None
    if rank == 0 and world == 1 and args.mode == "rollout" and not args.no_power and not pmc:
        power = measure_power(reg, dev, total)
        trace("power leg done")

    if world > 1:
        dist.barrier()
    if rank == 0:
        value = total * K / max(el, 1e-12)
        config = {"workload": "BASELINE config 4: DDPG rollout, uniform random policy in the actor range drawn on "
                              "device, sigma=%g, integrator=reference(RK45), reward+done on device, auto-reset, all "
                              "transitions written to HBM" % args.sigma,
                  "trajectory_set": args.workload, "mode": args.mode, "envs_per_gpu": n_local, "total_envs": total, "obs_layout": args.obs_layout,
                  "noise_math": args.noise_math, "noise_law": args.noise_law,
                  "noise_law_is_library_default": bool(args.noise_law_is_library_default), "sigma": args.sigma, "seed": seed,
                  "is_mismatched": bool(args.mismatched),
                  "mean_episode_return": mean_ret,
                  "ranks": dist.get_world_size() if world > 1 else 1,
                  "rank_launcher": os.environ.get("MRSIM_BENCH_LAUNCHER", "external (torchrun)" if world > 1 else "none"),
                  "returns_allgather": ("%s all_gather_into_tensor (%s), the returns of every episode, %d episodes per collective"
                                        % (args.dist_backend, gatherer.mode, args.gather_interval if args.mode == "rollout" else 1))
                  if (world > 1 or gatherer._force) else "local"}
        config.update(launch_desc)
        if args.mode == "rollout" and args.carry == "f64":
            dtype = "f64 positions and carried RK45 state; f32 Box-Muller normals and stage-noise sums"
        else:
            dtype = "f64 positions; f32 Box-Muller normals, stage-noise sums and carried K0 / h_abs"
        out = {"metric": "env-steps/sec at N parallel envs; trajectory RMSE vs CPU ref", "value": value, "unit": "env-steps/s",
               "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": el / max(K, 1) * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
               "data": "synthetic", "launches": launches, "config": config, "roofline": roof, "cpu_baseline": cpu}
        # every rank's own rate over the timed region (its envs x K / its own clock between the two barriers); `value` is
        # total envs x K / the slowest rank's clock
        out["per_rank_value"] = [n_local * K / max(s_, 1e-12) for s_ in per_rank_s]
        if region_phases is not None:
            out["timed_region_phases_us"] = region_phases  # rank 0's wall time of the K-step region, by phase
        if K < 4 * ep:
            out["short_region_note"] = ("K < 4 episodes: the timed region is one or two launch groups per stream and its `value` is "
                                        "host-launch-bound (nothing is done to precondition it); `sustained` is the kernel's rate")
        if sustained is not None:
            out["sustained"] = sustained
        if rmse is not None:
            out["trajectory_rmse_vs_cpu_ref"] = rmse
        if step_path is not None:
            out["step_path"] = step_path
        if mixed is not None:
            out["mixed_trajectory_set"] = mixed
        if other_law is not None:
            out["other_noise_law"] = other_law
        if actor_leg is not None:
            out["actor_in_loop"] = actor_leg
        if learner_leg is not None:
            out["learner"] = learner_leg
        if power is not None:
            out["power"] = power
        if facade is not None:
            out["facade"] = facade
        if streaming is not None:
            out["streaming_point"] = streaming
        if consumers is not None:
            out["consumers"] = consumers
        if rank_check is not None:
            out["rank_verification"] = rank_check
        if pmc:
            out["note"] = "run under rocprofv3 counter collection: kernels are serialised, timings are not representative"
        sys.stdout.flush()
        line = (json.dumps(out) + "\n").encode()
        while line:
            line = line[os.write(json_fd, line):]
    if world > 1 or dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
