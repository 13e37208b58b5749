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

With --gpus N>1 every rank owns a contiguous shard of N x 262144 envs (weak scaling), there is no
data-path collective, and episode returns are all-gathered over RCCL at episode boundaries (every
51 steps) inside the timed region.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel of the chosen mode) and
`cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 97   # fp64 positions -> reads 40 + writes 57 per env-step
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=51000)   # 1000 episodes
    ap.add_argument("--warmup", type=int, default=10200)   # 200 episodes = 35 ms: the GPU clocks have settled (tools/clock_ramp_probe.py)
    ap.add_argument("--envs-per-gpu", type=int, default=262144)
    ap.add_argument("--mode", choices=["rollout", "step"], default="rollout")
    ap.add_argument("--rollout-len", type=int, default=51)
    ap.add_argument("--policy", choices=["kernel", "fused"], default="kernel", help="step mode only")
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph", help="step mode only")
    ap.add_argument("--graph-len", type=int, default=51)
    ap.add_argument("--obs-layout", choices=["aos", "soa"], default="aos")
    ap.add_argument("--noise-math", choices=["fast", "spec"], default="fast")
    ap.add_argument("--sigma", type=float, default=1.0)
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
    ap.add_argument("--no-step-path", action="store_true", help="skip the extra one-launch-per-step measurement")
    ap.add_argument("--no-mixed-set", action="store_true", help="skip the extra mixed-trajectory-set measurement")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real runs; gloo only to rehearse")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    return ap.parse_args()


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
    return {"value": n * steps / el, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{n} envs x {steps} steps ({el:.1f} s) of the same workload, oracle/mrsim_oracle.c "
                      f"with OpenMP over {threads} host threads"}


def committed_traffic(args, n_local):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC passes
    (profiles/rNN/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this
    same command and corrected as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from
    inside the timed process, so this is null unless the run matches the profiled configuration."""
    import glob
    if not (args.sigma == 1.0 and args.noise_math == "fast" and n_local == 262144 and args.obs_layout == "aos"
            and args.workload == "ddpg"):
        return None, None
    key = "mr_rollout_kernel" if args.mode == "rollout" else "mr_step_kernel"
    if args.mode == "rollout" and args.rollout_len != 51:
        return None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")), reverse=True):
        try:
            for name, k in json.load(open(f))["kernels"].items():
                if name.startswith(key):
                    return int(k["hbm_bytes_per_launch"]), os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def measure_step_path(cfg, n_local, dev, seed, steps=10200, samples=204):
    """Secondary figure reported beside the headline: the same workload driven through the drop-in gym loop,
    one launch per MR_Env.step() ([policy kernel -> actions in HBM] + [step kernel], hipGraph of 51 steps)."""
    import torch
    from mr_rl_amd import MRVecEnv
    env = MRVecEnv(n_local, cfg=cfg, device=dev, seed=seed)
    env.reset()
    ep = cfg.max_timesteps + 1
    graph = env.capture_steps(ep, policy="kernel")
    for _ in range(40):  # ~20 ms: lets the GPU clocks settle
        graph.replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps // ep):
        graph.replay()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    k = (steps // ep) * ep
    act = torch.empty((n_local, 2), dtype=torch.float32, device=dev)
    ms = sorted(env.step_timed(env.random_policy(out=act)) for _ in range(samples))
    avg_ms = sum(ms) / len(ms)
    ach = n_local * ALGO_BYTES_PER_ENV_STEP / (avg_ms * 1e-3) / 1e9
    env.check_status()
    return {"mode": "step (one launch per env.step, hipGraph of 51 steps, policy kernel + step kernel)",
            "value": n_local * k / el, "unit": "env-steps/s", "steps": k, "ms_per_step": el / k * 1e3,
            "kernel": "mr_step_kernel", "avg_kernel_us": round(avg_ms * 1e3, 3),
            "roofline_frac": round(ach / HBM_PEAK_GBS, 4)}


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


def measure_mixed_set(args, n_local, env_id0, world, dev, seed, barrier, steps=10200):
    """Secondary figure (every rank takes part, same barrier / max-over-ranks protocol as the headline): the fused
    rollout on BASELINE config 5's mixed straight-line / figure-eight / random-waypoint trajectory set with the
    goal reward, returns all-gathered once per episode."""
    import torch
    import torch.distributed as dist
    from mr_rl_amd import MRConfig, MRVecEnv
    from mr_rl_amd.dist import ReturnGatherer
    cfg = MRConfig(noise_var=args.sigma, auto_reset=True, obs_layout=args.obs_layout, noise_math=args.noise_math,
                   seed=seed, is_mismatched=args.mismatched)
    tab = mixed_goal_table(cfg, seed)
    env = MRVecEnv(n_local, cfg=cfg, device=dev, seed=seed, env_id0=env_id0, goal_table=tab)
    env.reset()
    g = ReturnGatherer(env, world)
    ep = cfg.max_timesteps + 1
    bufs = {}
    want = ("obs", "rew", "done", "actions")

    def run(n_ep):
        for _ in range(n_ep):
            env.rollout(ep, actions=None, want=want, out=bufs)
            g.gather()

    run(20)
    barrier()
    t0 = time.perf_counter()
    run(steps // ep)
    g.finish()
    torch.cuda.synchronize(dev)
    barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    k = (steps // ep) * ep
    env.check_status()
    return {"workload": "BASELINE config 5 trajectory set: env_id mod 3 -> straight line / figure eight / random "
                        "waypoints (goal table), goal reward, same policy / noise / outputs as the headline",
            "value": n_local * world * k / el, "unit": "env-steps/s", "steps": k, "ms_per_step": el / k * 1e3,
            "mean_episode_return": g.last_mean()}


def trajectory_rmse(dev):
    """Second half of BASELINE's metric: trajectory RMSE vs the CPU reference, on the committed golden trajectories
    the reference itself produced (tests/golden/ref_sim.npz, sigma = 0, 1000-2000 steps each), through the same
    fused kernel the timed region runs."""
    import numpy as np
    from mr_rl_amd import MRConfig, MRVecEnv
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_sim.npz"))
    names = sorted({k.split("/")[0] for k in g.files})
    worst, per = 0.0, {}
    for name in names:
        G = {k.split("/")[1]: g[k] for k in g.files if k.startswith(name + "/")}
        env = MRVecEnv(64, cfg=MRConfig(noise_var=0.0, a0=float(G["a0"]), is_mismatched=bool(G["mismatched"])), device=dev)
        env._prev_mismatched = bool(G["mismatch_at_reset"])
        env.reset(init=np.tile(G["init"][None, :], (64, 1)), is_mismatched=bool(G["mismatched"]))
        traj = env.rollout(len(G["actions"]), actions=G["actions"].astype(np.float32), shared_actions=True,
                           want=("traj",))["traj"][:, 0, :].cpu().numpy()
        per[name] = float(np.sqrt(np.mean(np.sum((traj - G["pos"]) ** 2, axis=1))))
        worst = max(worst, per[name])
    return {"value": worst, "unit": "position units (max over %d golden trajectories)" % len(names), "target": 1e-5,
            "fixtures": "tests/golden/ref_sim.npz"}


def main():
    args = parse()
    lib_so = os.path.join(ROOT, "mr_rl_amd", "libmrsim.so")
    if not os.path.exists(lib_so):  # git-ignored build product missing in this checkout: build it (no fallback path)
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            t_wait = time.time()
            while not os.path.exists(lib_so) and time.time() - t_wait < 600:
                time.sleep(1.0)
            time.sleep(2.0)
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
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    n_local = args.envs_per_gpu
    total = n_local * world
    env_id0, _ = shard_of(total, rank, world)
    seed = 7
    cfg = MRConfig(noise_var=args.sigma, auto_reset=True, obs_layout=args.obs_layout, noise_math=args.noise_math,
                   seed=seed, is_mismatched=args.mismatched)
    goal_table = None
    if args.workload == "mixed":
        goal_table = mixed_goal_table(cfg, seed)
    env = MRVecEnv(n_local, cfg=cfg, device=dev, seed=seed, env_id0=env_id0, goal_table=goal_table)
    env.reset()
    gatherer = ReturnGatherer(env, world)
    K, W = max(args.steps, 1), max(args.warmup, 0)
    ep = cfg.max_timesteps + 1
    done_steps = [0]
    WANT = ("obs", "rew", "done", "actions")

    if args.mode == "rollout":
        T = args.rollout_len
        bufs = {}
        # rank 0 attaches a pair of HIP events to every full-length dispatch of the TIMED region (non-blocking,
        # hipExtLaunchKernelGGL on the launch stream); they are read after the region for roofline.achieved
        from mr_rl_amd._lib import EventPair
        ev_pool = [EventPair() for _ in range(min(K // T + 1, 4096))] if rank == 0 else []
        ev_used = []
        ev_on = [False]

        def run(nsteps):
            """exactly nsteps env steps in launches of <= T, cut at episode boundaries"""
            left = nsteps
            while left > 0:
                chunk = min(left, T, ep - (done_steps[0] % ep))
                ev = None
                if ev_on[0] and chunk == T and len(ev_used) < len(ev_pool):
                    ev = ev_pool[len(ev_used)]
                    ev_used.append(ev)
                env.rollout(chunk, actions=None, want=WANT, out=bufs if chunk == T else None, events=ev)
                done_steps[0] += chunk
                left -= chunk
                if done_steps[0] % ep == 0:
                    gatherer.gather()  # RCCL all-gather of this episode's returns
        launch_desc = {"rollout_len": T, "transition_bytes_per_env_step": 33}
    else:
        G = args.graph_len
        act = torch.empty((n_local, 2), dtype=torch.float32, device=dev)

        def eager_step():
            if args.policy == "kernel":
                env.step(env.random_policy(out=act))
            else:
                env.step(None)

        graph = env.capture_steps(G, policy=args.policy) if args.launch == "graph" else None

        def run(nsteps):
            left = nsteps
            while left > 0:
                chunk = min(left, ep - (done_steps[0] % ep))
                if graph is not None and chunk == G:
                    graph.replay()
                else:
                    if graph is not None:
                        env.step_idx = 0
                    for _ in range(chunk):
                        eager_step()
                    if graph is not None:
                        env.advance_step_base(chunk)
                        env.step_idx = 0
                done_steps[0] += chunk
                left -= chunk
                if done_steps[0] % ep == 0:
                    gatherer.gather()
        launch_desc = {"policy": args.policy, "launch": args.launch, "graph_len": G if graph is not None else 0}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # clock settle (tools/clock_ramp_probe.py: ~35 ms of load after idle), independent of the W the caller asks for
    run(args.settle_episodes * ep)
    run(W)
    barrier()
    if args.mode == "rollout":
        ev_on[0] = True
    t0 = time.perf_counter()
    run(K)
    gatherer.finish()  # outstanding async all-gathers belong to the timed region
    torch.cuda.synchronize(dev)
    barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    env.check_status()
    mean_ret = gatherer.last_mean()
    mixed = None
    if args.mode == "rollout" and args.workload == "ddpg" and not args.no_mixed_set:
        mixed = measure_mixed_set(args, n_local, env_id0, world, dev, seed, barrier)

    # ---- duration of the dominant kernel, HIP events attached to the dispatch (hipExtLaunchKernelGGL) on the
    # stream it runs on.  Rollout mode: the dispatches OF the timed region.  Step mode (the timed region replays
    # hipGraphs, which cannot carry per-dispatch events): separate timed launches right after it, same state regime.
    roof = None
    if rank == 0:
        if args.mode == "rollout":
            T = args.rollout_len
            ev_on[0] = False
            ms = sorted(e.elapsed_ms() for e in ev_used)
            timed_where = "the %d full-length dispatches of the timed region" % len(ms)
            if not ms:  # fewer than T steps were timed: sample afterwards instead
                ns = args.kernel_samples or 10
                ms = sorted(env.rollout(T, actions=None, want=WANT, out=bufs, timed=True)["kernel_ms"] for _ in range(ns))
                timed_where = "%d launches right after the timed region" % len(ms)
            for e in ev_pool:
                e.close()
            units = n_local * T
            law = "mismatched" if args.mismatched else "nominal"
            kname = "mr_rollout_kernel<RK45,%s,%s>" % ("nonoise" if args.sigma == 0 else args.noise_math, law)
        else:
            ns = args.kernel_samples or 102
            ms = sorted(env.step_timed(env.random_policy(out=act) if args.policy == "kernel" else None)
                        for _ in range(ns))
            units = n_local
            timed_where = "%d launches right after the timed region (graph replays cannot carry events)" % len(ms)
            law = "mismatched" if args.mismatched else "nominal"
            kname = "mr_step_kernel<RK45,%s,%s,%s>" % ("nonoise" if args.sigma == 0 else args.noise_math, law,
                                                            args.obs_layout)
        avg_ms = sum(ms) / len(ms)
        ach = units * ALGO_BYTES_PER_ENV_STEP / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = committed_traffic(args, n_local)
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": kname, "kernel_timed_over": timed_where,
                "avg_kernel_us": round(avg_ms * 1e3, 3), "median_kernel_us": round(ms[len(ms) // 2] * 1e3, 3),
                "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP, "env_steps_per_launch": units}
        if args.mode == "rollout":
            roof["note"] = ("achieved counts SURVEY 8(d)'s ALGORITHMIC 97 B per env-step, which assume the env state "
                            "round-trips HBM every step; the fused rollout keeps it in registers and moves only "
                            "`traffic` bytes (about 0.36 x algorithmic), so achieved can exceed the HBM peak: the "
                            "kernel is VALU-bound (DESIGN.md section 7)")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, seed, args.cpu_seconds)
    rmse = trajectory_rmse(dev) if rank == 0 and world == 1 else None
    step_path = None
    if rank == 0 and world == 1 and args.mode == "rollout" and not args.no_step_path and args.workload == "ddpg":
        step_path = measure_step_path(cfg, n_local, dev, seed)

    if world > 1:
        dist.barrier()
    if rank == 0:
        value = total * K / max(el, 1e-12)
        config = {"workload": "BASELINE config 4: DDPG rollout, uniform random policy in the actor range drawn on "
                              "device, sigma=%g, integrator=reference(RK45), reward+done on device, auto-reset, all "
                              "transitions written to HBM" % args.sigma,
                  "trajectory_set": args.workload, "mode": args.mode, "envs_per_gpu": n_local, "total_envs": total, "obs_layout": args.obs_layout,
                  "noise_math": args.noise_math, "sigma": args.sigma, "seed": seed,
                  "is_mismatched": bool(args.mismatched),
                  "mean_episode_return": mean_ret,
                  "returns_allgather": ("%s all_gather_into_tensor (%s) every 51 steps"
                                        % (args.dist_backend, gatherer.mode)) if world > 1 else "local"}
        config.update(launch_desc)
        out = {"metric": "env-steps/sec at N parallel envs; trajectory RMSE vs CPU ref", "value": value, "unit": "env-steps/s",
               "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": el / max(K, 1) * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic", "config": config, "roofline": roof, "cpu_baseline": cpu}
        if rmse is not None:
            out["trajectory_rmse_vs_cpu_ref"] = rmse
        if step_path is not None:
            out["step_path"] = step_path
        if mixed is not None:
            out["mixed_trajectory_set"] = mixed
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
