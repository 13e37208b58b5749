"""Multi-GPU layer: one process per GPU, contiguous env shards, and the ONE collective of this
path -- an all-gather of per-env episode returns at episode boundaries (RCCL over xGMI when the
process group backend is "nccl"; the same code runs over gloo in the CPU tests).

The step itself never communicates: environments are independent and the RNG is keyed by the
GLOBAL env id, so rank r simply runs envs [env_id0, env_id0 + n_local).  xGMI is
point-to-point and an episode's returns are only 4 B per env (1 MiB per rank at 262 144 envs),
i.e. latency-bound: gather once per episode (every 51 steps), never per step.
"""


def shard_of(total_envs, rank, world_size):
    """Contiguous block partition: returns (env_id0, n_local).  The first (total % world) ranks get
    one extra env."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(int(total_envs), int(world_size))
    n_local = base + (1 if rank < rem else 0)
    env_id0 = rank * base + min(rank, rem)
    return env_id0, n_local


def all_shards(total_envs, world_size):
    return [shard_of(total_envs, r, world_size) for r in range(world_size)]


def gather_returns(local_returns, out=None, group=None):
    """all-gather of per-env episode returns: [n_local] on every rank -> [world * n_local] in GLOBAL
    env order.  Requires equal shard sizes (all_gather_into_tensor); use gather_returns_ragged
    otherwise.  Asynchronous with respect to the host on CUDA(HIP) tensors: the collective is
    enqueued on the current stream."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if out is None:
            return local_returns
        out.copy_(local_returns)
        return out
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty(world * local_returns.numel(), dtype=local_returns.dtype, device=local_returns.device)
    if dist.get_backend(group) == "gloo" and local_returns.is_cuda:
        # rehearsal path only (gloo has no device all_gather_into_tensor): stage through the host
        tmp = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(tmp, local_returns.cpu().contiguous(), group=group)
        out.copy_(tmp)
        return out
    dist.all_gather_into_tensor(out, local_returns.contiguous(), group=group)
    return out


def gather_returns_ragged(local_returns, total_envs, group=None):
    """Same for unequal shards (total_envs not divisible by the world size): pads to the largest shard."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_returns
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shards = all_shards(total_envs, world)
    nmax = max(n for _, n in shards)
    pad = torch.zeros(nmax, dtype=local_returns.dtype, device=local_returns.device)
    pad[: shards[rank][1]] = local_returns
    buf = torch.empty(world * nmax, dtype=local_returns.dtype, device=local_returns.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * nmax: r * nmax + n] for r, (_, n) in enumerate(shards)])


class ReturnGatherer:
    """Episode-return reduction for a sharded MRVecEnv.

    gather() is called at an episode boundary.  It stages the [n_local] returns (a 1 MiB device copy on the
    compute stream) and starts an ASYNCHRONOUS all-gather into one of two resident [total] buffers; the
    compute stream never waits for it, so the next episode's kernels overlap the collective (xGMI is
    point-to-point and 4 B/env is latency-bound: ~tens of us that would otherwise sit on the critical path
    of a ~250 us episode).  A buffer is only reused after the collective that last used it has been
    waited on (stream-side wait, no host block).  latest() / last_mean() wait for the newest gather.
    """

    def __init__(self, env, world_size=1, group=None, source=None, release=None, force_collective=False):
        """source: optional callable returning the [n_local] returns of the episode that just ended, after making the
        current stream wait for whatever produces them (mr_rl_amd.collector.RolloutCollector: `lambda:
        collector.ready()["final_ret"]`); default env.final_ret.  release: optional callable invoked once the returns
        have been staged (the collector may then overwrite that buffer set)."""
        import torch
        self.env, self.world, self.group = env, int(world_size), group
        self._source = source if source is not None else (lambda: env.final_ret)
        self._release = release
        self._force = bool(force_collective)  # run the collective path on a one-rank group too (host-overhead rehearsal)
        n = env.num_envs
        self._stage = [torch.zeros(n, dtype=torch.float32, device=env.device) for _ in range(2)]
        self._all = [torch.zeros(self.world * n, dtype=torch.float32, device=env.device) for _ in range(2)]
        self._pending = [None, None]
        self.n_gathers = 0
        self.mode = "async"  # "async" -> "sync" if the backend rejects async_op (decided once, at the first gather)

    def _distributed(self):
        import torch.distributed as dist
        return (self.world > 1 or self._force) and dist.is_available() and dist.is_initialized()

    def gather(self):
        import torch.distributed as dist
        if not self._distributed():
            self.n_gathers += 1  # single process: env.final_ret already IS the global array (see latest())
            return
        k = self.n_gathers % 2
        if self._pending[k] is not None:
            self._pending[k].wait()  # the collective that used these buffers two episodes ago
            self._pending[k] = None
        self._stage[k].copy_(self._source())
        if self._release is not None:
            self._release()
        if dist.get_backend(self.group) == "gloo" and self._stage[k].is_cuda:
            gather_returns(self._stage[k], out=self._all[k], group=self.group)  # rehearsal path, synchronous
        elif self.mode == "async":
            try:
                self._pending[k] = dist.all_gather_into_tensor(self._all[k], self._stage[k], group=self.group,
                                                               async_op=True)
            except (RuntimeError, TypeError, NotImplementedError) as exc:
                # Every rank runs the same backend, so every rank takes this branch at the same (first) gather.
                import sys
                print(f"[mr_rl_amd.dist] async all_gather_into_tensor unavailable ({exc}); using the blocking form",
                      file=sys.stderr, flush=True)
                self.mode = "sync"
                dist.all_gather_into_tensor(self._all[k], self._stage[k], group=self.group)
        else:
            dist.all_gather_into_tensor(self._all[k], self._stage[k], group=self.group)
        self.n_gathers += 1

    def latest(self):
        """[total] returns of the most recent gather, in global env order (waits for that collective)."""
        if self.n_gathers == 0:
            return None
        if not self._distributed():
            return self._source()
        k = (self.n_gathers - 1) % 2
        if self._pending[k] is not None:
            self._pending[k].wait()
            self._pending[k] = None
        return self._all[k]

    def finish(self):
        """Wait (stream-side) for every outstanding collective; call before the final synchronize."""
        for k in range(2):
            if self._pending[k] is not None:
                self._pending[k].wait()
                self._pending[k] = None

    def last_mean(self):
        r = self.latest()
        return None if r is None else float(r.mean().item())


def verify_shards(probe, total_envs, rank, world_size, device="cpu", group=None):
    """Self-verification of a sharded run over the run's own backend.  probe(env_id0) -> 1-D float64 tensor: a deterministic
    function of the GLOBAL env ids a shard starts at (e.g. sums of final positions and returns of one episode of its first envs --
    the RNG is keyed by global env id, so any rank can recompute any shard's probe).  Every rank contributes {rank, env_id0,
    probe(env_id0)} to ONE all_gather_into_tensor; rank 0 recomputes every shard's probe itself and compares bit for bit.
    Returns on rank 0 a dict {"per_rank": [{rank, env_id0, probe, equals_rank0_recomputation}], "all_equal": bool}, None elsewhere.
    device: where the gathered tensor lives ("cpu" for gloo, the rank's cuda device for nccl = RCCL)."""
    import torch
    import torch.distributed as dist
    env_id0, _ = shard_of(total_envs, rank, world_size)
    mine_probe = probe(env_id0).to(torch.float64).reshape(-1)
    k = int(mine_probe.numel())
    mine = torch.cat([torch.tensor([float(rank), float(env_id0)], dtype=torch.float64), mine_probe.cpu()]).to(device)
    distributed = world_size > 1 and dist.is_available() and dist.is_initialized()
    if distributed:
        every = torch.zeros(world_size * (k + 2), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(every, mine, group=group)
        every = every.view(world_size, k + 2).cpu()
    else:
        every = mine.cpu().unsqueeze(0)
    if rank != 0:
        return None
    rows, ok = [], True
    for r in range(world_size):
        e0, _ = shard_of(total_envs, r, world_size)
        want = (mine_probe if r == 0 else probe(e0).to(torch.float64).reshape(-1)).cpu()
        same = bool(int(every[r, 0]) == r and int(every[r, 1]) == e0 and torch.equal(every[r, 2:], want))
        ok = ok and same
        rows.append({"rank": r, "env_id0": int(every[r, 1]), "probe": [float(x) for x in every[r, 2:]],
                     "equals_rank0_recomputation": same})
    return {"per_rank": rows, "all_equal": ok}
