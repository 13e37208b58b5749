"""RolloutCollector: the DDPG rollout workload (RL/MR_ddpg.py:270-311, collection side) for one GPU's envs, with the env
set split into S contiguous sub-shards that run as S fused-rollout launches on S HIP streams.

Why sub-shards.  At N = 262 144 one launch is exactly one resident round of waves (4 per SIMD).  The waves of a SIMD do
not finish together (the oldest run ahead: DESIGN.md section 7b), so every launch ends with a tail in which SIMDs hold one
or two waves, and on ONE stream the next launch cannot start before the last wave of the previous one has left.  Two
half-size launch chains on two streams depend only on their own halves: the head of one chain's next episode fills the
slots the other chain's tail frees.  No env ever waits for an env of another sub-shard -- the path has no data
dependence between envs -- and results are bit-identical to the single launch (same global env ids, same
step indices; tests/test_gpu_round2.py).

Data flow.  Transitions go to `depth` (default 2) rotating sets of [T, N, ...] buffers, the sub-shards writing their
columns (MrsimRolloutIO.row_stride).  collect() only enqueues; ready(k) makes the CURRENT stream wait for episode k's
launches and hands out its buffers; release(k) (called by the consumer once it is done with them, on the current
stream) lets the sub-shard streams overwrite that set `depth` episodes later.  A learner therefore reads episode k while
the envs produce episode k + 1.

Episode returns / lengths go to row (k mod E) of one of two [E, N] blocks (E = returns_interval): BlockReturnGatherer
all-gathers a whole block -- the returns of E consecutive episodes of every env -- with ONE collective, so the per-episode
host work of the collection loop is just the S launches.  (The reference logs once per 100 episodes, RL/MR_ddpg.py:317-320;
E = 1 gives one collective per episode, which at ~115 us per episode is bound by the ~100 us of Python / RCCL enqueue it
costs, not by the GPU.)
"""
from . import _lib  # noqa: F401  (the ctypes binding must be loadable: no CPU fallback)
from .config import MRConfig
from .dist import all_shards
from .vec_env import MRVecEnv


_STREAMS = {}


def _sub_shard_streams(dev, count):
    """The HIP streams the sub-shard launches go to: ONE set per device for the whole process, shared by every collector.
    torch hands out pool streams round robin and not every pair of them runs its kernels side by side -- measured: the second
    collector created in a process (pool streams 2 and 3) had its two sub-shard launches run one after the other, 12 % off the
    collection rate, every other collector overlapped them (profiles/r03/NOTES.md).  Collectors are used one at a time; two used
    at once merely share the streams' order."""
    import torch
    key = (str(dev), int(count))
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(count)]
    return _STREAMS[key]


class RolloutCollector:
    WANT = ("obs", "rew", "done", "actions")

    def __init__(self, num_envs, cfg=None, device="cuda", seed=None, env_id0=0, goal_table=None, streams=2, T=None,
                 want=WANT, carry="f64", depth=2, returns_interval=1, policy=None, stream_list=None):
        """stream_list: caller-owned torch streams for the sub-shards (len = number of sub-shards) instead of the process-wide
        shared set -- for two collectors that must run side by side (an evaluation collector beside a training one); the default
        shared set orders every collector of a device on the same streams (collectors are then used one at a time)."""
        import torch
        self.env = MRVecEnv(num_envs, cfg=cfg if cfg is not None else MRConfig(auto_reset=True), device=device, seed=seed,
                            env_id0=env_id0, goal_table=goal_table)
        env = self.env
        self.N, self.S, self.depth, self.carry = env.num_envs, max(1, int(streams)), int(depth), carry
        self.T = int(T) if T is not None else env.cfg.max_timesteps + 1   # one episode per launch by default
        self.E = max(1, int(returns_interval))
        self.want = tuple(want)
        self.policy = policy   # None: the uniform exploration policy drawn in-kernel; a DeviceActor: actor + OU noise in-kernel
        dev = env.device
        if policy is not None:
            policy.ou_tensor(env.num_envs)   # created (and its zero fill awaited) HERE, before any sub-shard stream can read it
        self.shards = [(a, n) for a, n in all_shards(self.N, self.S) if n > 0]
        if stream_list is not None:
            if len(stream_list) != len(self.shards):
                raise ValueError("stream_list must hold one stream per sub-shard (%d)" % len(self.shards))
            self.streams = list(stream_list)
        else:
            self.streams = _sub_shard_streams(dev, len(self.shards))
        T_, N = self.T, self.N
        soa = env._soa
        shapes = {"traj": ((T_, N, 2), torch.float64), "state_prime": ((T_, N, 2), torch.float32),
                  "obs": ((T_, 5, N) if soa else (T_, N, 5), torch.float32), "rew": ((T_, N), torch.float32),
                  "done": ((T_, N), torch.uint8), "actions": ((T_, N, 2), torch.float32)}
        self.sets = [{k: torch.empty(shapes[k][0], dtype=shapes[k][1], device=dev) for k in self.want}
                     for _ in range(self.depth)]
        self.ret_blocks = [torch.zeros((self.E, N), dtype=torch.float32, device=dev) for _ in range(2)]
        self.len_blocks = [torch.zeros((self.E, N), dtype=torch.int32, device=dev) for _ in range(2)]
        self._done_ev = [[torch.cuda.Event() for _ in self.shards] for _ in range(self.depth)]
        self._free_ev = [None] * self.depth     # recorded by release(): set b may be overwritten after it
        self._ret_free_ev = [None, None]        # recorded by free_returns_block(): block b may be overwritten after it
        self._start_ev = torch.cuda.Event()
        self._prepared = {}                     # (shard, set, steps, parameter block) -> prepared launch
        self._prepared_version = env._params_version
        self.episodes = 0
        self._synced_streams = False

    # ------------------------------------------------------------------------------------------------------------
    def reset(self, **kw):
        """MR_Env.reset of every env (current stream); the sub-shard streams start behind it.  Launches still in flight on
        the sub-shard streams read and write the state the reset kernel is about to overwrite: the current stream waits for
        them first (join), so reset() may follow collect() directly."""
        import torch
        self.join()
        obs = self.env.reset(**kw)
        self._start_ev.record(torch.cuda.current_stream(self.env.device))
        for st in self.streams:
            st.wait_event(self._start_ev)
        self._synced_streams = True
        return obs

    def _where(self, k):
        return k % self.depth, (k // self.E) % 2, k % self.E   # buffer set, returns block, row of the block

    def collect(self, events=None, steps=None, after=None):
        """Enqueue the next launch group: `steps` (default T = one episode) steps per sub-shard, each sub-shard on its
        own stream, into buffer set episodes % depth (rows [0, steps)).  Nothing is synchronised.  events: optional
        list of _lib.EventPair, one per sub-shard.  after: a torch stream whose work enqueued so far (e.g. a parameter
        upload of the in-kernel policy) the launches of this group must wait for.  With a DeviceActor of several parameter
        blocks, launch group k reads block k % slots."""
        import torch
        env, k = self.env, self.episodes
        after_ev = None
        if after is not None:
            after_ev = torch.cuda.Event()
            after_ev.record(after)
        assert self._synced_streams, "call reset() first"
        T = self.T if steps is None else int(steps)
        assert 1 <= T <= self.T
        if env._params_version != self._prepared_version:   # reset() kwargs / set_init_space replaced the parameter block
            self._prepared.clear()
            self._prepared_version = env._params_version
        b, blk, row = self._where(k)
        free_set, free_blk = self._free_ev[b], (self._ret_free_ev[blk] if row == 0 else None)
        # An event that has already completed orders nothing: asking (hipEventQuery, ~1 us) is cheaper than making every sub-shard
        # stream wait for it (hipStreamWaitEvent, ~10 us of host time each -- a consumer that calls ready() / release() per episode
        # otherwise pays more host time in waits than in launches: 118 instead of 144 G env-steps/s at 90 us per episode)
        if free_set is not None and free_set.query():
            free_set = self._free_ev[b] = None
        if free_blk is not None and free_blk.query():
            free_blk = None
        for s, ((first, n), st) in enumerate(zip(self.shards, self.streams)):
            if free_set is not None:
                st.wait_event(free_set)         # the consumer has released this buffer set
            if free_blk is not None:
                st.wait_event(free_blk)         # the collective that read this returns block has finished
            if after_ev is not None:
                st.wait_event(after_ev)
            launch = self._launch_for(s, b, blk, row, T, k)
            launch(env.step_idx, events=None if events is None else events[s])
            self._done_ev[b][s].record(st)
        if row == 0:
            self._ret_free_ev[blk] = None
        env.step_idx += T
        self.episodes += 1
        return k

    def _launch_for(self, s, b, blk, row, T, k=0):
        """The prepared launch (argument block built once, vec_env.launch_rollout(prepare_only=True)) of sub-shard s into
        buffer set b / returns row (blk, row) for T steps (launch group k: which parameter block an in-kernel policy reads)."""
        slot = 0 if self.policy is None else k % len(self.policy.blobs)
        key = (s, b, T, slot)
        launch = self._prepared.get(key)
        first = self.shards[s][0]
        if launch is None:
            bufs, (first, n) = self.sets[b], self.shards[s]
            launch = self._prepared[key] = self.env.launch_rollout(
                T, first, n, traj=bufs.get("traj"), sp_T=bufs.get("state_prime"), obs_T=bufs.get("obs"),
                rew_T=bufs.get("rew"), done_T=bufs.get("done"), acts_T=bufs.get("actions"),
                final_ret=self.ret_blocks[blk][row], final_len=self.len_blocks[blk][row], carry=self.carry,
                stream=self.streams[s], prepare_only=True, actor=self.policy, actor_slot=slot)
        # the returns row rotates every launch group: re-pointed in the prepared argument block (one block per buffer set, not per row)
        launch.io.final_ret = self.ret_blocks[blk][row].data_ptr() + 4 * first
        launch.io.final_len = self.len_blocks[blk][row].data_ptr() + 4 * first
        return launch

    def prime(self, schedule):
        """Build the argument blocks of the next len(schedule) collect(steps=schedule[i]) calls now (host work only, nothing
        is enqueued): a steady loop sees each (buffer set, returns row, length) combination again and again and pays this
        once; a short or irregular schedule can pay it ahead of time."""
        if self.env._params_version != self._prepared_version:
            self._prepared.clear()
            self._prepared_version = self.env._params_version
        for i, T in enumerate(schedule):
            b, blk, row = self._where(self.episodes + i)
            for s in range(len(self.shards)):
                self._launch_for(s, b, blk, row, int(T), self.episodes + i)

    def wait_episode(self, k=None):
        """Make the current stream wait for the launches of episode k (default: the newest)."""
        import torch
        k = self.episodes - 1 if k is None else k
        assert 0 <= k < self.episodes and k >= self.episodes - self.depth, "that episode's events were reused"
        cur = torch.cuda.current_stream(self.env.device)
        for ev in self._done_ev[k % self.depth]:
            if not ev.query():                    # (a consumer that lags an episode behind has nothing to wait for)
                cur.wait_event(ev)
        return k

    def ready(self, k=None):
        """Make the current stream wait for episode k (default: the newest) and return its buffers: the [T, N, ...]
        transition tensors plus "final_ret" / "final_len" [N] (returns / lengths of the episodes that ended in that
        launch).  The transition buffers stay valid until release(k) + `depth` further collect()s."""
        import torch
        k = self.wait_episode(k)
        b, blk, row = self._where(k)
        out = dict(self.sets[b])
        if "obs" in out and self.env._soa:
            out["obs"] = out["obs"].transpose(1, 2)
        if "done" in out:
            out["done"] = out["done"].view(torch.bool)
        out["final_ret"], out["final_len"] = self.ret_blocks[blk][row], self.len_blocks[blk][row]
        return out

    def release(self, k=None):
        """The consumer is done with episode k's buffers (as far as the work it has enqueued on the current stream)."""
        import torch
        k = self.episodes - 1 if k is None else k
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.env.device))
        self._free_ev[k % self.depth] = ev

    def free_returns_block(self, blk):
        """Whatever the current stream has enqueued so far is the last reader of returns block `blk`."""
        import torch
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.env.device))
        self._ret_free_ev[blk] = ev

    def join(self):
        """Current stream waits for everything enqueued so far on the sub-shard streams (env state included)."""
        import torch
        cur = torch.cuda.current_stream(self.env.device)
        for st in self.streams:
            ev = torch.cuda.Event()
            ev.record(st)
            cur.wait_event(ev)

    def check_status(self):
        self.join()
        return self.env.check_status()


class BlockReturnGatherer:
    """Episode-return reduction for a RolloutCollector: gather() is called after every collect() and, every E =
    collector.E launch groups, starts ONE asynchronous all_gather_into_tensor of the [E, n_local] block of returns those
    launch groups wrote (RCCL over xGMI when the group's backend is "nccl").  Rows are indexed by the COLLECTOR's launch
    group, whatever its length: with one episode per launch (the default T) a row is an episode; a schedule that cuts an
    episode into several launch groups (bench.py --steps 20) spends a row per group, and a row only holds the returns of
    the envs whose episode ended in that group.  The comm stream waits for the block's last
    episode through the collector's events; the sub-shard chains never wait for the collective -- only, one block
    later, for the event that says the collective has read the block they are about to overwrite.  Result layout:
    [world, E, n_local] = returns of E consecutive episodes in global env order.  Single process (and not forced):
    no collective at all."""

    def __init__(self, col, world_size=1, group=None, force_collective=False):
        import torch
        self.col, self.world, self.group = col, int(world_size), group
        self._force = bool(force_collective)
        self._all = [torch.zeros((self.world, col.E, col.N), dtype=torch.float32, device=col.env.device) for _ in range(2)]
        self._pending = [None, None]
        self._last = None          # block index of the newest gathered (or, single process, completed) block
        self._k_seen = -1          # newest launch group gather() has been called for
        self.n_gathers = 0         # launch groups seen
        self.n_collectives = 0
        self.mode = "async"

    def _distributed(self):
        import torch.distributed as dist
        return (self.world > 1 or self._force) and dist.is_available() and dist.is_initialized()

    def _retire(self, blk):
        """stream-side wait for the collective that read block blk, then let the collector overwrite the block"""
        if self._pending[blk] is not None:
            self._pending[blk].wait()
            self._pending[blk] = None
            self.col.free_returns_block(blk)

    def gather(self):
        import torch.distributed as dist
        col = self.col
        k = col.episodes - 1           # the launch group collect() has just enqueued: its index is what rows / events go by
        if k < 0 or k == self._k_seen:
            return
        if k != self._k_seen + 1:
            raise RuntimeError(f"BlockReturnGatherer.gather() must follow every collect(): launch groups "
                               f"{self._k_seen + 1}..{k - 1} were skipped")
        self._k_seen = k
        self.n_gathers += 1
        if (k + 1) % col.E:
            return
        blk = (k // col.E) % 2
        self._last = blk
        if not self._distributed():
            return
        self._retire(1 - blk)          # the chains are about to write the other block again
        self._retire(blk)              # (only if this block's previous collective is still un-waited: E = 1 corner)
        col.wait_episode(k)            # the block's last episode; the chains run their episodes in order
        src, dst = col.ret_blocks[blk].view(-1), self._all[blk].view(-1)
        if dist.get_backend(self.group) == "gloo" and src.is_cuda:
            from .dist import gather_returns
            gather_returns(src, out=dst, group=self.group)   # rehearsal path (gloo has no device collective), synchronous
            col.free_returns_block(blk)
        elif self.mode == "async":
            try:
                self._pending[blk] = dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True)
            except (RuntimeError, TypeError, NotImplementedError) as exc:
                import sys
                print(f"[mr_rl_amd.collector] async all_gather_into_tensor unavailable ({exc}); using the blocking form",
                      file=sys.stderr, flush=True)
                self.mode = "sync"
                dist.all_gather_into_tensor(dst, src, group=self.group)
                col.free_returns_block(blk)
        else:
            dist.all_gather_into_tensor(dst, src, group=self.group)
            col.free_returns_block(blk)
        self.n_collectives += 1

    def latest(self):
        """[world, E, n_local] returns of the newest complete block (waits, stream-side, for its collective); None before
        the first block is complete."""
        if self._last is None:
            return None
        blk = self._last
        if not self._distributed():
            self.col.wait_episode()
            return self.col.ret_blocks[blk].unsqueeze(0)
        if self._pending[blk] is not None:
            self._pending[blk].wait()
            self._pending[blk] = None
            self.col.free_returns_block(blk)
        return self._all[blk]

    def finish(self):
        """Wait (stream-side) for every outstanding collective; call before the final synchronize."""
        for blk in range(2):
            self._retire(blk)

    def last_mean(self):
        r = self.latest()
        if r is None:   # fewer than E episodes so far: the newest episode's local returns
            if self.col.episodes == 0:
                return None
            return float(self.col.ready()["final_ret"].mean().item())
        return float(r.mean().item())
