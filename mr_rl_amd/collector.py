"""RolloutCollector: the DDPG rollout workload (RL/MR_ddpg.py:270-311, collection side) for one GPU's envs, with the env
set split into S contiguous sub-shards that run as S fused-rollout launches on S HIP streams.

Why sub-shards.  At N = 262 144 one launch is exactly one resident round of waves (4 per SIMD).  The waves of a SIMD do
not finish together (the oldest run ahead: DESIGN.md section 7), so every launch ends with a tail in which SIMDs hold one
or two waves, and on ONE stream the next launch cannot start before the last wave of the previous one has left.  Two
half-size launch chains on two streams depend only on their own halves: the head of one chain's next episode fills the
slots the other chain's tail frees.  No env ever waits for an env of another sub-shard -- the path has no data
dependence between envs -- and results are bit-identical to the single launch (same global env ids, same
step indices; tests/test_gpu_round2.py).

Data flow.  Transitions go to `depth` (default 2) rotating sets of [T, N, ...] buffers, the sub-shards writing their
columns (MrsimRolloutIO.row_stride); episode returns / lengths to rotating [N] buffers.  collect() only enqueues;
ready(k) makes the CURRENT stream wait for episode k's launches and hands out its buffers; release(k) (called by the
consumer once it is done with them, on the current stream) lets the sub-shard streams overwrite that set `depth`
episodes later.  A learner therefore reads episode k while the envs produce episode k + 1.
"""
import numpy as np

from . import _lib
from .config import MRConfig
from .dist import all_shards
from .vec_env import MRVecEnv


class RolloutCollector:
    WANT = ("obs", "rew", "done", "actions")

    def __init__(self, num_envs, cfg=None, device="cuda", seed=None, env_id0=0, goal_table=None, streams=2, T=None,
                 want=WANT, carry="f64", depth=2):
        import torch
        self.env = MRVecEnv(num_envs, cfg=cfg if cfg is not None else MRConfig(auto_reset=True), device=device, seed=seed,
                            env_id0=env_id0, goal_table=goal_table)
        env = self.env
        self.N, self.S, self.depth, self.carry = env.num_envs, max(1, int(streams)), int(depth), carry
        self.T = int(T) if T is not None else env.cfg.max_timesteps + 1   # one episode per launch by default
        self.want = tuple(want)
        dev = env.device
        self.shards = [(a, n) for a, n in all_shards(self.N, self.S) if n > 0]
        # a sub-shard should start on a 256-env block boundary only for tidiness; any split is correct
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.shards]
        T_, N = self.T, self.N
        soa = env._soa
        shapes = {"traj": ((T_, N, 2), torch.float64), "state_prime": ((T_, N, 2), torch.float32),
                  "obs": ((T_, 5, N) if soa else (T_, N, 5), torch.float32), "rew": ((T_, N), torch.float32),
                  "done": ((T_, N), torch.uint8), "actions": ((T_, N, 2), torch.float32)}
        self.sets = [{k: torch.empty(shapes[k][0], dtype=shapes[k][1], device=dev) for k in self.want}
                     for _ in range(self.depth)]
        self.final_ret = [torch.zeros(N, dtype=torch.float32, device=dev) for _ in range(self.depth)]
        self.final_len = [torch.zeros(N, dtype=torch.int32, device=dev) for _ in range(self.depth)]
        self._done_ev = [[torch.cuda.Event() for _ in self.shards] for _ in range(self.depth)]
        self._free_ev = [None] * self.depth     # recorded by release(): set b may be overwritten after it
        self._start_ev = torch.cuda.Event()
        self.episodes = 0
        self._synced_streams = False

    # ------------------------------------------------------------------------------------------------------------
    def reset(self, **kw):
        """MR_Env.reset of every env (current stream); the sub-shard streams start behind it."""
        import torch
        obs = self.env.reset(**kw)
        self._start_ev.record(torch.cuda.current_stream(self.env.device))
        for st in self.streams:
            st.wait_event(self._start_ev)
        self._synced_streams = True
        return obs

    def collect(self, events=None, steps=None):
        """Enqueue the next launch group: `steps` (default T = one episode) steps per sub-shard, each sub-shard on its
        own stream, into buffer set episodes % depth (rows [0, steps)).  Nothing is synchronised.  events: optional
        list of _lib.EventPair, one per sub-shard."""
        env, k = self.env, self.episodes
        assert self._synced_streams, "call reset() first"
        T = self.T if steps is None else int(steps)
        assert 1 <= T <= self.T
        b = k % self.depth
        bufs = self.sets[b]
        for s, ((first, n), st) in enumerate(zip(self.shards, self.streams)):
            if self._free_ev[b] is not None:
                st.wait_event(self._free_ev[b])       # the consumer has released this set
            env.launch_rollout(T, first, n, traj=bufs.get("traj"), sp_T=bufs.get("state_prime"), obs_T=bufs.get("obs"),
                               rew_T=bufs.get("rew"), done_T=bufs.get("done"), acts_T=bufs.get("actions"),
                               final_ret=self.final_ret[b], final_len=self.final_len[b], carry=self.carry, stream=st,
                               events=None if events is None else events[s])
            self._done_ev[b][s].record(st)
        env.step_idx += T
        self.episodes += 1
        return k

    def ready(self, k=None):
        """Make the current stream wait for episode k (default: the newest) and return its buffers: the [T, N, ...]
        transition tensors plus "final_ret" / "final_len" [N].  Valid until release(k) + `depth` further collect()s."""
        import torch
        k = self.episodes - 1 if k is None else k
        assert 0 <= k < self.episodes and k >= self.episodes - self.depth, "that episode's buffers were overwritten"
        b = k % self.depth
        cur = torch.cuda.current_stream(self.env.device)
        for ev in self._done_ev[b]:
            cur.wait_event(ev)
        out = dict(self.sets[b])
        if "obs" in out and self.env._soa:
            out["obs"] = out["obs"].transpose(1, 2)
        if "done" in out:
            out["done"] = out["done"].view(torch.bool)
        out["final_ret"], out["final_len"] = self.final_ret[b], self.final_len[b]
        return out

    def release(self, k=None):
        """The consumer is done with episode k's buffers (as far as the work it has enqueued on the current stream)."""
        import torch
        k = self.episodes - 1 if k is None else k
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.env.device))
        self._free_ev[k % self.depth] = ev

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
