"""CPU, world_size 2 over gloo: the N>1 path -- contiguous shards keyed by GLOBAL env id and the
all-gather of episode returns (mr_rl_amd/dist.py).  The env compute of each rank is stood in for by
the oracle here (tests may use it; the product path has no CPU compute)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mr_rl_amd.dist import all_shards, gather_returns, gather_returns_ragged, shard_of


def test_shard_of_partitions_exactly():
    for total, world in [(2097152, 8), (262144, 1), (10, 3), (7, 8), (1000, 6)]:
        shards = all_shards(total, world)
        assert shards[0][0] == 0 and sum(n for _, n in shards) == total
        for (a, n), (b, _) in zip(shards, shards[1:]):
            assert a + n == b
        assert max(n for _, n in shards) - min(n for _, n in shards) <= 1
    assert shard_of(2097152, 3, 8) == (3 * 262144, 262144)
    with pytest.raises(ValueError):
        shard_of(10, 4, 4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        env_id0, n_local = shard_of(total, rank, world)
        p = O.default_params(sigma=1.0, auto_reset=1)
        v = O.VecOracle(n_local, p, seed=11, env_id0=env_id0)
        v.reset(0)
        lo, hi = [-20.0, -2 * np.pi], [20.0, 2 * np.pi]
        for t in range(1, 52):  # one full episode: every env times out at step 51
            v.step(v.random_policy(t, lo, hi), t)
        local = torch.from_numpy(v.final_ret.astype(np.float32))
        if total % world == 0:
            allr = gather_returns(local)
        else:
            allr = gather_returns_ragged(local, total)
        pos = torch.from_numpy(v.envs["y"].copy())
        q.put((rank, allr.numpy().copy(), env_id0, pos.numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [64, 37])
def test_sharded_run_equals_unsharded_and_gathers_in_global_order(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # unsharded reference run
    from oracle import oracle as O
    p = O.default_params(sigma=1.0, auto_reset=1)
    v = O.VecOracle(total, p, seed=11)
    v.reset(0)
    lo, hi = [-20.0, -2 * np.pi], [20.0, 2 * np.pi]
    for t in range(1, 52):
        v.step(v.random_policy(t, lo, hi), t)
    for rank, allr, env_id0, pos in got:
        np.testing.assert_array_equal(allr, v.final_ret.astype(np.float32))      # same on every rank, global order
        np.testing.assert_array_equal(pos, v.envs["y"][env_id0:env_id0 + len(pos)])  # bit-identical trajectories
    assert np.all(v.final_ret == 510.0)


def test_gather_returns_single_process_is_identity():
    x = torch.arange(5, dtype=torch.float32)
    assert gather_returns(x) is x
    out = torch.zeros(5)
    assert torch.equal(gather_returns(x, out=out), x)


def _gatherer_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mr_rl_amd.dist import ReturnGatherer

        class FakeEnv:  # the only members ReturnGatherer touches
            num_envs = 5
            device = torch.device("cpu")
            final_ret = torch.zeros(5)

        env = FakeEnv()
        g = ReturnGatherer(env, world)
        outs = []
        for ep in range(5):  # more episodes than buffers: exercises buffer reuse
            env.final_ret = torch.arange(5, dtype=torch.float32) + 100 * rank + 1000 * ep
            g.gather()
            outs.append(g.latest().clone().numpy())
        g.finish()
        q.put((rank, outs, g.last_mean()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_return_gatherer_double_buffering_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gatherer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, outs, mean in got:
        for ep, o in enumerate(outs):
            want = np.concatenate([np.arange(5) + 100 * r + 1000 * ep for r in range(world)]).astype(np.float32)
            np.testing.assert_array_equal(o, want)
        assert mean == pytest.approx(float(np.concatenate([np.arange(5) + 100 * r + 4000 for r in range(world)]).mean()))


def _block_gatherer_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mr_rl_amd.collector import BlockReturnGatherer

        class FakeEnv:
            device = torch.device("cpu")

        class FakeCollector:  # the members BlockReturnGatherer touches
            E, N, env = 3, 4, FakeEnv()
            episodes = 0

            def __init__(self):
                self.ret_blocks = [torch.zeros(self.E, self.N) for _ in range(2)]
                self.freed, self.waited = [], []

            def wait_episode(self, k=None):
                self.waited.append(k)

            def free_returns_block(self, blk):
                self.freed.append(blk)

        col = FakeCollector()
        g = BlockReturnGatherer(col, world)
        outs = []
        for k in range(4 * col.E):   # 4 blocks over 2 buffers
            blk, row = (k // col.E) % 2, k % col.E
            col.ret_blocks[blk][row] = torch.arange(col.N, dtype=torch.float32) + 10 * k + 1000 * rank
            col.episodes = k + 1
            g.gather()
            if (k + 1) % col.E == 0:
                outs.append(g.latest().clone().numpy())
        g.finish()
        q.put((rank, outs, g.n_collectives, col.waited, col.freed))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_block_return_gatherer_world2():
    """BlockReturnGatherer over gloo, world size 2: every E-th gather() all-gathers the [E, n] block of returns; every rank
    ends up with [world, E, n] in rank order; a block is handed back to the collector only after its collective."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_block_gatherer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    E, N = 3, 4
    for rank, outs, ncoll, waited, freed in got:
        assert ncoll == 4 and waited == [2, 5, 8, 11]
        assert len(freed) >= 3 and set(freed) <= {0, 1}
        for b, o in enumerate(outs):
            assert o.shape == (world, E, N)
            for r in range(world):
                for j in range(E):
                    k = b * E + j
                    np.testing.assert_array_equal(o[r, j], np.arange(N) + 10 * k + 1000 * r)


def test_block_return_gatherer_follows_the_collectors_launch_groups():
    """Rows and events are indexed by the collector's launch groups (collect() calls), not by a count the gatherer keeps
    itself: calling gather() twice for one group is harmless, skipping a group is an error (a schedule that cuts
    episodes into several groups -- bench.py --steps 20 --warmup 5 -- once made the two counts drift apart)."""
    import torch
    from mr_rl_amd.collector import BlockReturnGatherer

    class FakeEnv:
        device = "cpu"

    class FakeCollector:
        E, N, env, episodes = 2, 3, FakeEnv(), 0

        def __init__(self):
            self.ret_blocks = [torch.zeros(self.E, self.N) for _ in range(2)]

        def wait_episode(self, k=None):
            pass

    col = FakeCollector()
    g = BlockReturnGatherer(col, 1)
    g.gather()                       # nothing collected yet
    assert g.n_gathers == 0 and g.latest() is None
    for k in range(5):
        col.ret_blocks[(k // 2) % 2][k % 2] = float(k)
        col.episodes = k + 1
        g.gather(); g.gather()       # the second call is a no-op
        assert g.n_gathers == k + 1
        if k % 2:
            assert g.latest().shape == (1, 2, 3) and float(g.latest()[0, 1, 0]) == k
    col.episodes = 8                 # groups 5 and 6 never reported
    with pytest.raises(RuntimeError, match="skipped"):
        g.gather()


# ---------------------------------------------------------------------------------------------------------------------
# verify_shards: the self-verification `bench.py --gpus N` writes into its record (rank_verification), over gloo with the oracle
# standing in for the env compute
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_probe(env_id0, n=64, seed=11):
    from oracle import oracle as O
    p = O.default_params(sigma=1.0, auto_reset=1)
    v = O.VecOracle(n, p, seed=seed, env_id0=env_id0)
    v.reset(0)
    lo, hi = [-20.0, -2 * np.pi], [20.0, 2 * np.pi]
    for t in range(1, 52):
        v.step(v.random_policy(t, lo, hi), t)
    return torch.tensor([v.envs["y"][:, 0].sum(), v.envs["y"][:, 1].sum(), float(v.final_ret.sum())], dtype=torch.float64)


def _verify_worker(rank, world, port, total, corrupt_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mr_rl_amd.dist import verify_shards

        def probe(env_id0):
            # a rank that ran the WRONG shard (e.g. env ids offset by one): what the verification must catch
            return _oracle_probe(env_id0 + (1 if (rank == corrupt_rank and rank != 0) else 0))
        out = verify_shards(probe, total, rank, world, device="cpu")
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("corrupt_rank", [-1, 1])
def test_verify_shards_over_gloo_detects_a_rank_on_the_wrong_shard(corrupt_rank):
    world, total = 2, 256
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verify_worker, args=(r, world, port, total, corrupt_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] is None                                   # only rank 0 reports
    rep = got[0]
    assert [r["rank"] for r in rep["per_rank"]] == [0, 1] and [r["env_id0"] for r in rep["per_rank"]] == [0, 128]
    assert rep["per_rank"][0]["equals_rank0_recomputation"] is True
    assert rep["per_rank"][1]["equals_rank0_recomputation"] is (corrupt_rank != 1)
    assert rep["all_equal"] is (corrupt_rank != 1)
    assert len(rep["per_rank"][1]["probe"]) == 3 and rep["per_rank"][1]["probe"][2] == 64 * 510.0


def test_verify_shards_single_process():
    from mr_rl_amd.dist import verify_shards
    rep = verify_shards(lambda e0: torch.tensor([float(e0), 2.0]), 100, 0, 1)
    assert rep["all_equal"] and rep["per_rank"] == [{"rank": 0, "env_id0": 0, "probe": [0.0, 2.0], "equals_rank0_recomputation": True}]
