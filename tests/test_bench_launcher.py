"""`python bench.py --gpus N` must start N ranks itself: the launcher's
argv / environment, the relay of rank 0's JSON line and the exit-code propagation, all on CPU (no GPU call on that path)."""
import argparse
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_rank_commands_env_and_argv():
    argv = ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    cmds = bench.rank_commands(4, argv, 29555, base_env={"PATH": "/usr/bin", "WORLD_SIZE_UNRELATED": "x"})
    assert len(cmds) == 4
    for r, (cmd, env) in enumerate(cmds):
        assert cmd[0] == sys.executable and cmd[1] == os.path.join(ROOT, "bench.py") and cmd[2:] == argv
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r)
        assert env["WORLD_SIZE"] == "4" and env["LOCAL_WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29555"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"  # dmabuf IPC only on this pool
        assert env["PATH"] == "/usr/bin"                  # caller's environment is kept


def test_gpus_flag_is_parsed_and_defaults_to_one():
    assert bench.parse([]).gpus == 1
    a = bench.parse(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)


def _fake_ranks(monkeypatch, snippet):
    """replace the rank command by a tiny python program (the real one needs a GPU)"""
    def fake(n, argv, port, **kw):
        out = []
        for cmd, env in bench.rank_commands.__wrapped__(n, argv, port, **kw):
            out.append(([sys.executable, "-c", snippet], env))
        return out
    fake.__wrapped__ = bench.rank_commands
    if not hasattr(bench.rank_commands, "__wrapped__"):
        orig = bench.rank_commands
        orig.__wrapped__ = orig
    monkeypatch.setattr(bench, "rank_commands", fake)


def test_launcher_relays_rank0_json_and_world_size(monkeypatch, capfd):
    snippet = ("import os, json, sys\n"
               "r = int(os.environ['RANK'])\n"
               "line = json.dumps({'n_gpus': int(os.environ['WORLD_SIZE']), 'rank': r, 'port': os.environ['MASTER_PORT']})\n"
               "print(line if r == 0 else 'noise from rank %d' % r, flush=True)\n")
    _fake_ranks(monkeypatch, snippet)
    rc = bench.launch_ranks(argparse.Namespace(gpus=3, master_port=0), ["--gpus", "3"])
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0
    assert len(out) == 1, out  # only rank 0's line reaches stdout
    d = json.loads(out[0])
    assert d["n_gpus"] == 3 and d["rank"] == 0 and int(d["port"]) > 0


def test_launcher_propagates_a_failing_rank(monkeypatch, capfd):
    snippet = ("import os, sys, time\n"
               "r = int(os.environ['RANK'])\n"
               "if r == 1:\n"
               "    sys.exit(7)\n"
               "time.sleep(30)\n")  # the other ranks would hang on the lost peer: the launcher must stop them
    _fake_ranks(monkeypatch, snippet)
    import time
    t0 = time.time()
    rc = bench.launch_ranks(argparse.Namespace(gpus=2, master_port=0), ["--gpus", "2"])
    assert rc == 7
    assert time.time() - t0 < 20
    assert "rank 1 exited with code 7" in capfd.readouterr().err


def test_main_takes_launcher_path_only_without_world_size(monkeypatch):
    called = {}
    monkeypatch.setattr(bench, "launch_ranks", lambda args, argv: called.setdefault("argv", argv) and 0 or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "20"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and called["argv"] == ["--gpus", "2", "--steps", "20"]


def test_region_schedule_cuts_launch_groups_at_episode_boundaries():
    """RolloutRegion.schedule(n) = the launch-group lengths run(n) will use (what the timed region primes): groups of at
    most T steps, cut where an episode (ep steps) ends, starting from the steps already done."""
    import bench
    reg = object.__new__(bench.RolloutRegion)
    reg.T, reg.ep, reg.done_steps = 51, 51, 0
    assert reg.schedule(20) == [20]
    assert reg.schedule(51000) == [51] * 1000
    reg.done_steps = 5                      # the driver's --warmup 5
    assert reg.schedule(20) == [20]
    assert reg.schedule(100) == [46, 51, 3]
    reg.T = 20                              # --rollout-len 20: groups never straddle an episode boundary
    reg.done_steps = 0
    assert reg.schedule(60) == [20, 20, 11, 9]
    assert sum(reg.schedule(12345)) == 12345 and reg.schedule(0) == []
