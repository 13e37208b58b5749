"""The committed bench lines (profiles/r05/bench_default.json, bench_driver_flags.json, bench_mismatched.json: the JSON line of
`python bench.py`, of the driver's `--gpus 1 --steps 20 --warmup 5` and of `--mismatched` on one MI355X box) against the driver's
contract and this round's additions.  A CPU test: it reads the records, it does not run the bench."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles", "r05")
CONTRACT = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict}


def _line(name):
    path = os.path.join(PROF, name)
    if not os.path.exists(path):
        pytest.fail(f"{os.path.relpath(path, ROOT)} is not committed (profiles of the round that changed bench.py's schema)")
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


def _check_roofline(r, envs_per_launch):
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-4
    assert r["north_star_hbm_target_met"] == (r["frac"] >= 0.60)
    assert r["env_steps_per_launch"] == envs_per_launch
    # achieved = bytes priced / the kernel's average launch duration
    assert abs(r["achieved"] - r["bytes_priced"] / (r["avg_kernel_us"] * 1e-6) / 1e9) < 0.02 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] == r["bytes_priced"]
    assert ("model" in r["bytes_priced_source"]) == (r["traffic"] is None)
    assert r["algorithmic_equiv"]["bytes_per_env_step"] == 97


@pytest.mark.parametrize("name", ["bench_default.json", "bench_driver_flags.json"])
def test_committed_bench_line_keeps_the_contract(name):
    d = _line(name)
    for k, t in CONTRACT.items():
        assert k in d and isinstance(d[k], t), (k, type(d.get(k)))
    assert d["metric"] == "env-steps/sec at N parallel envs; trajectory RMSE vs CPU ref" and d["unit"] == "env-steps/s"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["config"]["envs_per_gpu"] == 262144 and "BASELINE config 4" in d["config"]["workload"]
    assert "model" not in d["config"]
    # value = envs x steps / time
    assert abs(d["value"] - 262144 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    # ONE law: the headline runs the library's default
    assert d["config"]["noise_law"] == "collapsed" and d["config"]["noise_law_is_library_default"] is True
    _check_roofline(d["roofline"], 262144 * 51)
    assert d["roofline"]["noise_law"] == "collapsed"
    if name == "bench_driver_flags.json":
        assert d["steps"] == 20 and d["warmup"] == 5 and "short_region_note" in d and "short_region_preconditioning" not in d
    else:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
        assert c["reference_python"]["value_one_process"] > 1000
    o = d["other_noise_law"]
    assert o["noise_law"] == "per_stage" and o["roofline"]["noise_law"] == "per_stage"
    _check_roofline(o["roofline"], 262144 * 51)
    assert d["sustained"]["value"] > 1e10 and d["trajectory_rmse_vs_cpu_ref"]["value"] < 1e-5


def test_committed_bench_line_carries_this_rounds_legs():
    d = _line("bench_default.json")
    f = d["facade"]                                        # the literal drop-in call, measured
    assert f["value"] > f["reference_python"]["record"]["value_one_process"] and f["value"] > 20000
    assert set(f["phases_us"]) == {"mrsim_step_call", "mrsim_host_wait_word", "python_around_them"}
    assert f["phases_us"]["mrsim_host_wait_word"] <= f["wait_with_mrsim_stream_synchronize_us"] + 0.5
    s = d["streaming_point"]                               # N = 2 097 152 on one GPU
    assert s["envs"] == 2097152 and set(s["rollout"]) == {"collapsed", "per_stage"}
    for law in ("collapsed", "per_stage"):
        r = s["rollout"][law]
        assert abs(r["roofline"]["frac"] - r["bytes_per_launch"] / (r["avg_kernel_us"] * 1e-6) / 1e9 / 8000.0) < 2e-3
    assert s["step_kernel"]["algorithmic_bytes_per_env_step"] == 97 and 0.3 < s["step_kernel"]["roofline"]["frac"] < 1.0
    b = d["learner"]["by_batch"]                           # the learner across compute units for large batches
    assert set(b) == {"64", "256", "1024", "4096"}
    assert b["4096"]["multi_workgroup"]["us_per_update"] < 0.2 * b["4096"]["single_workgroup"]["us_per_update"]
    assert b["4096"]["multi_workgroup"]["transitions_per_s"] > 10 * b["64"]["single_workgroup"]["transitions_per_s"]
    rows = {(r["envs"], r["bookkeeping"]): r for r in d["learner"]["reference_shaped_loop"]["rows"]}   # DDPG.train's iteration
    assert rows[(256, "one launch")]["us_per_iteration"] < 0.8 * rows[(256, "pytorch statements")]["us_per_iteration"]
    assert rows[(1, "in the step kernel")]["iterations_per_s"] > 7200   # one env, learner included, against the reference's env alone
    assert rows[(256, "in the step kernel")]["us_per_iteration"] <= 1.05 * rows[(256, "one launch")]["us_per_iteration"]


def test_committed_mismatched_line_has_its_own_roofline_and_counters():
    d = _line("bench_mismatched.json")
    assert d["config"]["is_mismatched"] is True and d["config"]["noise_law"] == "collapsed"
    r = d["roofline"]
    _check_roofline(r, 262144 * 51)
    assert "mismatched" in r["kernel"]
    assert r["traffic"] is not None and "pmc_traffic.json" in r["traffic_source"]      # this model's own PMC passes
    assert r["valu"] is not None and r["valu"]["insts_valu_per_wave_step"] < 320       # 353 static before the 2-D draw
    assert r["valu_issue_frac"] is not None


def test_rank_verification_record_of_the_two_rank_rehearsal():
    d = _line("bench_gpus2_gloo_rehearsal.json")
    v = d["rank_verification"]
    assert d["n_gpus"] == 2 and v["world_size_from_process_group"] == 2 and len(v["devices"]) == 2
    assert v["probe"]["all_equal"] is True and [r["rank"] for r in v["probe"]["per_rank"]] == [0, 1]
    assert v["probe"]["per_rank"][1]["env_id0"] == d["config"]["envs_per_gpu"]
    assert len(d["per_rank_value"]) == 2
