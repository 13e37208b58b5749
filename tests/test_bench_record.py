"""The committed bench lines (profiles/r03/bench_default.json, bench_driver_flags.json: what `python bench.py` printed on the GPU
box) against the driver's contract: the keys it reads, their types, and the internal consistency of the figures."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = json.load(open(os.path.join(ROOT, "BASELINE.json")))


@pytest.mark.parametrize("name", ["bench_default.json", "bench_driver_flags.json"])
def test_committed_bench_line_keeps_the_contract(name):
    d = json.load(open(os.path.join(ROOT, "profiles", "r03", name)))
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict), ("per_rank_value", list)):
        assert isinstance(d[key], typ), key
    assert d["metric"] == BASE["metric"] and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["n_gpus"] == 1 and "workload" in d["config"]
    assert "model" not in d["config"]
    total = d["config"]["total_envs"]
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]        # value = envs x steps / time
    assert len(d["per_rank_value"]) == 1 and abs(d["per_rank_value"][0] - d["value"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_frac", "frac_at_spec_issue_rates",
                "hbm_frac_of_achievable", "north_star_hbm_target_met"):
        assert key in r, key
    assert 0.0 < r["frac"] <= 1.0 and 0.0 < r["hbm_frac"] < 1.0 and r["north_star_hbm_target_met"] is (r["hbm_frac"] >= 0.6)
    assert abs(r["hbm_frac"] - r["hbm_achieved_GBs"] / r["hbm_peak_GBs"]) < 1e-3
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == d["unit"]
    assert c["reference_python"]["value_one_process"] > 0 and c["reference_python"]["cores"] >= 1                                            # the reference's own path, quoted
    a = d["actor_in_loop"]
    assert a["roofline"]["bound"] == "mfma" and 0 < a["roofline"]["frac"] < 1
    for k in ("bf16x3", "bf16"):
        assert a[k]["value"] > a["value"] and a[k]["roofline"]["bound"] == "mfma"
    assert d["sustained"]["value"] > 0.8 * 1e11 and d["mixed_trajectory_set"]["value"] > 0
