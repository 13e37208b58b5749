#!/usr/bin/env python3
"""Turn the scratch output of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the committed evidence under
profiles/<tag>/:  kernel_stats_bench_{rollout,step}.csv, bench_under_rocprof_rollout.json and pmc_traffic.json.

Usage (in the repo root, after `gpurun -- bash tools/profile_round.sh r01`):  python tools/collect_profiles.py r01
"""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
dst = f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    f = sorted(glob.glob(pattern), key=os.path.getmtime)
    return f[-1] if f else None


for sub, name in (("kt", "kernel_stats_bench_rollout.csv"), ("kt_step", "kernel_stats_bench_step.csv")):
    f = newest(f"{src}/{sub}/*/*kernel_stats.csv")
    if f:
        shutil.copy(f, f"{dst}/{name}")
        print("copied", f, "->", name)
if os.path.exists(f"{src}/kt_bench.json"):
    line = [l for l in open(f"{src}/kt_bench.json").read().splitlines() if l.startswith("{")]
    if line:
        open(f"{dst}/bench_under_rocprof_rollout.json", "w").write(line[-1] + "\n")


def counter_avg(dirname, counter, kernel_sub):
    """mean Counter_Value over the dispatches of kernels whose name contains kernel_sub"""
    f = newest(f"{src}/{dirname}/*/*counter_collection.csv")
    if not f:
        return None, 0
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
         if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]]
    return (sum(v) / len(v), len(v)) if v else (None, 0)


ALGO = 97
out = {
    "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B, "
             "MI355X_MICROARCH.md HBM section), WRITE_SIZE is exact",
    "commands": "tools/profile_round.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs of `python3 bench.py "
                "--no-cpu-baseline --steps 102 [--mode step --launch eager]` and of tools/membench); tools/collect_profiles.py",
    "calibration": {}, "kernels": {},
}
for label, d, n in (("membench pattern<256> n=16777216", "cal", 16777216), ("membench pattern<256> n=262144", "cal262k", 262144)):
    fr, _ = counter_avg(f"{d}_FETCH_SIZE", "FETCH_SIZE", "pattern<256, 0>")
    wr, _ = counter_avg(f"{d}_WRITE_SIZE", "WRITE_SIZE", "pattern<256, 0>")
    if fr is None or wr is None:
        continue
    out["calibration"][label] = {"known_read_KiB": n * 44 / 1024, "FETCH_SIZE_raw_KiB": fr, "FETCH_SIZE_x2_KiB": 2 * fr,
                                 "known_write_KiB": n * 61 / 1024, "WRITE_SIZE_KiB": wr}
N = 262144
for label, pre, sub, units in (("mr_rollout_kernel<RK45,fast,nominal> T=51 N=262144", "pmc", "mr_rollout_kernel<true, 2, false", N * 51),
                               ("mr_step_kernel<RK45,fast,nominal,aos> N=262144", "pmc_step", "mr_step_kernel<true, 2, false, true", N)):
    fr, nf = counter_avg(f"{pre}_FETCH_SIZE", "FETCH_SIZE", sub)
    wr, nw = counter_avg(f"{pre}_WRITE_SIZE", "WRITE_SIZE", sub)
    if fr is None or wr is None:
        continue
    b = (2 * fr + wr) * 1024
    out["kernels"][label] = {"FETCH_SIZE_raw_KiB": fr, "WRITE_SIZE_KiB": wr, "dispatches_averaged": [nf, nw],
                             "hbm_bytes_per_launch": b, "algorithmic_bytes_per_launch": units * ALGO,
                             "traffic_over_algorithmic": b / (units * ALGO), "bytes_per_env_step": b / units}
if out["kernels"]:
    json.dump(out, open(f"{dst}/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))
else:
    print("no PMC output found under", src, "- pmc_traffic.json left untouched")
