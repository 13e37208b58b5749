"""tools/collect_profiles.py must refuse to produce profiles/<tag>/ from a round in which a profiler pass failed, is
missing, or describes another build (a rocprofv3 SIGSEGV was hidden behind an older CSV)."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tools", "collect_profiles.py")
EXPECTED = ["kt", "kt_s1", "kt_s1_ps", "kt_s1_mis", "kt_step", "kt_2m", "kt_2m_ps", "kt_2m_step", "instbench"] + \
           [f"{p}_{c}" for c in ("FETCH_SIZE", "WRITE_SIZE")
            for p in ("pmc_f64", "pmc_f32", "pmc_ps_f64", "pmc_mis_f64", "pmc_2m_f64", "pmc_2m_ps_f64", "pmc_2m_step", "pmc_step", "cal", "cal262k")] + \
           [f"valu_{g}_{c}" for g in "abc" for c in ("f64", "f32", "ps", "mis")] + \
           ["kt_actor", "pmc_actor_a", "pmc_actor_b", "pmc_actor_FETCH_SIZE", "pmc_actor_WRITE_SIZE", "valu_a_mixed",
            "kt_actor_bf", "pmc_actor_bf_a", "kt_actor_b1", "pmc_actor_b1_a"]


def _tree(tmp_path, status_lines, sha_ok=True):
    (tmp_path / "mr_rl_amd").mkdir()
    (tmp_path / "bench.py").write_text("# bench\n")
    (tmp_path / "mr_rl_amd" / "libmrsim.so").write_bytes(b"\x7fELF-not-really")
    src = tmp_path / "gpurun_out" / "prof_t9"
    src.mkdir(parents=True)
    if status_lines is not None:     # a round taken in two parts (profile_round.sh <tag> a|b): the collector unites them
        half = len(status_lines) // 2
        (src / "status_a.txt").write_text("\n".join(status_lines[:half]) + "\n")
        (src / "status_b.txt").write_text("\n".join(status_lines[half:]) + "\n")
    h = lambda p: hashlib.sha256(p.read_bytes()).hexdigest()  # noqa: E731
    b, l = h(tmp_path / "bench.py"), h(tmp_path / "mr_rl_amd" / "libmrsim.so")
    if not sha_ok:
        b = "0" * 64
    for part in "ab":
        (src / f"sha_{part}.txt").write_text(f"{b}  /x/bench.py\n{l}  /x/mr_rl_amd/libmrsim.so\n")
        (src / f"manifest_{part}.txt").write_text(f"status_{part}.txt\nsha_{part}.txt\n")
    return src


def _run(tmp_path):
    return subprocess.run([sys.executable, SCRIPT, "t9"], cwd=tmp_path, capture_output=True, text=True)


def test_refuses_when_a_pass_failed(tmp_path):
    _tree(tmp_path, [f"{k} {139 if k == 'pmc_f64_FETCH_SIZE' else 0}" for k in EXPECTED])
    r = _run(tmp_path)
    assert r.returncode != 0 and "pmc_f64_FETCH_SIZE=139" in r.stderr
    assert not (tmp_path / "profiles").exists()


def test_refuses_when_a_pass_is_missing_or_no_status(tmp_path):
    _tree(tmp_path, [f"{k} 0" for k in EXPECTED if k != "valu_b_f64"])
    r = _run(tmp_path)
    assert r.returncode != 0 and "valu_b_f64=absent" in r.stderr and not (tmp_path / "profiles").exists()
    for part in "ab":
        (tmp_path / "gpurun_out" / "prof_t9" / f"status_{part}.txt").unlink()
    r = _run(tmp_path)
    assert r.returncode != 0 and "status*.txt not found" in r.stderr


def test_refuses_another_builds_round(tmp_path):
    _tree(tmp_path, [f"{k} 0" for k in EXPECTED], sha_ok=False)
    r = _run(tmp_path)
    assert r.returncode != 0 and "this tree has" in r.stderr and not (tmp_path / "profiles").exists()


def test_refuses_when_a_csv_is_absent(tmp_path):
    _tree(tmp_path, [f"{k} 0" for k in EXPECTED])  # every pass "succeeded" but left no output
    r = _run(tmp_path)
    assert r.returncode != 0 and "expected exactly one" in r.stderr and not (tmp_path / "profiles").exists()
