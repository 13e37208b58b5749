"""CPU, build container only: the oracle against the LIVE reference (imported unmodified from /root/reference)
on freshly drawn scenarios -- beyond the committed goldens.  Skipped wherever the reference is absent (e.g. the
GPU box).  MR_simulator.py needs only numpy + scipy, so no stand-ins are involved here."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle as O

REF = os.environ.get("MRSIM_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "MR_simulator.py")),
                                reason="reference checkout not present")


def _ref_sim():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import MR_simulator
    return MR_simulator


def _f32(a):
    return np.asarray(a, np.float64).astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("seed", range(6))
def test_sigma0_random_scenarios(seed):
    MS = _ref_sim()
    rng = np.random.default_rng(1000 + seed)
    mis = bool(seed % 2)
    a0 = float(rng.uniform(0.5, 2.0))
    init = rng.uniform(-150, 150, 2) if seed % 3 else rng.uniform(-0.5, 0.5, 2)
    T = 300
    acts = _f32(np.stack([rng.uniform(-20, 20, T), rng.uniform(-2 * np.pi, 2 * np.pi, T)], 1))
    acts[rng.uniform(size=T) < 0.15] = 0.0
    s = MS.Simulator(); s.noise_var = 0.0; s.a0 = a0
    s.reset_start_pos(init.copy()); s.is_mismatched = mis
    o = O.Sim(O.default_params(a0=a0, sigma=0.0, mismatched=int(mis)))
    o.reset(*init)
    for k, (f, al) in enumerate(acts):
        y_ref = s.step(f, al)
        y = o.step(f, al)
        assert np.abs(y - y_ref).max() < 1e-10, (seed, k)
        assert abs(o.e.h_abs - s.integrator.h_abs) < 1e-12 and abs(o.e.t - s.integrator.t) < 1e-12


@pytest.mark.parametrize("seed", range(4))
def test_noise_tape_random_scenarios(seed):
    """sigma > 0: record the reference's numpy.random.normal draws and replay them through the oracle."""
    MS = _ref_sim()
    rng = np.random.default_rng(2000 + seed)
    mis = bool(seed % 2)
    sigma = float(rng.choice([0.25, 0.5, 1.0, 2.0]))
    init = rng.uniform(80, 130, 2) if seed < 2 else rng.uniform(-1, 1, 2)
    T = 120
    acts = _f32(np.stack([rng.uniform(0, 20, T), rng.uniform(0, 2 * np.pi, T)], 1))
    tape = []
    orig = np.random.normal

    def rec(loc=0.0, scale=1.0, size=None):
        v = orig(loc, scale, size)
        tape.extend(np.atleast_1d(v).tolist())
        return v

    np.random.seed(77 + seed)
    np.random.normal = rec
    try:
        s = MS.Simulator(); s.noise_var = sigma; s.a0 = 1.0
        s.reset_start_pos(init.copy()); s.is_mismatched = mis
        ref = np.array([s.step(f, al).copy() for f, al in acts])
    finally:
        np.random.normal = orig
    o = O.Sim(O.default_params(a0=1.0, sigma=sigma, mismatched=int(mis)), noise_kind=O.NOISE_TAPE, tape=np.array(tape))
    o.reset(*init)
    for k, (f, al) in enumerate(acts):
        assert np.abs(o.step(f, al) - ref[k]).max() < 1e-10, (seed, k)
    assert o.nz.tape_pos == len(tape)
