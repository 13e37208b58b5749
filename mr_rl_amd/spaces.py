"""Minimal `Box` with the members the reference's callers read (MR_env.py:34-45,
RL/MR_ddpg.py:343-345: `.shape[0]`, `.high`, `.low`, `.sample()`, `.contains()`).
gym / gymnasium are used when importable; neither is a dependency."""
import numpy as np


class Box:
    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        """Numeric bounds test (gym's dtype rule changed across versions)."""
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


def make_box(low, high, seed=None):
    """seed: seeds the space's own sampler (MR_Env.reset(init=None) draws init_space.sample(), MR_env.py:172-173), so two
    envs built with the same seed start in the same places without a separate env.seed() call."""
    box = None
    for mod in ("gymnasium", "gym"):
        try:
            spaces = __import__(mod + ".spaces", fromlist=["Box"])
            box = spaces.Box(low=np.asarray(low, dtype=np.float32), high=np.asarray(high, dtype=np.float32))
            break
        except Exception:
            continue
    if box is None:
        box = Box(low, high)
    if seed is not None:
        box.seed(int(seed))
    return box
