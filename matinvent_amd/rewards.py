"""Synthetic reward with the `Reward.scoring` return contract (rewards/reward.py:8-115 returns
(rewards, prop_dict, failed_mask)).  The reference's property calculators (pymatgen HHI, ALIGNN,
DFT, ...) are out of scope; benchmarks and smoke runs use this."""
import numpy as np


class SyntheticReward:
    def __init__(self, root_dir="rewards", reward_threshold=0.8, seed=7, mode="uniform", **kwargs):
        self.root_dir, self.threshold, self.mode = root_dir, reward_threshold, mode
        self.rng = np.random.default_rng(seed)

    def scoring(self, samples, label="tmp"):
        strucs = samples[0] if isinstance(samples, tuple) else samples
        n = len(strucs)
        if self.mode == "light":  # a deterministic structure-dependent score: prefers light elements
            z = np.array([np.mean(getattr(s, "species", [50])) for s in strucs], dtype=float)
            r = np.clip(1.0 - z / 94.0, 0.0, 1.0)
        else:
            r = self.rng.random(n)
        return r, {"synthetic": r.copy()}, np.zeros(n, dtype=bool)
