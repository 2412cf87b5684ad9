"""Synthetic reward with the `Reward.scoring` return contract (rewards/reward.py:8-115 returns
(rewards, prop_dict, failed_mask)).  The reference's property calculators (pymatgen HHI, ALIGNN,
DFT, ...) are out of scope; benchmarks and smoke runs use this."""
import numpy as np


class SyntheticReward:
    def __init__(self, root_dir="rewards", reward_threshold=0.8, seed=7, mode="uniform", n_props=1, reduce="mean", **kwargs):
        assert reduce in ("mean", "min")
        self.root_dir, self.threshold, self.mode = root_dir, reward_threshold, mode
        self.n_props, self.reduce = int(n_props), reduce
        self.rng = np.random.default_rng(seed)

    def scoring(self, samples, label="tmp"):
        strucs = samples[0] if isinstance(samples, tuple) else samples
        n = len(strucs)
        if self.mode == "light":  # a deterministic structure-dependent score: prefers light elements
            z = np.array([np.mean(getattr(s, "species", [50])) for s in strucs], dtype=float)
            r = np.clip(1.0 - z / 94.0, 0.0, 1.0)
        else:
            r = self.rng.random(n)
        props = {"synthetic": r.copy()}
        for k in range(1, self.n_props):  # multi-objective (BASELINE config 5): extra columns, reduced like rewards/reward.py:102-106
            props[f"synthetic_{k}"] = self.rng.random(n)
        cols = np.stack(list(props.values()))
        r = cols.mean(axis=0) if self.reduce == "mean" else cols.min(axis=0)
        return r, props, np.zeros(n, dtype=bool)
