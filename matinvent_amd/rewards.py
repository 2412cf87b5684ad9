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


def linear_scaling(values, minv=0.0, maxv=6.0):
    """rewards/reward.py:8-12: (v - minv) / (maxv - minv) clipped to [0, 1]."""
    ss = (np.asarray(values, dtype=float) - minv) / (maxv - minv)
    return np.clip(ss, 0.0, 1.0)


class Reward:
    """rewards/reward.py:33-115 -- the scalarisation the RL loop ranks samples with: every property is mapped to [0, 1] by
    `linear_scaling` in one of three modes (`ascending`; `descending` = scale -v between -maxv and -minv; a float target = scale
    -|v - target| the same way), the scaled columns are reduced (`mean` / `min` / `weight`: weighted sum), and samples for which
    any calculator returned NaN get reward 0 and are reported in the failed mask.  `prop_cfg` entries need `name`, `calculator`
    (an object with `.calc(samples, label) -> array`), `target`, `minv`, `maxv` (+ `weight`)."""

    def __init__(self, root_dir, prop_cfg, reward_threshold, reduce="mean", **kwargs):
        assert reduce in ("mean", "min", "weight")
        self.root_dir, self.prop_cfg, self.threshold, self.reduce, self.cfg = root_dir, prop_cfg, reward_threshold, reduce, kwargs
        import os
        os.makedirs(self.root_dir, exist_ok=True)

    @staticmethod
    def _get(cfg, key):
        return cfg[key] if isinstance(cfg, dict) or hasattr(cfg, "__getitem__") and not hasattr(cfg, key) else getattr(cfg, key)

    def calc_props(self, samples, label="tmp"):
        prop_dict, raw = {}, []
        for c in self.prop_cfg:
            p = np.asarray(self._get(c, "calculator").calc(samples, label), dtype=float)
            raw.append(p)
            prop_dict[self._get(c, "name")] = np.nan_to_num(p, nan=0.0).astype(float)
        return prop_dict, np.isnan(np.array(raw)).any(axis=0)

    def scoring(self, samples, label="tmp"):
        prop_dict, failed = self.calc_props(samples, label)
        scaled = {}
        for c in self.prop_cfg:
            name, target, minv, maxv = (self._get(c, k) for k in ("name", "target", "minv", "maxv"))
            if target == "ascending":
                scaled[name] = linear_scaling(prop_dict[name], minv, maxv)
            elif target == "descending":
                scaled[name] = linear_scaling(-prop_dict[name], -maxv, -minv)
            elif isinstance(target, float):
                scaled[name] = linear_scaling(-np.abs(prop_dict[name] - target), -maxv, -minv)
            else:
                raise TypeError("prop cfg.target must be a float or descending or ascending")
        cols = np.array(list(scaled.values()))
        if self.reduce == "mean":
            rewards = cols.mean(axis=0)
        elif self.reduce == "min":
            rewards = cols.min(axis=0)
        else:
            rewards = np.array([scaled[self._get(c, "name")] * self._get(c, "weight") for c in self.prop_cfg]).sum(axis=0)
        rewards[failed] = 0.0
        return rewards, prop_dict, failed


class PyMatGen:
    """The composition / cell properties of rewards/calculators/pymatgen/calc.py:163-205 that need no external database:
    `density` (g/cm^3, :45-53).  `hhi` (:57-73) is the mass-fraction-weighted Herfindahl-Hirschman index of the elements'
    geological reserves; pymatgen ships that table as data (hhi.csv), which is not available offline -- pass `hhi_table`
    (CSV: symbol, HHI production, HHI reserve) to use it; without the table the task fails per sample (NaN -> reward 0, as the
    reference's calculators do on error, :84-89)."""

    def __init__(self, root_dir="rewards", task="density", hhi_table=None, **kwargs):
        assert task in ("density", "hhi"), f"task {task!r} needs pymatgen / SMACT data that this build does not carry"
        self.root_dir, self.task = root_dir, task
        self.hhi = None
        if hhi_table is not None:
            import csv
            with open(hhi_table) as f:
                self.hhi = {r[0].strip(): float(r[2]) for r in csv.reader(f) if len(r) >= 3 and not r[0].startswith("#")}

    def calc(self, samples, label="tmp"):
        from .structure import MASSES, SYMBOLS, density
        strucs = samples[0] if isinstance(samples, tuple) else samples
        out = []
        for s in strucs:
            try:
                if self.task == "density":
                    out.append(density(s.species, s.lengths, s.angles))
                else:
                    if self.hhi is None:
                        raise KeyError("no HHI table")
                    m = np.array([MASSES[int(z)] for z in s.species])
                    out.append(float(sum(mi / m.sum() * self.hhi[SYMBOLS[int(z)]] for mi, z in zip(m, s.species))))
            except Exception:
                out.append(np.nan)
        return np.array(out, dtype=float)
