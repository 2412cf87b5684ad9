"""Experience replay with the reference's ReplayBuffer surface and semantics (memory/replay_buffer.py:11-104): the rows of the
buffer and of the new samples are merged, de-duplicated on the composition's reduced formula keeping the highest reward (:78-91),
sorted by reward, cut to `buffer_size`, and THEN only rows with reward > reward_cutoff stay (:72-75); `sample` draws
min(len, sample_size) rows without replacement (:93-101); `memory_purge` drops the rows whose reduced formula is in the given
structures (:103-105).  CPU bookkeeping on <= 100 rows; kept so that BASELINE config 5 (replay enabled) runs."""
import numpy as np

from .structure import reduced_formula


def _formula(data):
    return reduced_formula(int(z) for z in np.asarray(data.atom_types).reshape(-1).tolist())


class ReplayBuffer:
    def __init__(self, buffer_size=100, sample_size=8, reward_cutoff=0.0, seed=0, **kwargs):
        self.buffer_size, self.sample_size, self.reward_cutoff = buffer_size, sample_size, reward_cutoff
        self.rows = []  # (reward, reduced formula, data), sorted by descending reward
        self.rng = np.random.default_rng(seed)

    def __len__(self):
        return len(self.rows)

    def extend(self, data_list, strucs, rewards):
        rows = self.rows + [(float(r), _formula(d), d) for d, r in zip(data_list, rewards)]
        rows.sort(key=lambda x: -x[0])                     # stable: ties keep buffer-before-new order, like sort_values + concat
        seen, uniq = set(), []
        for row in rows:                                   # drop_duplicates(subset=["comp"]) keeps the first = highest reward
            if row[1] not in seen:
                seen.add(row[1])
                uniq.append(row)
        self.rows = [row for row in uniq[: self.buffer_size] if row[0] > self.reward_cutoff]

    def sample(self):
        if not self.rows:
            return [], np.zeros(0)
        idx = self.rng.choice(len(self.rows), size=min(self.sample_size, len(self.rows)), replace=False)
        return [self.rows[i][2] for i in idx], np.array([self.rows[i][0] for i in idx])

    def memory_purge(self, strucs):
        drop = set()
        for s in strucs:
            drop.add(reduced_formula(int(z) for z in (s.species if hasattr(s, "species") else np.asarray(s.atom_types).reshape(-1).tolist())))
        self.rows = [row for row in self.rows if row[1] not in drop]
