"""Minimal experience replay with the reference's ReplayBuffer surface (memory/replay_buffer.py:11-104:
top-K by reward, one entry per composition, random sample above a reward cutoff).  CPU bookkeeping on
<= 100 rows; out of the hot path, kept only so that BASELINE config 5 (replay enabled) runs."""
import numpy as np


def _composition(data):
    z, c = np.unique(np.asarray(data.atom_types), return_counts=True)
    return tuple(zip(z.tolist(), c.tolist()))


class ReplayBuffer:
    def __init__(self, buffer_size=100, sample_size=10, reward_cutoff=0.0, seed=0, **kwargs):
        self.buffer_size, self.sample_size, self.reward_cutoff = buffer_size, sample_size, reward_cutoff
        self.rows = []  # (reward, composition, data)
        self.rng = np.random.default_rng(seed)

    def __len__(self):
        return len(self.rows)

    def extend(self, data_list, strucs, rewards):
        best = {comp: (r, comp, d) for r, comp, d in self.rows}
        for d, r in zip(data_list, rewards):
            if r < self.reward_cutoff:
                continue
            comp = _composition(d)
            if comp not in best or best[comp][0] < r:
                best[comp] = (float(r), comp, d)
        self.rows = sorted(best.values(), key=lambda x: -x[0])[: self.buffer_size]

    def sample(self):
        if not self.rows:
            return [], np.zeros(0)
        idx = self.rng.choice(len(self.rows), size=min(self.sample_size, len(self.rows)), replace=False)
        return [self.rows[i][2] for i in idx], np.array([self.rows[i][0] for i in idx])

    def memory_purge(self, strucs):
        pass
