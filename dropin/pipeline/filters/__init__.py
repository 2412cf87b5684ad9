"""pipeline/filters of the reference.  `invalid_filter` (opt_filter.py:49-61) is the geometric validity pre-filter of
matinvent_amd (device-side distance / volume / cell-edge quantities; SMACT charge neutrality is not reproduced).  The
MatterSim-relaxation / SUN `OptFilter` wraps external evaluators that need network assets: out of
scope -- a pass-through keeps configs that name it loadable."""
from matinvent_amd.filters import invalid_filter  # noqa: F401


class OptFilter:
    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, sample_data, sample_struc, energies=None):
        return sample_data, sample_struc, {}
