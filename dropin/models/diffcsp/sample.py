from matinvent_amd.sampling import ATOM_DIST, DEFAULT_STEP_LR, DiffCSPSampler, SampleDataset  # noqa: F401
from matinvent_amd.data import data2struc, lattices_to_params_shape  # noqa: F401
