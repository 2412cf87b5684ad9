from matinvent_amd.data import CrystalDataset as DiffCSPDataset  # noqa: F401
