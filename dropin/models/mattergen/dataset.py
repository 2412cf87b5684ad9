from matinvent_amd.mattergen import MatterGenDataset  # noqa: F401
