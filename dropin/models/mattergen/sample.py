from matinvent_amd.mattergen import MatterGenSampler  # noqa: F401
