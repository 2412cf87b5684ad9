from matinvent_amd.mattergen import MatterGenModule  # noqa: F401
