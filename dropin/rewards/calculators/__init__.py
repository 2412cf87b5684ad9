from matinvent_amd.rewards import PyMatGen  # noqa: F401
