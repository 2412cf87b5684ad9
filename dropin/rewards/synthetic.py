from matinvent_amd.rewards import SyntheticReward  # noqa: F401
