from matinvent_amd.rewards import Reward, linear_scaling  # noqa: F401
