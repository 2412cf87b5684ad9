from matinvent_amd.pipeline import ReinL, get_device  # noqa: F401
