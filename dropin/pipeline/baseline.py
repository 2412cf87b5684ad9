from matinvent_amd.pipeline import Baseline  # noqa: F401
