from matinvent_amd.pipeline import MatInvent  # noqa: F401
