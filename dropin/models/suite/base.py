from matinvent_amd.suite import ModelSuite, get_device  # noqa: F401
