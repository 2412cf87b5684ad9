from matinvent_amd.suite import MatterGenSuite  # noqa: F401
