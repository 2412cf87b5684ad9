from matinvent_amd.suite import DiffCSPSuite, MatterGenSuite  # noqa: F401
