from matinvent_amd.suite import DiffCSPSuite  # noqa: F401
