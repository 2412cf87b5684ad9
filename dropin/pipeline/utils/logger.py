from matinvent_amd.logger import CSVLogger, Logger  # noqa: F401
