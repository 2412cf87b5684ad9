from pipeline.filters import OptFilter, invalid_filter  # noqa: F401
