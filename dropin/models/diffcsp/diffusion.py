from matinvent_amd.diffcsp import DiffCSPModule, SinusoidalTimeEmbeddings, MAX_ATOMIC_NUM  # noqa: F401
