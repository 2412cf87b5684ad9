from matinvent_amd.cspnet import CSPNet, MAX_ATOMIC_NUM  # noqa: F401
