from matinvent_amd.schedules import BetaScheduler, SigmaScheduler, d_log_p_wrapped_normal, sigma_norm  # noqa: F401
