"""matinvent_amd: MI355X (gfx950) hot path of MatInvent's RL-diffusion inner loop.

Host-side mirror of the reference's plug-in surface (models/suite, models/diffcsp,
pipeline/mat_invent.py) over the C ABI declared in include/matinvent_hip.h.
"""
__version__ = "0.1.0"
