"""hydra.utils.instantiate stand-in: import `_target_`, call it with the other keys."""
import importlib


def instantiate(cfg, *args, **kwargs):
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg.pop("_recursive_", None)
    kwargs.pop("_recursive_", None)
    kwargs.pop("_convert_", None)
    mod, name = target.rsplit(".", 1)
    fn = getattr(importlib.import_module(mod), name)
    cfg.update(kwargs)
    return fn(*args, **cfg)
