"""Hydra/OmegaConf-compatible configuration front-end (neither package is in the image).

Implements the subset the reference uses (main.py:8-21, configs/**):
  * YAML tree with attribute access (`Config`), `create / load / save / merge / to_container`;
  * `defaults:` list composition with config groups and `group=option` overrides, `key.path=value`,
    `+key=value`, `~key` command-line overrides;
  * `${a.b}` interpolation (whole-node or inside strings, nested) and custom resolvers
    (`${calc:'${eval_size} * 12'}` -- the reference registers `calc` = eval, main.py:8);
  * `instantiate(cfg, **kwargs)`: recursive `_target_` construction with keyword overrides,
    `_recursive_`, `_partial_`;
  * `hydra.run.dir` (the run directory main() changes into).
So the reference's `configs/` tree and command lines work unchanged.
"""
import copy
import functools
import importlib
import os
import re
from typing import Any, Callable, Dict, List

import yaml

_RESOLVERS: Dict[str, Callable] = {}


def register_new_resolver(name: str, fn: Callable, replace: bool = True):
    if name in _RESOLVERS and not replace:
        raise ValueError(f"resolver {name!r} already registered")
    _RESOLVERS[name] = fn


class Config(dict):
    """dict with attribute access (DictConfig stand-in).  Lists hold Config items for mappings."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def __delattr__(self, k):
        del self[k]

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(x):
    if isinstance(x, Config):
        return x
    if isinstance(x, dict):
        return Config({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return [_wrap(v) for v in x]
    return x


def create(obj=None) -> Config:
    return _wrap(copy.deepcopy(obj) if obj is not None else {})


def load(path: str) -> Config:
    with open(path) as f:
        return _wrap(yaml.safe_load(f) or {})


def to_container(cfg, resolve: bool = True):
    if resolve:
        cfg = resolved(cfg)
    if isinstance(cfg, dict):
        return {k: to_container(v, False) for k, v in cfg.items()}
    if isinstance(cfg, list):
        return [to_container(v, False) for v in cfg]
    return cfg


def save(cfg, path: str):
    with open(path, "w") as f:
        yaml.safe_dump(to_container(cfg, resolve=False), f, sort_keys=False)


def merge(*cfgs) -> Config:
    """Deep merge, later arguments win (OmegaConf.merge); lists are replaced."""
    out = Config()
    for c in cfgs:
        if c is None:
            continue
        for k, v in c.items():
            if isinstance(v, dict) and isinstance(out.get(k), dict):
                out[k] = merge(out[k], v)
            else:
                out[k] = copy.deepcopy(_wrap(v))
    return out


# ---- interpolation ---------------------------------------------------------------------------
_INNERMOST = re.compile(r"\$\{([^${}]*)\}")


def _select(root, path: str):
    node = root
    for part in path.split("."):
        if isinstance(node, list):
            node = node[int(part)]
        else:
            node = node[part]
    return node


def _parse_scalar(text: str):
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def _resolve_str(s: str, root, depth=0):
    if depth > 50:
        raise RecursionError(f"interpolation cycle in {s!r}")
    while True:
        m = _INNERMOST.search(s)
        if m is None:
            return s
        body = m.group(1).strip()
        if ":" in body and body.split(":", 1)[0] in _RESOLVERS:
            name, arg = body.split(":", 1)
            arg = arg.strip()
            if len(arg) >= 2 and arg[0] == arg[-1] and arg[0] in "'\"":
                arg = arg[1:-1]
            val = _RESOLVERS[name](arg)
        else:
            val = _resolve_node(_select(root, body), root, depth + 1)
        if m.start() == 0 and m.end() == len(s):
            return val  # whole-node interpolation keeps its type (dict, int, ...)
        s = s[:m.start()] + str(val) + s[m.end():]


def _resolve_node(node, root, depth=0):
    if isinstance(node, str):
        return _resolve_str(node, root, depth) if "${" in node else node
    if isinstance(node, dict):
        return Config({k: _resolve_node(v, root, depth) for k, v in node.items()})
    if isinstance(node, list):
        return [_resolve_node(v, root, depth) for v in node]
    return node


def resolved(cfg, root=None):
    """Copy of `cfg` with every interpolation resolved against `root` (default: cfg itself)."""
    return _resolve_node(cfg, cfg if root is None else root)


# ---- composition -------------------------------------------------------------------------------
def _set_path(cfg, path: str, value, create_missing: bool):
    parts = path.split(".")
    node = cfg
    for p in parts[:-1]:
        if p not in node:
            if not create_missing:
                raise KeyError(f"override {path!r}: key {p!r} does not exist (prefix with + to add)")
            node[p] = Config()
        node = node[p]
    if parts[-1] not in node and not create_missing:
        raise KeyError(f"override {path!r}: key does not exist (prefix with + to add)")
    node[parts[-1]] = _wrap(value)


def compose(config_dir: str, config_name: str = "base", overrides: List[str] = ()) -> Config:
    """hydra.compose: primary config + its `defaults:` groups + command-line overrides.
    Group files land under the group's name (Hydra's default package); the primary config's own
    keys win over group content with the same name (Hydra >= 1.1 `_self_` last)."""
    primary = load(os.path.join(config_dir, config_name + ".yaml"))
    defaults = primary.pop("defaults", [])
    groups: Dict[str, str] = {}
    order: List[str] = []
    for item in defaults:
        if item == "_self_":
            continue
        if isinstance(item, dict):
            for g, opt in item.items():
                g = g.replace("override ", "").strip()
                groups[g] = opt
                order.append(g)
        else:
            raise ValueError(f"unsupported defaults entry {item!r}")
    plain = []
    for ov in overrides:
        key, eq, val = ov.partition("=")
        if eq and key in groups and not key.startswith(("+", "~")):
            groups[key] = val  # config-group selection, e.g. model=diffcsp
        else:
            plain.append(ov)
    cfg = Config()
    for g in order:
        if groups[g] in (None, "null"):
            continue
        cfg[g] = load(os.path.join(config_dir, g, str(groups[g]) + ".yaml"))
    cfg = merge(cfg, primary)
    for ov in plain:
        if ov.startswith("~"):
            node, _, leaf = ov[1:].rpartition(".")
            (_select(cfg, node) if node else cfg).pop(leaf, None)
            continue
        add = ov.startswith("+")
        key, eq, val = ov.lstrip("+").partition("=")
        if not eq:
            raise ValueError(f"cannot parse override {ov!r}")
        _set_path(cfg, key, _parse_scalar(val), create_missing=add)
    return cfg


def run_dir(cfg) -> str:
    """hydra.run.dir of the composed config (configs/base.yaml:27-30), resolved."""
    d = cfg.get("hydra", {}).get("run", {}).get("dir", ".")
    return _resolve_node(d, cfg)


# ---- instantiate -----------------------------------------------------------------------------------
def _locate(target: str):
    mod, _, name = target.rpartition(".")
    obj = importlib.import_module(mod)
    return getattr(obj, name)


def instantiate(cfg, *args, **kwargs):
    """hydra.utils.instantiate: build `cfg._target_(**cfg, **kwargs)`; nested mappings / list items
    carrying `_target_` are built first (depth first) unless `_recursive_: false`.  Keyword
    arguments override config keys and are themselves instantiated when they are configs."""
    if isinstance(cfg, list):
        return [instantiate(c) if isinstance(c, (dict, list)) else c for c in cfg]
    if not isinstance(cfg, dict):
        return cfg
    node = Config(cfg)
    for k, v in kwargs.items():
        node[k] = v
    recursive = node.pop("_recursive_", True)
    partial = node.pop("_partial_", False)
    node.pop("_convert_", None)
    target = node.pop("_target_", None)

    def build(v):
        if not recursive:
            return v
        if isinstance(v, dict):
            return instantiate(v) if "_target_" in v else Config({k: build(x) for k, x in v.items()})
        if isinstance(v, list):
            return [build(x) for x in v]
        return v

    built = {k: build(v) for k, v in node.items()}
    if target is None:
        return Config(built)
    fn = _locate(target) if isinstance(target, str) else target
    return functools.partial(fn, *args, **built) if partial else fn(*args, **built)


register_new_resolver("calc", eval)  # main.py:8
