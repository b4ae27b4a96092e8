"""Hydra-compatible command line for the three seed-label CLIs.

The reference's entry points are ``@hydra.main(config_path="configs/",
config_name=...)`` functions (``pre_compute_pp_score.py:83``,
``generate_mask.py:30``, ``gen_label_files.py:31``) driven by ``key=value``
overrides (``README.md:52-70``).  hydra-core / omegaconf are not part of this
image, so the subset of Hydra 1.x semantics those CLIs rely on is provided here
on top of PyYAML:

  * ``defaults: [- data_paths: fw70_2m.yaml]`` config groups, overridable as
    ``data_paths=nusc.yaml`` (with or without the ``.yaml`` suffix);
  * dotted overrides ``a.b.c=value`` (values parsed as YAML), ``+key=value`` to
    add keys, ``~key`` to delete;
  * ``${key}``, ``${a.b}``, ``${hydra:runtime.cwd}`` and ``${hydra:run.dir}``
    interpolation, resolved lazily on access;
  * ``???`` mandatory values (``MissingMandatoryValue`` on access);
  * attribute / item access, ``.get``, ``**cfg.section`` unpacking,
    ``to_yaml`` and ``save`` (the CLIs dump their resolved config next to their
    outputs, ``generate_mask.py:38-46``).

Unlike Hydra 1.x the working directory is NOT changed to ``outputs/<date>/``;
every path in the shipped configs hangs off ``work_dir = ${hydra:runtime.cwd}``
precisely because the reference has to undo that chdir.
"""
from __future__ import annotations

import copy
import datetime
import os
import re
import sys
from typing import Any, Callable, Dict, List, Optional

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")
_INTERP = re.compile(r"\$\{([^${}]+)\}")


class MissingMandatoryValue(KeyError):
    pass


class ConfigNode(dict):
    """dict with attribute access and lazy ``${...}`` resolution against the root."""

    def __init__(self, data: Optional[dict] = None, root: Optional["ConfigNode"] = None, path: str = ""):
        super().__init__()
        object.__setattr__(self, "_root", root if root is not None else self)
        object.__setattr__(self, "_path", path)
        for k, v in (data or {}).items():
            dict.__setitem__(self, k, self._wrap(k, v))

    def _wrap(self, k, v):
        if isinstance(v, dict) and not isinstance(v, ConfigNode):
            return ConfigNode(v, root=self._root, path=f"{self._path}.{k}" if self._path else str(k))
        return v

    # --- resolution
    def _lookup(self, dotted: str):
        if dotted.startswith("hydra:"):
            return _hydra_resolver(dotted[6:])
        node: Any = self._root
        for part in dotted.split("."):
            node = node[part] if not isinstance(node, list) else node[int(part)]
        return node

    def _resolve(self, v, key="?"):
        if isinstance(v, str):
            if v == "???":
                full = f"{self._path}.{key}" if self._path else key
                raise MissingMandatoryValue(f"Missing mandatory value: {full} (pass {full}=...)")
            m = _INTERP.fullmatch(v)
            if m:                      # whole-string interpolation keeps the referenced type
                return self._lookup(m.group(1).strip())
            if "${" in v:
                return _INTERP.sub(lambda mm: str(self._lookup(mm.group(1).strip())), v)
        elif isinstance(v, list):
            return [self._resolve(x, key) for x in v]
        return v

    def __getitem__(self, k):
        return self._resolve(dict.__getitem__(self, k), k)

    def __getattr__(self, k):
        try:
            return self[k]
        except MissingMandatoryValue:
            raise
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __setitem__(self, k, v):
        dict.__setitem__(self, k, self._wrap(k, v))

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __iter__(self):
        # a Python-level __iter__ also forces ``**node`` through keys()/__getitem__
        # (CPython would otherwise copy the raw, unresolved storage)
        return iter(list(dict.keys(self)))

    def keys(self):
        return dict.keys(self)

    def items(self):
        return [(k, self[k]) for k in dict.keys(self)]

    def values(self):
        return [self[k] for k in dict.keys(self)]

    def to_container(self, resolve: bool = True) -> dict:
        out = {}
        for k in dict.keys(self):
            v = dict.__getitem__(self, k)
            if isinstance(v, ConfigNode):
                out[k] = v.to_container(resolve)
            else:
                try:
                    out[k] = self._resolve(v, k) if resolve else v
                except MissingMandatoryValue:
                    out[k] = "???"
        return out


_RUNTIME = {"cwd": None, "run_dir": None}


def _hydra_resolver(what: str):
    if _RUNTIME["cwd"] is None:
        _RUNTIME["cwd"] = os.getcwd()
    if what == "runtime.cwd":
        return _RUNTIME["cwd"]
    if what == "run.dir":
        if _RUNTIME["run_dir"] is None:
            now = datetime.datetime.now()
            _RUNTIME["run_dir"] = os.path.join("outputs", now.strftime("%Y-%m-%d"), now.strftime("%H-%M-%S"))
        return _RUNTIME["run_dir"]
    raise KeyError(f"unsupported hydra resolver key: {what}")


def to_yaml(cfg: ConfigNode, resolve: bool = False) -> str:
    return yaml.safe_dump(cfg.to_container(resolve=resolve), sort_keys=False, default_flow_style=None)


def save(config: ConfigNode, f: str, resolve: bool = False) -> None:
    with open(f, "w") as fh:
        fh.write(to_yaml(config, resolve=resolve))


def _load_yaml(path: str) -> dict:
    with open(path, "r") as fh:
        return yaml.safe_load(fh) or {}


def _set_dotted(d: dict, dotted: str, value, must_exist: bool):
    parts = dotted.split(".")
    node = d
    for p in parts[:-1]:
        if p not in node or not isinstance(node[p], dict):
            if must_exist:
                raise KeyError(f"Could not override '{dotted}': key '{p}' is not in the config (use +{dotted}=...)")
            node[p] = {}
        node = node[p]
    if must_exist and parts[-1] not in node:
        raise KeyError(f"Could not override '{dotted}': no such key in the config (use +{dotted}=... to add it)")
    node[parts[-1]] = value


def compose(config_name: str, overrides: Optional[List[str]] = None, config_dir: str = CONFIG_DIR) -> ConfigNode:
    """Load ``<config_dir>/<config_name>`` with its defaults list and apply overrides."""
    overrides = list(overrides or [])
    if not config_name.endswith(".yaml"):
        config_name += ".yaml"
    base = _load_yaml(os.path.join(config_dir, config_name))
    defaults = base.pop("defaults", []) or []
    groups: Dict[str, str] = {}
    for entry in defaults:
        if isinstance(entry, dict):
            for g, choice in entry.items():
                groups[g] = choice
    rest = []
    for ov in overrides:
        if "=" in ov:
            k, v = ov.split("=", 1)
            if k.lstrip("+") in groups and "." not in k:
                groups[k.lstrip("+")] = v
                continue
        rest.append(ov)
    merged: dict = {}
    for g, choice in groups.items():
        choice = str(choice)
        fn = choice if choice.endswith(".yaml") else choice + ".yaml"
        path = os.path.join(config_dir, g, fn)
        if not os.path.exists(path):
            avail = sorted(os.listdir(os.path.join(config_dir, g)))
            raise FileNotFoundError(f"config group {g}: no option '{choice}' (available: {avail})")
        merged[g] = _load_yaml(path)
    for k, v in base.items():          # primary config overrides group content on key clash
        if isinstance(v, dict) and isinstance(merged.get(k), dict):
            merged[k].update(v)
        else:
            merged[k] = v
    for ov in rest:
        if ov.startswith("~"):            # Hydra: delete the key
            node, parts = merged, ov[1:].split("=")[0].split(".")
            for part in parts[:-1]:
                node = node.get(part) if isinstance(node, dict) else None
                if node is None:
                    break
            if isinstance(node, dict) and parts[-1] in node:
                del node[parts[-1]]
            else:
                raise KeyError(f"Could not delete '{ov[1:]}': no such key in the config")
            continue
        if "=" not in ov:
            raise ValueError(f"cannot parse override '{ov}' (expected key=value)")
        k, v = ov.split("=", 1)
        add = k.startswith("+")
        k = k.lstrip("+")
        _set_dotted(merged, k, yaml.safe_load(v) if v != "" else "", must_exist=not add)
    return ConfigNode(merged)


def main(config_name: str) -> Callable:
    """Decorator in the shape of ``hydra.main``: ``fn(cfg)`` becomes a CLI taking
    ``key=value`` overrides from ``sys.argv``; calling it with a ConfigNode / dict
    bypasses argv (used by tests)."""

    def deco(fn):
        def wrapper(cfg=None, argv: Optional[List[str]] = None):
            if cfg is None:
                cfg = compose(config_name, sys.argv[1:] if argv is None else argv)
            elif not isinstance(cfg, ConfigNode):
                cfg = ConfigNode(copy.deepcopy(dict(cfg)))
            return fn(cfg)

        wrapper.__name__ = fn.__name__
        wrapper.__doc__ = fn.__doc__
        wrapper.__wrapped__ = fn
        return wrapper

    return deco
