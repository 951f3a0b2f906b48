"""Stand-ins for `hydra` / `omegaconf`, which the reference's example scripts import and this image does not have.

Test plumbing only: `hydra.main(...)` hands the function back undecorated, `DictConfig` is a read-only attribute view of
the example's own yaml (`conf/<name>.yaml`, parsed with PyYAML, `${a.b}` interpolations resolved, the `hydra:` section
dropped).  The example FILES are imported from /root/reference as they are; sizes are shrunk through `cfg` only."""
from __future__ import annotations

import re
import sys
import types
from collections.abc import Mapping

import yaml


class DictConfig(Mapping):
    def __init__(self, data: dict):
        object.__setattr__(self, "_d", {k: DictConfig(v) if isinstance(v, dict) else v for k, v in data.items()})

    def __getattr__(self, key):
        try:
            return self._d[key]
        except KeyError:
            raise AttributeError(key) from None

    def __getitem__(self, key):
        return self._d[key]

    def __iter__(self):
        return iter(self._d)

    def __len__(self):
        return len(self._d)


def _lookup(root: dict, dotted: str):
    cur = root
    for part in dotted.split("."):
        cur = cur[part]
    return cur


def _resolve(node, root, output_dir):
    if isinstance(node, dict):
        return {k: _resolve(v, root, output_dir) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, output_dir) for v in node]
    if isinstance(node, str) and "${" in node:
        whole = re.fullmatch(r"\$\{([^}]+)\}", node)
        if whole:
            ref = whole.group(1)
            return output_dir if ref.startswith("hydra:") else _resolve(_lookup(root, ref), root, output_dir)
        return re.sub(r"\$\{([^}]+)\}", lambda m: output_dir if m.group(1).startswith("hydra:")
                      else str(_resolve(_lookup(root, m.group(1)), root, output_dir)), node)
    return node


def load_cfg(yaml_path: str, output_dir: str, overrides: dict) -> DictConfig:
    with open(yaml_path) as f:
        raw = yaml.safe_load(f)
    raw.pop("hydra", None)
    raw.pop("defaults", None)
    for dotted, val in overrides.items():
        parts = dotted.split(".")
        cur = raw
        for p in parts[:-1]:
            cur = cur[p]
        cur[parts[-1]] = val
    return DictConfig(_resolve(raw, raw, output_dir))


def install(monkeypatch) -> None:
    hydra = types.ModuleType("hydra")
    hydra.main = lambda *a, **k: (lambda fn: fn)
    omegaconf = types.ModuleType("omegaconf")
    omegaconf.DictConfig = DictConfig
    monkeypatch.setitem(sys.modules, "hydra", hydra)
    monkeypatch.setitem(sys.modules, "omegaconf", omegaconf)
