"""Tiny replacement for the hydra entry points of the reference examples (hydra/omegaconf are not
installed): `python script.py key=value ...` overrides over a defaults dict."""
import ast
import sys


def parse(defaults: dict) -> dict:
    cfg = dict(defaults)
    for a in sys.argv[1:]:
        if "=" not in a:
            raise SystemExit(f"expected key=value, got {a!r}")
        k, v = a.split("=", 1)
        if k not in cfg:
            raise SystemExit(f"unknown option {k!r}; known: {sorted(cfg)}")
        try:
            cfg[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            cfg[k] = v
    return cfg
