"""ppsci.equation.ide (/root/reference/ppsci/equation/ide/__init__.py)."""
from .volterra import Volterra  # noqa: F401

__all__ = ["Volterra"]
