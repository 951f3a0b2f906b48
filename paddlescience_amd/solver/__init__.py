from .solver import Solver  # noqa: F401

__all__ = ["Solver"]
