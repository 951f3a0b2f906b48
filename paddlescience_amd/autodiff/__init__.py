from .ad import clear, hessian, jacobian  # noqa: F401

__all__ = ["jacobian", "hessian", "clear"]
