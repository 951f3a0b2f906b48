from .base import Constraint  # noqa: F401
from .geometric import BoundaryConstraint, InitialConstraint, InteriorConstraint, PeriodicConstraint  # noqa: F401
from .supervised import SupervisedConstraint  # noqa: F401

__all__ = ["Constraint", "InteriorConstraint", "BoundaryConstraint", "InitialConstraint", "PeriodicConstraint", "SupervisedConstraint"]
