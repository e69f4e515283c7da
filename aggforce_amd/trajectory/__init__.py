"""Trajectory containers and augmenters (reference: trajectory/__init__.py)."""
from .core import ForcesTrajectory, CoordsTrajectory, Trajectory, AugmentedTrajectory
from .augment import Augmenter
from .gausstraj import CondNormal, SimpleCondNormal

# the reference's name for the premapped Gaussian augmenter (JAX-based there)
JCondNormal = CondNormal

__all__ = [
    "ForcesTrajectory",
    "CoordsTrajectory",
    "Trajectory",
    "AugmentedTrajectory",
    "Augmenter",
    "CondNormal",
    "JCondNormal",
    "SimpleCondNormal",
]
