"""Map objects (reference: map/__init__.py)."""
from .core import LinearMap, CLAMap, trjdot
from .tmap import (
    TMap,
    SeperableTMap,
    CLAFTMap,
    AugmentedTMap,
    ComposedTMap,
    NullForcesTMap,
    RATMap,
)
from .tools import lmap_augvariables, smear_map

__all__ = [
    "LinearMap",
    "CLAMap",
    "trjdot",
    "TMap",
    "SeperableTMap",
    "CLAFTMap",
    "AugmentedTMap",
    "ComposedTMap",
    "NullForcesTMap",
    "RATMap",
    "lmap_augvariables",
    "smear_map",
]
