"""Module path of the reference's NumPy augmenter (trajectory/simplegausstraj.py)."""
from .gausstraj import SimpleCondNormal

__all__ = ["SimpleCondNormal"]
