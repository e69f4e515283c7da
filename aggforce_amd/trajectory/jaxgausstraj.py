"""Module path of the reference's JAX augmenter (trajectory/jaxgausstraj.py): ``JCondNormal`` is ``CondNormal``
(closed-form log-gradients on the GPU, same constructor and methods -- see ``gausstraj``)."""
from .gausstraj import CondNormal as JCondNormal, SimpleCondNormal, _ident  # noqa: F401

__all__ = ["JCondNormal"]
