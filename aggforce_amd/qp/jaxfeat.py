"""Module path of the reference's JAX featuriser (qp/jaxfeat.py): ``gb_feat`` is the HIP one (``gbfeat``)."""
from .gbfeat import gb_feat

__all__ = ["gb_feat"]
