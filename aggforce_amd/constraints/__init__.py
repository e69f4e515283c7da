"""Molecular-constraint helpers (reference: constraints/__init__.py)."""
from .hints import Constraints
from .constfinder import guess_pairwise_constraints
from .tools import reduce_constraint_sets, constraint_lookup_dict, group_layout, groups_csr

__all__ = [
    "Constraints",
    "guess_pairwise_constraints",
    "reduce_constraint_sets",
    "constraint_lookup_dict",
    "group_layout",
    "groups_csr",
]
