"""Type alias for molecular constraints (reference: constraints/hints.py:7)."""
from typing import FrozenSet, Set

Constraints = Set[FrozenSet[int]]
