"""Module path of the reference's noised maps (qp/jgauss.py): the same four functions (``gauss``)."""
from .gauss import joptgauss_map, stagedjforcegauss_map, stagedjoptgauss_map, stagedjslicegauss_map

__all__ = ["joptgauss_map", "stagedjoptgauss_map", "stagedjslicegauss_map", "stagedjforcegauss_map"]
