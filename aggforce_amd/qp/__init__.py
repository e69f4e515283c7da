"""Force-map optimisers (reference: qp/__init__.py)."""
from .qplinear import qp_linear_map, qp_form, make_bond_constraint_matrix
from .basicagg import constraint_aware_uni_map
from .featlinearmap import (
    FeatZipper,
    Multifeaturize,
    multifeaturize,
    GeneralizedFeatures,
    GeneralizedFeaturizer,
    qp_feat_linear_map,
    id_feat,
)
from .gbfeat import gb_feat
from .gauss import (
    joptgauss_map,
    stagedjforcegauss_map,
    stagedjoptgauss_map,
    stagedjslicegauss_map,
)

__all__ = [
    "qp_linear_map",
    "qp_form",
    "make_bond_constraint_matrix",
    "constraint_aware_uni_map",
    "FeatZipper",
    "Multifeaturize",
    "multifeaturize",
    "GeneralizedFeatures",
    "GeneralizedFeaturizer",
    "qp_feat_linear_map",
    "id_feat",
    "gb_feat",
    "joptgauss_map",
    "stagedjoptgauss_map",
    "stagedjslicegauss_map",
    "stagedjforcegauss_map",
]
