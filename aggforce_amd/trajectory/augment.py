"""The Augmenter plug-in interface (reference: trajectory/augment.py:13-110)."""
from abc import ABC, abstractmethod
from typing import Tuple, TypeVar

_A = TypeVar("_A", bound="Augmenter")


class Augmenter(ABC):
    r"""Models a conditional density g(y|x) used to extend a trajectory's state space.

    ``sample(source)`` draws y for every frame of x = source; ``log_gradient(source,
    generated)`` returns (d log g / d x, d log g / d y); ``astype`` returns an instance
    working at the requested precision.

    Optional fast path: an augmenter may define
    ``augment_trajectory(coords, forces, kbt) -> (full_coords, full_forces)`` that performs
    sampling, both log-gradients and the concatenation of AugmentedTrajectory._augment in
    one fused device pass; ``AugmentedTrajectory`` uses it when present.
    """

    @abstractmethod
    def __init__(self) -> None:
        """Parameters controlling the transform go here."""

    @abstractmethod
    def sample(self, source):
        """Generate augmenting positions (n_frames, n_generated, n_dims) from source positions."""

    @abstractmethod
    def log_gradient(self, source, generated) -> Tuple:
        """(grad wrt source, grad wrt generated) of log g(generated | source)."""

    @abstractmethod
    def astype(self: _A, *args, **kwargs) -> _A:
        """Instance of the same kind at a given numerical precision."""
