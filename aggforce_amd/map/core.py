"""LinearMap and CLAMap: the reference's map objects with GPU-backed application.

Reference: map/core.py (LinearMap 46-317, CLAMap 320-430).  The objects keep the
reference's constructor arguments, properties, algebra and error behaviour; the
contraction itself (``util.trjdot``, util.py:119-125) runs in the HIP kernels K3
(``aggf_linearmap_apply``, MFMA) and K3b (``aggf_slice_gather`` for one-hot rows).
Arrays may be NumPy arrays (copied to the GPU and back) or torch ROCm tensors (results
stay on the device).
"""
from typing import Callable, Dict, Final, List, Literal, Optional, Union

import numpy as np

from .. import _kernels as K
from ..util import trjdot


class _Taggable:
    """Carries a free-form ``tags`` dictionary (reference map/core.py:21-43)."""

    def __init__(self, tags: Union[None, Dict[str, str]]) -> None:
        self.tags = {} if tags is None else tags


class _PendingMap:
    """Result of LinearMap.map_async: finished (and checked) by ``result()``."""

    def __init__(self, out, flag, stream, template) -> None:
        self._out, self._flag, self._stream, self._template = out, flag, stream, template

    def _join(self):
        import torch

        main = torch.cuda.current_stream(self._out.device)
        main.wait_stream(self._stream)
        self._out.record_stream(main)

    def _take_flag_value(self) -> bool:
        """Reads the flag and hands it back to the pool -- once: ``result()`` raising and the caller's ``discard()``
        in a ``finally`` used to return the same flag twice, and two later kernels then shared it."""
        flag, self._flag = self._flag, None
        return False if flag is None else K.read_flag(flag)

    def result(self):
        self._join()
        if self._take_flag_value():
            raise ValueError(
                "NaN handling is on and results seem to depend on NaN "
                "positions in input array. Check input and standard_matrix."
            )
        return K.like_input(self._out, self._template)

    def discard(self) -> None:
        self._join()
        self._take_flag_value()


class LinearMap:
    """Linear fine-grained -> coarse-grained map given by its standard matrix.

    ``standard_matrix`` has shape (n_cg_sites, n_fg_sites).  Calling the map contracts
    it with arrays of shape (n_steps, n_fg_sites, 3).
    """

    n_dim: Final = 3

    def __init__(
        self,
        mapping: Union[List[List[int]], np.ndarray],
        n_fg_sites: Union[int, None] = None,
        handle_nans: Union[bool, Literal["safe"]] = True,
        nan_check_threshold: float = 1e-6,
    ) -> None:
        """Build from a 2-D matrix, or from per-cg-site lists of fg indices (uniform weights).

        ``[[0,2,3],[4]]`` with ``n_fg_sites=6`` gives rows ``[1/3,0,1/3,1/3,0,0]`` and
        ``[0,0,0,0,1,0]`` (reference map/core.py:133-144).  ``handle_nans``: NaN inputs that
        only meet zero coefficients are ignored, any other NaN raises ``ValueError``
        (map/core.py:219-238); ``"safe"`` and ``True`` are equivalent here because the input
        is never modified; ``False`` propagates NaNs like a plain product.
        """
        if hasattr(mapping, "detach"):
            mapping = mapping.detach().cpu().numpy()
        if isinstance(mapping, np.ndarray) and mapping.ndim == 2:
            if n_fg_sites is not None:
                raise ValueError(
                    "Cannot specify n_fg_sites when mapping is ArrayLike. Let it be inferred."
                )
            matrix = mapping
        elif hasattr(mapping, "__iter__"):
            if n_fg_sites is None:
                raise ValueError("n_fg_sites is required when mapping is a list of index lists.")
            rows = [list(r) for r in mapping]
            matrix = np.zeros((len(rows), n_fg_sites))
            for site, members in enumerate(rows):
                row = np.zeros(n_fg_sites)
                row[members] = 1 / len(members)
                matrix[site] = row
        else:
            raise ValueError(f"Cannot understand mapping {mapping}.")
        self._standard_matrix = matrix
        self.handle_nans = handle_nans
        # (min and max propagate NaN and show +-inf: two passes without the temporaries of np.isfinite(matrix).all(),
        # which cost 1 ms per fitted 256 x 4096 force map)
        if self.handle_nans and matrix.size and not (np.isfinite(matrix.min()) and np.isfinite(matrix.max())):
            raise ValueError("Nan checking can only be performed if standard_matrix is itself finite.")
        self.nan_check_threshold = nan_check_threshold
        self._dev_cache: Dict = {}
        self._onehot = None
        self._host_ready = None  # event of an asynchronous download still filling _standard_matrix (see from_device)

    @classmethod
    def from_device(cls, matrix_dev, handle_nans: Union[bool, Literal["safe"]] = True,
                    nan_check_threshold: float = 1e-6) -> "LinearMap":
        """A map whose (n_cg, n_fg) matrix was just computed on the GPU (a fitted force map): the device copy is used
        as it is, the NumPy ``standard_matrix`` is filled by an ASYNCHRONOUS download into pinned memory and waited
        for on first access.  The fit used to download (8 MB at 256 x 4096), check and classify the matrix on the host
        -- 3 ms with the GPU idle -- before the apply kernel could be launched.  The matrix must be finite and is
        taken to be dense (no one-hot rows): what K2 returns when its status says so."""
        import torch

        host = torch.empty(tuple(matrix_dev.shape), dtype=matrix_dev.dtype, pin_memory=True)
        host.copy_(matrix_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(matrix_dev.device))
        self = cls.__new__(cls)
        self._standard_matrix = host.numpy()
        self.handle_nans = handle_nans
        self.nan_check_threshold = nan_check_threshold
        self._dev_cache = {(matrix_dev.dtype, str(matrix_dev.device)): (self._standard_matrix, matrix_dev)}
        self._onehot = (self._standard_matrix, None)
        self._host_ready = ev
        return self

    def __getstate__(self):
        """Maps are plain objects that users pickle (the reference's are): finish a pending download, keep the
        matrix as an ordinary array and drop what belongs to this process (device copies, events)."""
        state = dict(self.__dict__)
        state["_standard_matrix"] = np.array(self.standard_matrix)
        state["_dev_cache"] = {}
        state["_onehot"] = None
        state["_host_ready"] = None
        return state

    # ------------------------------------------------------------------ properties
    @property
    def standard_matrix(self) -> np.ndarray:
        """The mapping in standard matrix format."""
        if self._host_ready is not None:
            self._host_ready.synchronize()
            self._host_ready = None
        return self._standard_matrix

    @property
    def n_cg_sites(self) -> int:
        return self._standard_matrix.shape[0]  # (shape and dtype need no finished download)

    @property
    def n_fg_sites(self) -> int:
        return self._standard_matrix.shape[1]

    @property
    def participating_fg(self) -> List[List[int]]:
        """For each cg site, the fg sites with a positive coefficient."""
        table: List[List[int]] = [[] for _ in range(self.n_cg_sites)]
        for cg, fg in zip(*np.nonzero(self.standard_matrix > 0)):
            table[cg].append(fg)
        return table

    def close_to_identity(self, threshold: float = 1e-8) -> bool:
        """True if square and within ``threshold`` (Frobenius norm) of the identity."""
        m = self.standard_matrix
        if m.shape[0] != m.shape[1]:
            return False
        return bool(np.sqrt(((np.identity(m.shape[0], dtype=m.dtype) - m) ** 2).sum()) <= threshold)

    # ------------------------------------------------------------------ device side
    def _device_matrix(self, tdtype, device):
        key = (tdtype, str(device))
        hit = self._dev_cache.get(key)
        if hit is None or hit[0] is not self._standard_matrix:
            import torch

            t = torch.from_numpy(np.ascontiguousarray(self._standard_matrix)).to(device=device, dtype=tdtype)
            self._dev_cache = {k: v for k, v in self._dev_cache.items() if v[0] is self._standard_matrix}
            self._dev_cache[key] = (self._standard_matrix, t)
            return t
        return hit[1]

    def _onehot_index(self):
        """Atom index per row if every row is a unit vector (slice map), else None."""
        if self._onehot is None or self._onehot[0] is not self._standard_matrix:
            m = self._standard_matrix
            idx = None
            # (a one-hot matrix has exactly one non-zero per row: the count rejects a dense force map in one pass)
            if (m.size and np.count_nonzero(m) == m.shape[0] and np.all((m == 0) | (m == 1))
                    and np.all(m.sum(axis=1) == 1)):
                idx = np.argmax(m, axis=1).astype(np.int32)
            self._onehot = (m, idx)
        return self._onehot[1]

    def _out_dtype(self, points) -> np.dtype:
        dt = np.result_type(K.np_dtype_of(points), self._standard_matrix.dtype)
        return dt if dt in (np.float32, np.float64) else np.dtype(np.float64)

    def __call__(self, points):
        """Map an array of shape (n_steps, n_fg_sites, 3); NaN policy per ``handle_nans``."""
        shape = tuple(points.shape)
        if len(shape) != 3 or shape[2] != self.n_dim or shape[1] != self.n_fg_sites:
            raise ValueError(
                f"points of shape {shape} cannot be mapped by a ({self.n_cg_sites},{self.n_fg_sites}) LinearMap"
            )
        import torch

        out_t = K.torch_dtype(self._out_dtype(points))
        p = K.as_device(points)
        idx = self._onehot_index()
        if idx is not None and self.handle_nans:
            # slice map: a gather.  A NaN at a selected site is exactly the case in which the
            # reference's NaN->0 / NaN->-1 products differ (map/core.py:226-236).
            key = ("idx", str(p.device))
            hit = self._dev_cache.get(key)
            if hit is None or hit[0] is not self._standard_matrix:
                hit = (self._standard_matrix, torch.from_numpy(idx).to(p.device))
                self._dev_cache[key] = hit
            probe = K.take_flag(p.device)
            out = K.slice_gather(p, hit[1], out_t, nan_probe=probe)
            if K.read_flag(probe):
                raise ValueError(
                    "NaN handling is on and results seem to depend on NaN "
                    "positions in input array. Check input and standard_matrix."
                )
            return K.like_input(out, points)
        m = self._device_matrix(out_t, p.device)
        if not self.handle_nans:
            return K.like_input(K.linearmap_apply(p, m), points)
        # plain product with the NaN scan of the input fused into the same pass; only if a NaN was seen
        # (rare) are the reference's two extra products formed (map/core.py:226-236)
        probe = K.take_flag(p.device)
        out = K.linearmap_apply(p, m, nan_probe=probe)
        # (the probe is conservative: the LDS-DMA tile kernel reports NaNs of the OUTPUT, which an infinity meeting a
        # zero coefficient produces too; the reference's test is on the input, map/core.py:13-16)
        if not K.read_flag(probe) or not K.has_nan(p):
            return K.like_input(out, points)
        raw = K.linearmap_apply(p, m, nan_fill=0.0)
        pushed = K.linearmap_apply(p, m, nan_fill=-1.0)
        if not K.allclose(raw, pushed, rtol=1e-5, atol=self.nan_check_threshold):
            raise ValueError(
                "NaN handling is on and results seem to depend on NaN "
                "positions in input array. Check input and standard_matrix."
            )
        return K.like_input(raw, points)

    def call_with_sumsq(self, points):
        """``(self(points), s)`` with ``s`` the device scalar sum of squares of the result, accumulated
        by the apply kernel itself (fixed order) -- or ``s = None`` when the fused form does not apply
        (slice maps, NaNs in the input, handle_nans off): then the caller reduces the result itself."""
        import torch

        shape = tuple(points.shape)
        if (len(shape) != 3 or shape[2] != self.n_dim or shape[1] != self.n_fg_sites or not self.handle_nans
                or self._onehot_index() is not None):
            return self(points), None
        out_t = K.torch_dtype(self._out_dtype(points))
        p = K.as_device(points)
        m = self._device_matrix(out_t, p.device)
        probe = K.take_flag(p.device)
        out, ss = K.linearmap_apply(p, m, want_sumsq=True, nan_probe=probe)
        if K.read_flag(probe):
            return self(points), None  # NaN policy: the reference's two extra products (rare)
        return K.like_input(out, points), ss

    def map_async(self, points):
        """Start ``self(points)`` on a side stream and return a handle whose ``result()`` gives what
        ``self(points)`` would (same NaN policy, same errors) -- or None when this map has no
        deferred form (only slice maps with NaN handling do: a gather plus a NaN scan of its output).

        Used by ``project_forces``: the coordinate map does not depend on the fitted force map, so its
        (HBM-bound) gather overlaps the fit instead of following it.
        """
        shape = tuple(points.shape)
        if len(shape) != 3 or shape[2] != self.n_dim or shape[1] != self.n_fg_sites:
            return None
        idx = self._onehot_index()
        if idx is None or not self.handle_nans:
            return None
        import torch

        out_t = K.torch_dtype(self._out_dtype(points))
        p = K.as_device(points)
        key = ("idx", str(p.device))
        hit = self._dev_cache.get(key)
        if hit is None or hit[0] is not self._standard_matrix:
            hit = (self._standard_matrix, torch.from_numpy(idx).to(p.device))
            self._dev_cache[key] = hit
        main = torch.cuda.current_stream(p.device)
        side = K.side_stream(p.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            flag = K.take_flag(p.device)
            out = K.slice_gather(p, hit[1], out_t, nan_probe=flag)
        p.record_stream(side)
        return _PendingMap(out, flag, side, points)

    def flat_call(self, flattened):
        """Apply to (n_frames, n_fg_sites*3) and return (n_frames, n_cg_sites*3)."""
        shape = tuple(flattened.shape)
        if len(shape) != 2:
            raise ValueError(f"Expected array of rank 2; got array with shape {shape}.")
        if shape[1] % self.n_dim != 0:
            raise ValueError(f"Array of shape {shape} can't be reshaped with dim of {self.n_dim}.")
        mapped = self(flattened.reshape((shape[0], shape[1] // self.n_dim, self.n_dim)))
        return mapped.reshape((mapped.shape[0], mapped.shape[1] * mapped.shape[2]))

    # ------------------------------------------------------------------ algebra
    def _derive(self, matrix: np.ndarray) -> "LinearMap":
        return self.__class__(
            mapping=matrix, handle_nans=self.handle_nans, nan_check_threshold=self.nan_check_threshold
        )

    @property
    def T(self) -> "LinearMap":
        return self._derive(self.standard_matrix.T)

    def __matmul__(self, lm: "LinearMap", /) -> "LinearMap":
        return self._derive(self.standard_matrix @ lm.standard_matrix)

    def __rmul__(self, c: float, /) -> "LinearMap":
        return self._derive(c * self.standard_matrix)

    def __add__(self, lm: "LinearMap", /) -> "LinearMap":
        return self._derive(self.standard_matrix + lm.standard_matrix)

    def astype(self, *args, **kwargs) -> "LinearMap":
        """New map whose standard_matrix is ``standard_matrix.astype(*args, **kwargs)``."""
        return self._derive(self.standard_matrix.astype(*args, **kwargs))


class CLAMap(_Taggable):
    """Co-local affine map  x_t -> A(y_t) x_t + b(y_t)  (reference map/core.py:320-430).

    ``scale(copoints)`` returns (n_steps, n_cg_sites, n_fg_sites), ``trans(copoints)`` returns
    (n_steps, n_cg_sites, 3).  Featurised force maps are of this type.  A map may carry a
    fused ``apply(points, copoints)`` callable (used by ``qp_feat_linear_map`` so that the
    per-frame matrix is never materialised); ``scale``/``trans`` remain available.
    """

    n_dim: Final = 3

    def __init__(
        self,
        scale: Callable,
        trans: Callable,
        n_fg_sites: int,
        n_cg_sites: Optional[int] = None,
        zeroes_check: bool = True,
        tags: Optional[Dict[str, str]] = None,
        apply: Optional[Callable] = None,
    ) -> None:
        super().__init__(tags=tags)
        if zeroes_check:
            z = np.zeros((1, n_fg_sites, self.n_dim))
            mapped = trjdot(z, scale(z)) + trans(z)
            if n_cg_sites is None:
                n_cg_sites = mapped.shape[1]
            elif n_cg_sites != mapped.shape[1]:
                raise ValueError("n_cg_sites did not match results from zero test")
        elif n_cg_sites is None:
            raise ValueError("If n_cg_sites is not set, zeroes_check must be truthy.")
        self._n_cg_sites: Final = n_cg_sites
        self._n_fg_sites: Final = n_fg_sites
        self.scale: Final = scale
        self.trans: Final = trans
        self._apply = apply

    @property
    def n_cg_sites(self) -> int:
        return self._n_cg_sites

    @property
    def n_fg_sites(self) -> int:
        return self._n_fg_sites

    def __call__(self, points, copoints):
        if self._apply is not None:
            return self._apply(points, copoints)
        from .. import _kernels as K

        # trjdot(points, scale) + trans (reference map/core.py:428-430).  A per-frame (3-D) scale with a full
        # (n_steps, n_cg, 3) trans is ONE pass of kernel K3c; anything else the reference's expression accepts -- a
        # frame-independent 2-D scale (K3), a scalar or broadcastable trans -- takes that expression literally.
        scale, trans = self.scale(copoints), self.trans(copoints)
        n_steps = points.shape[0]
        if (getattr(scale, "ndim", 0) == 3 and tuple(getattr(trans, "shape", ())) == (n_steps, scale.shape[1], self.n_dim)):
            out = K.trjdot_frames(K.as_device(points), K.as_device(scale), K.as_device(trans))
            return K.like_input(out, points)
        mapped = trjdot(points, scale)
        if K.is_torch(mapped):
            import torch

            if not K.is_torch(trans):
                trans = torch.as_tensor(np.asarray(trans), device=mapped.device)
            return mapped + trans.to(mapped.device)
        return mapped + (trans.cpu().numpy() if K.is_torch(trans) else trans)
