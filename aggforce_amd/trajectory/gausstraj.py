"""Gaussian augmenters with closed-form log-gradients, computed on the GPU.

``CondNormal`` (also exported as ``JCondNormal``) replaces the reference's JAX ``JCondNormal``
(trajectory/jaxgausstraj.py:99-402) behind the same constructor -- ``(cov, premap=None,
source_postmap=None, seed=None, dtype=<unset>)``, ``cov`` a scalar variance or a full
``(3 n, 3 n)`` covariance matrix, ``premap`` a callable on FLATTENED ``(n_frames, 3 N)`` arrays
(``LinearMap.flat_call`` in every call of the reference, qp/jgauss.py:114-118,237,385,552),
``source_postmap`` a callable on ``(n_frames, N, 3)`` arrays (a ``LinearMap``,
qp/jgauss.py:282-286) -- and the same methods / attributes (``sample``, ``log_gradient``,
``astype``, ``to_SimpleCondNormal``, ``premap``, ``source_postmap``, ``cov``, ``dtype``).
With ``y = M x + L eps`` (``L L' = cov``) the JAX autodiff / vmap machinery is
    grad_y log g = -cov^-1 (y - M x),      grad_x log g = M' cov^-1 (y - M x)
(checked in the reference itself against SimpleCondNormal for M = I, tests/test_simplegausstraj.py:
20-29).  The maps must be LINEAR and act on sites, not on Cartesian components (``M (x) I_3``): a
``LinearMap`` / its bound ``flat_call`` / a matrix is taken as it is, any other callable is
probed once with unit displacements (and refused with ``ValueError`` if it is not of that form).
``SimpleCondNormal`` mirrors trajectory/simplegausstraj.py (identity premap).

Random numbers: JAX's threefry stream cannot be reproduced without JAX, so parity of the
noised path is defined conditional on the noise: tests inject ``eps``; production draws
from Philox4x32-10 keyed by (seed, global frame index, site, dim) on the device, which is
independent of how frames are sharded over GPUs.
"""
from typing import Callable, Final, Optional, Tuple, TypeVar

import numpy as np

from .. import _kernels as K
from ..map.core import LinearMap
from .augment import Augmenter

_UNSET: Final = object()

A = TypeVar("A")


def _ident(x: A, /) -> A:
    """Identity (the reference's default premap / source_postmap, jaxgausstraj.py:20-22)."""
    return x


def _to_numpy(x) -> np.ndarray:
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float64)


def _known_linear(obj):
    """The LinearMap behind ``obj`` without calling it -- None for the identity, ``_UNSET`` if it has to be probed."""
    if obj is None or obj is _ident:
        return None
    if isinstance(obj, LinearMap):
        return obj
    owner = getattr(obj, "__self__", None)  # bound method: LinearMap.flat_call / __call__ (ours or the reference's)
    if owner is not None and getattr(obj, "__name__", "") in ("flat_call", "__call__"):
        obj = owner
    if isinstance(obj, LinearMap):
        return obj
    if hasattr(obj, "standard_matrix"):
        return LinearMap(np.asarray(obj.standard_matrix), handle_nans=False)
    if hasattr(obj, "detach") or isinstance(obj, (np.ndarray, list, tuple)):
        mat = _to_numpy(obj)
        if mat.ndim != 2:
            raise ValueError(f"A map given as an array must be 2-D (n_cg, n_fg); got shape {mat.shape}.")
        return LinearMap(mat, handle_nans=False)
    if not callable(obj):
        raise ValueError(f"Cannot understand map {obj!r}: expected a LinearMap, a matrix or a callable.")
    return _UNSET


def _probe_linear(call: Callable, n_src: int, flat: bool, what: str) -> LinearMap:
    """Standard matrix of a callable that acts linearly on the site axis.

    ``flat``: the callable takes (n_frames, 3 n_src) and returns (n_frames, 3 n) (a premap); else it takes
    (n_frames, n_src, 3) and returns (n_frames, n, 3) (a source_postmap).  Unit displacements along x of every site give
    the columns; the y / z components of those images, the image of 0 and three random frames check that the callable
    really is ``x -> M x`` applied to every Cartesian component alike."""

    def run(frames: np.ndarray) -> np.ndarray:
        arg = frames.reshape(frames.shape[0], -1) if flat else frames
        out = _to_numpy(call(arg))
        if out.ndim != (2 if flat else 3) or out.shape[0] != frames.shape[0] or (flat and out.shape[1] % 3):
            raise ValueError(f"{what} returned an array of shape {out.shape} for input of shape {arg.shape}.")
        out = out.reshape(out.shape[0], -1, 3)
        if not flat and out.shape[2] != 3:
            raise ValueError(f"{what} returned an array of shape {out.shape}.")
        return out

    cols = []
    for s0 in range(0, n_src, 256):
        k = min(256, n_src - s0)
        basis = np.zeros((k, n_src, 3))
        basis[np.arange(k), s0 + np.arange(k), 0] = 1.0
        img = run(basis)
        if np.abs(img[:, :, 1:]).max(initial=0.0) > 1e-6 * max(1.0, np.abs(img[:, :, 0]).max(initial=0.0)):
            raise ValueError(f"{what} mixes Cartesian components; only maps acting on sites (M (x) I_3) are supported.")
        cols.append(img[:, :, 0].T)
    matrix = np.concatenate(cols, axis=1)  # (n, n_src)
    rng = np.random.default_rng(42100)
    test = np.concatenate([np.zeros((1, n_src, 3)), rng.standard_normal((3, n_src, 3))])
    want = np.einsum("cs,tsd->tcd", matrix, test)
    got = run(test)
    if got.shape != want.shape or np.abs(got - want).max(initial=0.0) > 1e-5 * max(1.0, np.abs(want).max(initial=0.0)):
        raise ValueError(f"{what} is not a linear map acting on sites (affine or non-linear callables are not supported).")
    return LinearMap(matrix, handle_nans=False)


class CondNormal(Augmenter):
    r"""Augmenter ``g(y|x) \propto exp[-(y - Ax)' E^-1 (y - Ax) / 2]``: Gaussian noise around mapped positions.

    Arguments (names, order and defaults of jaxgausstraj.py:140-146):
      cov             scalar variance (E = cov I) or a full (3 n, 3 n) covariance matrix over the flattened generated
                      coordinates (site-major, xyz innermost);
      premap          A: None (identity), LinearMap, (n, N) matrix, or a callable on flattened (n_frames, 3 N) arrays;
      source_postmap  applied to the log-gradient with respect to the source sites, an (n_frames, N, 3) array: None,
                      LinearMap, matrix or callable;
      seed            noise seed (None: drawn at random);
      dtype           dtype of the outputs; default: that of an array ``cov``, else float32 (None means float64, as
                      np.dtype(None) does in the reference).
    Keyword-only extras: ``frame_offset`` (global index of this shard's first frame: any sharding of the frames draws
    the same noise), ``var`` (alias of a scalar ``cov``).  Unlike the reference, ``log_gradient`` works before the
    first ``sample`` call also for a scalar ``cov`` (the reference needs ``sample`` to learn the dimension).
    """

    n_dim: Final = 3

    def __init__(
        self,
        cov=_UNSET,
        premap: Optional[Callable] = None,
        source_postmap: Optional[Callable] = None,
        seed: Optional[int] = None,
        dtype=_UNSET,
        *,
        frame_offset: int = 0,
        var=_UNSET,
    ) -> None:
        if (cov is _UNSET) == (var is _UNSET):
            raise TypeError("CondNormal needs the covariance: exactly one of cov (reference name) or var.")
        if cov is _UNSET:
            cov = var
        self._cov = cov  # as given (reference attribute)
        cov_arr = cov.detach().cpu().numpy() if hasattr(cov, "detach") else np.asarray(cov)
        if cov_arr.ndim == 0:
            if not float(cov_arr) > 0:
                raise ValueError("cov (variance) must be positive")
            self.var: Optional[float] = float(cov_arr)
            self._cov_matrix = None
        elif cov_arr.ndim == 2 and cov_arr.shape[0] == cov_arr.shape[1] and cov_arr.shape[0] % self.n_dim == 0:
            full = np.asarray(cov_arr, dtype=np.float64)
            if not np.allclose(full, full.T, rtol=1e-6, atol=1e-12 * max(1.0, np.abs(full).max())):
                raise ValueError("cov must be a symmetric matrix")
            try:
                self._chol = np.linalg.cholesky(full)  # y = mean + L eps  (map-sized host algebra)
            except np.linalg.LinAlgError as err:
                raise ValueError("cov must be positive definite") from err
            self._prec = np.linalg.inv(full)
            self._prec = 0.5 * (self._prec + self._prec.T)
            self.var = None
            self._cov_matrix = full
        else:
            raise ValueError(f"cov must be a scalar or a square (3 n, 3 n) matrix; got shape {cov_arr.shape}.")
        self.premap: Callable = _ident if premap is None else premap
        self.source_postmap: Callable = _ident if source_postmap is None else source_postmap
        self._pre = _known_linear(premap)          # LinearMap | None (identity) | _UNSET (probe on first use)
        self._post = _known_linear(source_postmap)
        self.seed = int(np.random.default_rng().integers(0, int(1e6))) if seed is None else int(seed)
        if dtype is _UNSET:
            self.dtype = np.dtype(cov_arr.dtype) if cov_arr.ndim == 2 and cov_arr.dtype in (np.float32, np.float64) \
                else np.dtype(np.float32)
        else:
            self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError(f"dtype must be float32 or float64; got {self.dtype}.")
        self.frame_offset = int(frame_offset)
        self._n_gen: Optional[int] = None if self._cov_matrix is None else self._cov_matrix.shape[0] // self.n_dim
        self._calls = 0       # every sample()/augment call uses a fresh Philox stream offset
        self._noise_queue = []  # injected standard-normal noise (tests / reproducibility)

    # ---- the maps ----------------------------------------------------------------------
    def premap_map(self, n_src: Optional[int] = None) -> Optional[LinearMap]:
        """The premap as a LinearMap (None = identity).  A callable that is not a LinearMap method is turned into its
        matrix on first use, which needs the number of source sites."""
        if self._pre is _UNSET:
            if n_src is None:
                raise ValueError("The premap is a callable whose matrix is not known yet: pass the number of source "
                                 "sites, or call sample / log_gradient first.")
            self._pre = _probe_linear(self.premap, n_src, True, "premap")
        return self._pre

    def source_postmap_map(self, n_src: Optional[int] = None) -> Optional[LinearMap]:
        if self._post is _UNSET:
            if n_src is None:
                raise ValueError("The source_postmap is a callable whose matrix is not known yet.")
            self._post = _probe_linear(self.source_postmap, n_src, False, "source_postmap")
        return self._post

    @property
    def cov(self) -> Optional[np.ndarray]:
        """Covariance matrix (3 n, 3 n); None for a scalar ``cov`` until the dimension has been seen (as in the
        reference, where the first ``sample`` call creates it: jaxgausstraj.py:306-309)."""
        if self._cov_matrix is not None:
            return self._cov_matrix.astype(self.dtype, copy=False)
        if self._n_gen is None:
            return None
        return np.diag(np.full(self.n_dim * self._n_gen, self.var, dtype=self.dtype))

    # ---- noise injection ------------------------------------------------------------
    def inject_noise(self, *eps) -> "CondNormal":
        """Queue standard-normal arrays (n_frames, n_generated, 3) used by the next draws."""
        self._noise_queue.extend(eps)
        return self

    def _matrix(self, n_src: int) -> np.ndarray:
        pre = self.premap_map(n_src)
        if pre is None:
            M = np.eye(n_src, dtype=self.dtype)
        else:
            if pre.n_fg_sites != n_src:
                raise ValueError(f"premap acts on {pre.n_fg_sites} sites, the source has {n_src}.")
            M = pre.standard_matrix.astype(self.dtype, copy=False)
        if self._n_gen is None:
            self._n_gen = M.shape[0]
        elif self._n_gen != M.shape[0]:
            raise ValueError(f"cov is over {self._n_gen} generated sites, the premap produces {M.shape[0]}.")
        return M

    def _correction_matrix(self, M: np.ndarray) -> np.ndarray:
        """Matrix whose columns give the source-force correction: M, or M Q' with a source_postmap Q
        (d/dsource after the postmap = Q M' r = (M Q')' r)."""
        post = self.source_postmap_map(M.shape[1])
        if post is None:
            return M
        Q = post.standard_matrix.astype(self.dtype, copy=False)
        if Q.shape[1] != M.shape[1]:
            raise ValueError(f"source_postmap acts on {Q.shape[1]} sites, the source has {M.shape[1]}.")
        return (M @ Q.T).astype(self.dtype, copy=False)

    def correction_columns(self, n_src: int, device):
        """Compressed columns (float64) of the matrix C with which the extended forces are [F - Fa C | Fa]."""
        import torch

        return K.premap_columns(self._correction_matrix(self._matrix(n_src)).astype(np.float64), torch.float64, device)

    def _next_noise(self, device):
        if self._noise_queue:
            return K.as_device(self._noise_queue.pop(0), K.torch_dtype(self.dtype))
        return None

    def _next_stream(self) -> int:
        stream_seed = self.seed + 0x9E3779B97F4A7C15 * self._calls
        self._calls += 1
        return stream_seed

    def _mean(self, src, m_dev, M: np.ndarray):
        import torch

        tdt = K.torch_dtype(self.dtype)
        pre = self.premap_map(src.shape[1])
        # Inside one project_forces call (K.upload_cache) the fit and the application of a noised map ask for the mean
        # of the SAME coordinates: the second pass over them (a strided gather, 1.8 ms at BASELINE config 5) is saved.
        # The consumers only read the mean.
        cache = K._cache_stack[-1] if K._cache_stack else None
        anchor = pre._standard_matrix if pre is not None else None  # (M may be a fresh cast of it)
        key = ("condnormal_mean", src.data_ptr(), tuple(src.shape), str(src.dtype), id(anchor), str(tdt))
        if cache is not None:
            hit = cache.get(key)
            if hit is not None and hit[0] is src and hit[1] is anchor:
                return hit[2]
        if pre is not None and pre._onehot_index() is not None:
            idx = torch.from_numpy(pre._onehot_index()).to(src.device)
            mean = K.slice_gather(src, idx, tdt)
        else:
            mean = K.linearmap_apply(src, m_dev)
        if cache is not None:
            cache[key] = (src, anchor, mean)  # (the entry keeps both alive: the ids in the key cannot be reused)
        return mean

    def _device_mean(self, coords):
        import torch

        c = K.as_device(coords)
        if c.dim() != 3 or c.shape[2] != self.n_dim:
            raise ValueError(f"Expected an array of shape (n_frames, n_sites, {self.n_dim}); got {tuple(c.shape)}.")
        M = self._matrix(c.shape[1])
        m_dev = torch.from_numpy(np.ascontiguousarray(M)).to(c.device)
        return c, M, self._mean(c, m_dev, M)

    # ---- full covariance: flattened (n_frames, 3 n) products (aggf_frames_matmul) ----------------
    def _full_sample(self, mean, noise):
        """y = mean + eps L' (jaxgausstraj.py:311-316: multivariate_normal(mean, cov))."""
        import torch

        T, n, _ = mean.shape
        tdt = K.torch_dtype(self.dtype)
        stream_seed = self._next_stream()
        if noise is None:
            noise = K.synth_normal(T, n, tdt, stream_seed & (2**64 - 1), self.frame_offset, device=mean.device)
        L = torch.from_numpy(np.ascontiguousarray(self._chol.astype(self.dtype))).to(mean.device)
        y = K.frames_matmul(noise.reshape(T, 3 * n), L, add=mean.reshape(T, 3 * n))
        return y.reshape(T, n, 3)

    def _full_neg_residual(self, gen, mean, scale: float = 1.0):
        """-scale cov^-1 (gen - mean): the log-gradient with respect to the generated sites (times ``scale``)."""
        import torch

        T, n, _ = mean.shape
        P = torch.from_numpy(np.ascontiguousarray(self._prec.astype(self.dtype))).to(mean.device)
        out = K.frames_matmul(gen.reshape(T, 3 * n), P, sub=mean.reshape(T, 3 * n), alpha=-float(scale))
        return out.reshape(T, n, 3)

    # ---- Augmenter interface ----------------------------------------------------------
    def augment_trajectory(self, coords, forces, kbt: float) -> Tuple:
        """Fused K5 pass: returns ([x; y], [F + kbt M' r; -kbt r]) with r = cov^-1 (y - M x)."""
        c, M, mean = self._device_mean(coords)
        f = K.as_device(forces)
        noise = self._next_noise(c.device)
        if self._cov_matrix is not None or f.dtype != c.dtype:
            # full covariance, or forces wider / narrower than the coordinates (NumPy promotion as in the reference's
            # trajectory/core.py:384-390: float64 forces are not demoted to float32 coordinates' dtype): the general
            # concatenation
            from .core import _augment_concat

            if self._cov_matrix is not None:
                y = self._full_sample(mean, noise)
            else:
                y, _ = K.condnormal_sites(mean, self.var, 0.0, noise, self._next_stream(), self.frame_offset,
                                          K.torch_dtype(self.dtype))
            d_src, d_gen = self._log_gradient_dev(c, M, mean, y)
            return _augment_concat(coords, forces, y, d_src, d_gen, kbt)
        cols = K.premap_columns(self._correction_matrix(M), K.torch_dtype(self.dtype), c.device)
        oc, of = K.condnormal_augment(c, f, cols, M.shape[0], mean, self.var, kbt, noise, self._next_stream(),
                                      self.frame_offset)
        return K.like_input(oc, coords), K.like_input(of, coords)

    def noise_sites(self, coords, kbt: float):
        """(y, Fa, C columns): the generated sites' coordinates and forces -kbt r as device arrays (n_frames,
        n_generated, 3) and the correction matrix C (compressed columns, float64) with which the extended forces
        are [F - Fa C | Fa] -- everything of ``augment_trajectory`` except the copies (same noise, same stream
        bookkeeping).  Used by the noised maps to fit and apply without the (n_frames, N + n_generated, 3) arrays."""
        import torch

        c, M, mean = self._device_mean(coords)
        noise = self._next_noise(c.device)
        out_dtype = torch.promote_types(c.dtype, K.torch_dtype(self.dtype))
        if self._cov_matrix is not None:
            y = self._full_sample(mean, noise)
            fa = self._full_neg_residual(y, mean, kbt)
            y, fa = y.to(out_dtype), fa.to(out_dtype)
        else:
            y, fa = K.condnormal_sites(mean, self.var, kbt, noise, self._next_stream(), self.frame_offset, out_dtype)
        return y, fa, self.correction_columns(c.shape[1], c.device)

    def sample(self, source):
        """Gaussian variates around the premapped ``source`` (jaxgausstraj.py:213-235): (n_frames, n, 3), self.dtype."""
        import torch

        c, M, mean = self._device_mean(source)
        noise = self._next_noise(c.device)
        if self._cov_matrix is not None:
            return K.like_input(self._full_sample(mean, noise), source)
        y, _ = K.condnormal_sites(mean, self.var, 0.0, noise, self._next_stream(), self.frame_offset,
                                  K.torch_dtype(self.dtype))
        return K.like_input(y, source)

    def _log_gradient_dev(self, c, M: np.ndarray, mean, g):
        """(d/dsource after the postmap, d/dgenerated) as device arrays in self.dtype."""
        import torch

        tdt = K.torch_dtype(self.dtype)
        if tuple(g.shape) != tuple(mean.shape):
            raise ValueError(f"generated has shape {tuple(g.shape)}; the premap implies {tuple(mean.shape)}.")
        if self._cov_matrix is not None:
            neg = self._full_neg_residual(g, mean)
        else:
            _, neg = K.residual_over_var(g, mean, self.var, tdt, want_pos=False)
        # d/dsource = C' r = (-C)' (-r): the (small) matrix carries the sign, one pass over the frames
        mt = torch.from_numpy(np.ascontiguousarray(-self._correction_matrix(M).T)).to(c.device)
        return K.linearmap_apply(neg, mt), neg

    def log_gradient(self, source, generated) -> Tuple:
        """(grad_source log g [through source_postmap], grad_generated log g), jaxgausstraj.py:237-289."""
        c, M, mean = self._device_mean(source)
        g = K.as_device(generated, K.torch_dtype(self.dtype))
        d_src, d_gen = self._log_gradient_dev(c, M, mean, g)
        return K.like_input(d_src, source), K.like_input(d_gen, source)

    def astype(self, dtype, *args, **kwargs) -> "CondNormal":  # noqa: ARG002
        """Instance with ``dtype`` (args / kwargs ignored, as in the reference); the noise stream continues."""
        new = self.__class__(cov=self._cov, premap=self.premap, source_postmap=self.source_postmap, seed=self.seed,
                             dtype=dtype, frame_offset=self.frame_offset)
        new._pre, new._post = self._pre, self._post
        new._calls = self._calls
        new._noise_queue = list(self._noise_queue)
        return new

    def to_SimpleCondNormal(self) -> "SimpleCondNormal":
        """SimpleCondNormal with the same variance: scalar cov, identity premap and source_postmap only
        (jaxgausstraj.py:379-402)."""
        if self.var is None:
            raise ValueError("Only can convert to SimpleCondNormal for scalar-specified covariance.")
        pre = self._pre if self._pre is not _UNSET else self.premap_map(self._n_src_hint())
        if pre is not None and not pre.close_to_identity():
            raise ValueError("Only can convert to SimpleCondNormal for identity premap.")
        post = self._post if self._post is not _UNSET else self.source_postmap_map(self._n_src_hint())
        if post is not None and not post.close_to_identity():
            raise ValueError("Only can convert to SimpleCondNormal for identity source_postmap.")
        return SimpleCondNormal(var=self.var, dtype=self.dtype)

    def _n_src_hint(self) -> Optional[int]:
        for m in (self._pre, self._post):
            if isinstance(m, LinearMap):
                return m.n_fg_sites
        return None


class SimpleCondNormal(CondNormal):
    """Identity-premap special case (reference trajectory/simplegausstraj.py:13-137)."""

    def __init__(self, var: float, seed: Optional[int] = None, dtype=_UNSET) -> None:
        super().__init__(var, premap=None, seed=seed, dtype=np.float32 if dtype is _UNSET else dtype)

    def astype(self, dtype, *args, **kwargs) -> "SimpleCondNormal":  # noqa: ARG002
        return self.__class__(var=self.var, dtype=dtype)
