"""Gaussian augmenters with closed-form log-gradients, computed on the GPU.

``CondNormal`` replaces the reference's JAX ``JCondNormal`` (trajectory/jaxgausstraj.py:
99-402) for the case the noised maps use: scalar covariance ``cov = var * I`` and a linear
premap ``M`` (``None`` = identity):
    y = M x + sqrt(var) eps,   grad_y log g = -(y - M x)/var,   grad_x log g = M'(y - M x)/var.
The JAX autodiff/vmap machinery reduces to these expressions (checked in the reference
itself against SimpleCondNormal for M = I, tests/test_simplegausstraj.py:20-29).
``SimpleCondNormal`` mirrors trajectory/simplegausstraj.py (identity premap).

Random numbers: JAX's threefry stream cannot be reproduced without JAX, so parity of the
noised path is defined conditional on the noise: tests inject ``eps``; production draws
from Philox4x32-10 keyed by (seed, global frame index, site, dim) on the device, which is
independent of how frames are sharded over GPUs.
"""
from typing import Optional, Tuple

import numpy as np

from .. import _kernels as K
from ..map.core import LinearMap
from .augment import Augmenter


class CondNormal(Augmenter):
    """y ~ N(M x, var I) with a LinearMap (or matrix) premap M."""

    n_dim = 3

    def __init__(
        self,
        var: float,
        premap=None,
        seed: Optional[int] = None,
        dtype=np.float32,
        frame_offset: int = 0,
        source_postmap=None,
    ) -> None:
        if not var > 0:
            raise ValueError("var must be positive")
        self.var = float(var)
        # linear map applied to the log-gradient with respect to the source sites (the reference's
        # JCondNormal(source_postmap=...), jaxgausstraj.py:281; used by the staged Gaussian maps)
        if source_postmap is None or isinstance(source_postmap, LinearMap):
            self.source_postmap = source_postmap
        else:
            self.source_postmap = LinearMap(np.asarray(source_postmap), handle_nans=False)
        if premap is None or isinstance(premap, LinearMap):
            self.premap = premap
        else:
            self.premap = LinearMap(np.asarray(premap), handle_nans=False)
        self.seed = int(np.random.default_rng().integers(0, int(1e6))) if seed is None else int(seed)
        self.dtype = np.dtype(dtype)
        self.frame_offset = int(frame_offset)
        self._calls = 0       # every sample()/augment call uses a fresh Philox stream offset
        self._noise_queue = []  # injected standard-normal noise (tests / reproducibility)

    # ---- noise injection ------------------------------------------------------------
    def inject_noise(self, *eps) -> "CondNormal":
        """Queue standard-normal arrays (n_frames, n_generated, 3) used by the next draws."""
        self._noise_queue.extend(eps)
        return self

    def _matrix(self, n_src: int) -> np.ndarray:
        if self.premap is None:
            return np.eye(n_src, dtype=self.dtype)
        return self.premap.standard_matrix.astype(self.dtype, copy=False)

    def _correction_matrix(self, M: np.ndarray) -> np.ndarray:
        """Matrix whose columns give the source-force correction: M, or M Q' with a source_postmap Q
        (d/dsource after the postmap = Q M' r = (M Q')' r)."""
        if self.source_postmap is None:
            return M
        Q = self.source_postmap.standard_matrix.astype(self.dtype, copy=False)
        return (M @ Q.T).astype(self.dtype, copy=False)

    def _next_noise(self, device):
        if self._noise_queue:
            return K.as_device(self._noise_queue.pop(0), K.torch_dtype(self.dtype))
        return None

    def _mean(self, src, m_dev, M: np.ndarray):
        import torch

        tdt = K.torch_dtype(self.dtype)
        # Inside one project_forces call (K.upload_cache) the fit and the application of a noised map ask for the mean
        # of the SAME coordinates: the second pass over them (a strided gather, 1.8 ms at BASELINE config 5) is saved.
        # The consumers only read the mean.
        cache = K._cache_stack[-1] if K._cache_stack else None
        anchor = self.premap._standard_matrix if self.premap is not None else None  # (M may be a fresh cast of it)
        key = ("condnormal_mean", src.data_ptr(), tuple(src.shape), str(src.dtype), id(anchor), str(tdt))
        if cache is not None:
            hit = cache.get(key)
            if hit is not None and hit[0] is src and hit[1] is anchor:
                return hit[2]
        if self.premap is not None and self.premap._onehot_index() is not None:
            idx = torch.from_numpy(self.premap._onehot_index()).to(src.device)
            mean = K.slice_gather(src, idx, tdt)
        else:
            mean = K.linearmap_apply(src, m_dev)
        if cache is not None:
            cache[key] = (src, anchor, mean)  # (the entry keeps both alive: the ids in the key cannot be reused)
        return mean

    # ---- Augmenter interface ----------------------------------------------------------
    def augment_trajectory(self, coords, forces, kbt: float) -> Tuple:
        """Fused K5 pass: returns ([x; y], [F + kbt M' r; -kbt r]) with r = (y - M x)/var."""
        import torch

        c = K.as_device(coords)
        f = K.as_device(forces, c.dtype)
        M = self._matrix(c.shape[1])
        m_dev = torch.from_numpy(np.ascontiguousarray(M)).to(c.device)
        mean = self._mean(c, m_dev, M)
        noise = self._next_noise(c.device)
        stream_seed = self.seed + 0x9E3779B97F4A7C15 * self._calls
        self._calls += 1
        cols = K.premap_columns(self._correction_matrix(M), K.torch_dtype(self.dtype), c.device)
        oc, of = K.condnormal_augment(c, f, cols, M.shape[0], mean, self.var, kbt, noise, stream_seed,
                                      self.frame_offset)
        return K.like_input(oc, coords), K.like_input(of, coords)

    def noise_sites(self, coords, kbt: float):
        """(y, Fa, C columns): the generated sites' coordinates and forces -kbt r as device arrays (n_frames,
        n_generated, 3) and the correction matrix C (compressed columns, float64) with which the extended forces
        are [F - Fa C | Fa] -- everything of ``augment_trajectory`` except the copies (same noise, same stream
        bookkeeping).  Used by the noised maps to fit and apply without the (n_frames, N + n_generated, 3) arrays."""
        import torch

        c = K.as_device(coords)
        M = self._matrix(c.shape[1])
        m_dev = torch.from_numpy(np.ascontiguousarray(M)).to(c.device)
        mean = self._mean(c, m_dev, M)
        noise = self._next_noise(c.device)
        stream_seed = self.seed + 0x9E3779B97F4A7C15 * self._calls
        self._calls += 1
        out_dtype = torch.promote_types(c.dtype, K.torch_dtype(self.dtype))
        y, fa = K.condnormal_sites(mean, self.var, kbt, noise, stream_seed, self.frame_offset, out_dtype)
        cols = K.premap_columns(self._correction_matrix(M).astype(np.float64), torch.float64, c.device)
        return y, fa, cols

    def sample(self, source):
        import torch

        c = K.as_device(source)
        M = self._matrix(c.shape[1])
        m_dev = torch.from_numpy(np.ascontiguousarray(M)).to(c.device)
        zeros = torch.zeros_like(c)
        mean = self._mean(c, m_dev, M)
        noise = self._next_noise(c.device)
        stream_seed = self.seed + 0x9E3779B97F4A7C15 * self._calls
        self._calls += 1
        cols = K.premap_columns(M, K.torch_dtype(self.dtype), c.device)
        oc, _ = K.condnormal_augment(c, zeros, cols, M.shape[0], mean, self.var, 0.0, noise, stream_seed,
                                     self.frame_offset)
        return K.like_input(oc[:, c.shape[1]:, :].to(K.torch_dtype(self.dtype)).contiguous(), source)

    def log_gradient(self, source, generated) -> Tuple:
        import torch

        tdt = K.torch_dtype(self.dtype)
        c = K.as_device(source)
        g = K.as_device(generated, tdt)
        M = self._matrix(c.shape[1])
        m_dev = torch.from_numpy(np.ascontiguousarray(M)).to(c.device)
        mean = self._mean(c, m_dev, M)
        r = (g - mean) / self.var
        mt = torch.from_numpy(np.ascontiguousarray(self._correction_matrix(M).T)).to(c.device)
        d_src = K.linearmap_apply(r.contiguous(), mt)
        return K.like_input(d_src, source), K.like_input(-r, source)

    def astype(self, dtype, *args, **kwargs) -> "CondNormal":  # noqa: ARG002
        new = self.__class__(var=self.var, premap=self.premap, seed=self.seed, dtype=dtype,
                             frame_offset=self.frame_offset, source_postmap=self.source_postmap)
        new._calls = self._calls
        new._noise_queue = list(self._noise_queue)
        return new

    def to_SimpleCondNormal(self) -> "SimpleCondNormal":
        if self.premap is not None and not self.premap.close_to_identity():
            raise ValueError("Only can convert to SimpleCondNormal for identity premap.")
        if self.source_postmap is not None and not self.source_postmap.close_to_identity():
            raise ValueError("Only can convert to SimpleCondNormal for identity source_postmap.")
        return SimpleCondNormal(var=self.var, dtype=self.dtype)


class SimpleCondNormal(CondNormal):
    """Identity-premap special case (reference trajectory/simplegausstraj.py:13-137)."""

    def __init__(self, var: float, seed: Optional[int] = None, dtype=np.float32) -> None:
        super().__init__(var=var, premap=None, seed=seed, dtype=dtype)

    def astype(self, dtype, *args, **kwargs) -> "SimpleCondNormal":  # noqa: ARG002
        return self.__class__(var=self.var, dtype=dtype)
