"""Second, independent derivation of gb_feat's features and divergences: tests/golden/g7_gbfeat_autodiff.npz.

TEST INFRASTRUCTURE (build container only; needs /root/reference for id_feat's labels).

``gb_feat`` (reference qp/jaxfeat.py) needs JAX, which is absent here, so its VALUES are
"parity unpinned" against the reference.  The oracle (oracle/aggforce_oracle.py:gb_feat_site)
and the HIP kernels K4 both use a hand-derived closed form of the divergence.  This script
does what the reference does instead of what we derived: it transcribes the forward pass
operation by operation with torch (CPU, float32 like JAX's default) --

    trjdot(points, smear_mat)                       jaxfeat.py:449-450, jaxutil.py:55-59
    distances(xyz=points, cross_xyz=cg_points)      jaxfeat.py:451, jaxutil.py:171,177
    gaussian_dist_basis / clipped_gauss             jaxfeat.py:235-240, 272-276
    channel_allocate (both array layouts)           jaxfeat.py:340-368
    gb_subfeat (collapse / channelize switches)     jaxfeat.py:441-464

-- and obtains the divergence by AUTOMATIC differentiation (torch.autograd.functional.jacobian),
following gb_subfeat_jac's two methods line by line (jaxfeat.py:529-565: "basic" = Jacobian of the
channelised collapsed features; "reorder" = Jacobian before channelising, then channel_allocate with
jac_shape=True), including the static out-of-range slice of the last label that JAX's scatter skips
(max_channels = max(ids), jaxfeat.py:115; SURVEY 3.3 Quirk A).  Labels come from the reference's own
id_feat (importable).  The two methods are checked against each other here; the fixture then pins
the oracle (tests/test_oracle_golden.py) and the kernels (tests/test_gpu_feat.py).

Still not JAX output -- but no longer a single derivation checked only against itself.

    python oracle/gen_g7_autodiff.py
"""
import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
if not os.path.isdir(os.path.join(REF, "src", "aggforce")):
    raise SystemExit("gen_g7_autodiff.py needs the reference mounted at /root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle", "standin"))
sys.path.insert(0, os.path.join(REF, "src"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from aggforce import LinearMap  # noqa: E402
from aggforce.constraints import reduce_constraint_sets  # noqa: E402
from aggforce.map import smear_map  # noqa: E402
from aggforce.qp import id_feat  # noqa: E402

from oracle import aggforce_oracle as orc  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
F32 = torch.float32


def trjdot(points, factor):
    return torch.einsum("tfd,cf->tcd", points, factor)


def distances(xyz, cross_xyz):
    displacement_matrix = xyz[:, None, :, :] - cross_xyz[:, :, None, :]
    return torch.linalg.norm(displacement_matrix, dim=-1)


def clipped_gauss(inp, center, width, clip):
    gauss = torch.exp(-(((inp - center) / width) ** 2))
    return torch.clamp(gauss, min=clip) - clip


def gaussian_dist_basis(dists, outer, inner, n_basis, width, dist_power, clip=1e-3):
    pow_grid_points = torch.linspace(inner**dist_power, outer**dist_power, n_basis, dtype=F32)
    grid_points = pow_grid_points ** (1 / dist_power)
    feats = [clipped_gauss(dists, o, width, clip) for o in grid_points]
    return torch.stack(feats, dim=-1)


def _set_slice(base, axis, start, stop, values):
    """base.at[..., start:stop, ...].set(values): a static slice is clamped to the axis; an empty
    slice leaves the array untouched (JAX skips scatters whose slice shape is empty)."""
    size = base.shape[axis]
    lo, hi = min(start, size), min(stop, size)
    if hi - lo <= 0:
        return base
    out = base.clone()
    index = [slice(None)] * base.dim()
    index[axis] = slice(lo, hi)
    out[tuple(index)] = values
    return out


def channel_allocate(feats, channels, max_channels, jac_shape=False):
    if jac_shape:
        n_feats, n_frames, _, n_dim = feats.shape
        base = torch.zeros((n_feats * max_channels, n_frames, n_dim), dtype=feats.dtype)
        per_site = [_set_slice(base, 0, n_feats * ch, n_feats * (ch + 1), feats[:, :, site, :])
                    for site, ch in enumerate(channels)]
        return torch.stack(per_site, 2)
    n_frames, _, n_feats = feats.shape
    base = torch.zeros((n_frames, n_feats * max_channels), dtype=feats.dtype)
    per_site = [_set_slice(base, 1, n_feats * ch, n_feats * (ch + 1), feats[:, site, :])
                for site, ch in enumerate(channels)]
    return torch.stack(per_site, 1)


def gb_subfeat(points, cg_points, channels, max_channels, smear_mat, collapse=False, channelize=True, **kw):
    points = trjdot(points, smear_mat)
    dists = distances(points, cg_points)
    gauss = gaussian_dist_basis(dists, **kw)[:, 0, :, :]
    out = channel_allocate(gauss, channels, max_channels) if channelize else gauss
    return out.sum(dim=(0, 1)) if collapse else out


def gb_subfeat_jac(points, cg_points, channels, max_channels, smear_mat, method, **kw):
    if method == "basic":
        jac = torch.autograd.functional.jacobian(
            lambda x: gb_subfeat(x, cg_points, channels, max_channels, smear_mat, collapse=True, **kw), points)
        return torch.swapaxes(jac.sum(dim=2), 0, 1)
    jac = torch.autograd.functional.jacobian(
        lambda x: gb_subfeat(x, cg_points, channels, max_channels, smear_mat, collapse=True, channelize=False, **kw),
        points)
    ch_jac = channel_allocate(jac, channels, max_channels, jac_shape=True)
    return torch.swapaxes(ch_jac.sum(dim=2), 0, 1)


def ordered(members):
    width = max([len(m) for m in members], default=1)
    arr = -np.ones((len(members), max(width, 1)), dtype=np.int64)
    for i, m in enumerate(members):
        arr[i, : len(m)] = m
    return arr


def build_cons(arr):
    cons = set()
    for row in arr:
        cons.add(frozenset(int(x) for x in row if x >= 0))
    return cons


def run_case(name, coords, cmat, members, kw, out):
    arr = ordered(members)
    cons = build_cons(arr)
    N = coords.shape[1]
    cmap = LinearMap(cmat)
    ids = id_feat(coords, cmap, cons, return_ids=True)          # the reference's labels
    channels = tuple(int(i) for i in ids)
    max_channels = max(channels)                                # jaxfeat.py:115
    reduced = reduce_constraint_sets(cons)
    smear = smear_map(site_groups=reduced, n_sites=N, return_mapping_matrix=True)
    pts = torch.from_numpy(coords.astype(np.float32))
    cg = torch.from_numpy(np.asarray(cmap(coords.astype(np.float32)), dtype=np.float32))
    sm = torch.from_numpy(smear)
    feats, d_re, d_ba = [], [], []
    for c in range(cmat.shape[0]):
        site = cg[:, c:c + 1, :]
        f = gb_subfeat(pts, site, channels, max_channels, sm, **kw)
        dr = gb_subfeat_jac(pts, site, channels, max_channels, sm, "reorder", **kw)
        db = gb_subfeat_jac(pts, site, channels, max_channels, sm, "basic", **kw)
        assert torch.isfinite(f).all() and torch.isfinite(dr).all(), name
        assert float((dr - db).abs().max()) < 2e-5 * max(1.0, float(dr.abs().max())), (name, "reorder vs basic")
        feats.append(f.numpy())
        d_re.append(dr.numpy())
        d_ba.append(db.numpy())
        # the oracle's closed form, right here
        of, od = orc.gb_feat_site(coords, cg[:, c, :].numpy(), ids, smear, **kw)
        ef = float(np.max(np.abs(of - feats[-1])))
        ed = float(np.max(np.abs(od - d_re[-1])))
        assert of.shape == feats[-1].shape and od.shape == d_re[-1].shape
        assert ef < 5e-6 and ed < 5e-5, (name, c, ef, ed)
    n_live = int((np.abs(np.stack(d_re)) > 0).sum())
    print(f"  {name:28s} N={N:3d} labels={max_channels + 1:3d} n_feat={feats[0].shape[2]:4d} "
          f"nonzero div entries={n_live}")
    out[f"{name}__coords"] = coords.astype(np.float32)
    out[f"{name}__cmat"] = cmat
    out[f"{name}__cons"] = arr
    out[f"{name}__ids"] = ids
    out[f"{name}__kw"] = np.array([kw["outer"], kw["inner"], kw["n_basis"], kw["width"], kw["dist_power"]])
    out[f"{name}__feats"] = np.stack(feats)
    out[f"{name}__divs"] = np.stack(d_re)
    out[f"{name}__divs_basic"] = np.stack(d_ba)


def main():
    print("G7 gb_feat by automatic differentiation (torch, float32)")
    rng = np.random.default_rng(42100 + 7)
    out = {}
    names = []
    # 1: multi-atom constraint groups (sizes 2, 3, 4), two-atom cg sites
    T, N = 6, 16
    coords = 6 * rng.random((T, N, 3)) + 1
    members = [[1, 2], [4, 5], [5, 6], [8, 9], [9, 10], [10, 11], [13, 15]]
    cmat = orc.list_mapping_matrix([[0, 1], [4, 7], [12, 13]], N)
    run_case("groups", coords, cmat, members, dict(outer=8.0, inner=0.0, n_basis=5, width=1.0, dist_power=0.5), out)
    names.append("groups")
    # 2: no constraints at all, dist_power 1, inner > 0, narrow Gaussians
    coords = 5 * rng.random((4, 9, 3)) + 1
    cmat = np.zeros((2, 9))
    cmat[0, [0, 3]] = [0.4, 0.6]
    cmat[1, [5, 8]] = [0.5, 0.5]
    run_case("free", coords, cmat, [], dict(outer=6.0, inner=1.5, n_basis=3, width=0.7, dist_power=1.0), out)
    names.append("free")
    # 3: the clip boundary.  exp(-((r-c)/w)^2) == 1e-3 at |r - c| = w sqrt(ln 1000) = 2.6283 w.  One atom is
    #    placed just inside and one just outside that radius for centre c_0 = inner = 2.0, along x from the site.
    w = 1.0
    edge = w * np.sqrt(np.log(1000.0))
    site_pos = np.array([5.0, 5.0, 5.0])
    coords = np.zeros((3, 6, 3))
    coords[:, 0] = site_pos + [0.3, 0.0, 0.0]           # site = midpoint of atoms 0 and 1
    coords[:, 1] = site_pos - [0.3, 0.0, 0.0]
    for t, eps in enumerate((1e-2, 1e-3, 5e-2)):
        coords[t, 2] = site_pos + [2.0 + edge - eps, 0.0, 0.0]   # inside: tiny but non-zero feature and gradient
        coords[t, 3] = site_pos + [2.0 + edge + eps, 0.0, 0.0]   # outside: exactly zero, zero gradient
        coords[t, 4] = site_pos + [0.0, 2.0, 0.0]                # on the centre: gradient zero, feature 1 - clip
        coords[t, 5] = site_pos + [0.0, 0.0, 2.0 - edge + eps] if 2.0 - edge + eps > 0 else site_pos + [0.0, 0.0, 0.05]
    cmat = np.zeros((1, 6))
    cmat[0, [0, 1]] = 0.5
    run_case("clip_edge", coords, cmat, [[4, 5]], dict(outer=7.0, inner=2.0, n_basis=2, width=w, dist_power=1.0), out)
    names.append("clip_edge")
    # 4: slice map whose site atom is constrained to a neighbour (CLN025's CA-HA situation: the smeared
    #    position differs from the site, r > 0), chain of three, the C4 bench's pair pattern
    T, N = 5, 15
    coords = 4 * rng.random((T, N, 3)) + 2
    members = [[3 * i, 3 * i + 1] for i in range(5)]
    cmat = orc.list_mapping_matrix([[0], [6], [12]], N)
    run_case("slice_pairs", coords, cmat, members, dict(outer=8.0, inner=0.0, n_basis=8, width=1.0, dist_power=0.5), out)
    names.append("slice_pairs")
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "g7_gbfeat_autodiff.npz"), **out)
    print("written", os.path.join(OUT, "g7_gbfeat_autodiff.npz"))


if __name__ == "__main__":
    main()
