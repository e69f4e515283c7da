"""project_forces_grid_cv over l2_regularization of the fused featurised fit at BASELINE config 4's size: the one-pass
form (qp/gbfeat.py:cv_id_gb) against the reference's loop of fits and applications, same generators."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aggforce_amd import LinearMap, agg  # noqa: E402
from aggforce_amd.qp import Multifeaturize, gb_feat, id_feat, qp_feat_linear_map  # noqa: E402
from aggforce_amd.util import Curry  # noqa: E402

T, N, n_cg = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (20000, 1024, 64)))
from aggforce_amd import _kernels as K  # noqa: E402

# bench.py's synthetic trajectory: a 1.5 A lattice with 0.3 A of noise per frame (distances stay in their ranges, so the
# zero-column compaction keeps what it keeps at config 4)
forces = K.synth_normal(T, N, torch.float32, 1234, frame_offset=0, sigma=30.0)
coords = K.synth_normal(T, N, torch.float32, 1235, frame_offset=0, sigma=0.3, lattice=1.5)
constraints = {frozenset([3 * i, 3 * i + 1]) for i in range(N // 3)}
cmap = LinearMap([[3 * (i * (N // n_cg) // 3)] for i in range(n_cg)], n_fg_sites=N)
feat = Multifeaturize([id_feat, Curry(gb_feat, outer=8.0, inner=0.0, n_basis=8, width=1.0)])
grid = {"l2_regularization": [1.0, 10.0, 100.0]}
out = {"T": T, "N": N, "n_cg": n_cg, "grid": grid["l2_regularization"], "n_folds": 5}
res = {}
for name, reuse in (("one_pass", True), ("loop", False)):
    for rep in range(2):  # the second run is the timed one (allocator and workspaces warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res[name] = agg.project_forces_grid_cv(grid, coords, forces, n_folds=5, rng=np.random.default_rng(3),
                                               reuse_gram=reuse, method_rng=np.random.default_rng(17), coord_map=cmap,
                                               constrained_inds=constraints, method=qp_feat_linear_map, featurizer=feat,
                                               kbt=0.6955215, n_constraint_frames=20)
        torch.cuda.synchronize()
        out[name + "_s"] = time.perf_counter() - t0
out["scores_one_pass"] = [res["one_pass"]["scores"][k] for k in res["one_pass"]["scores"]]
out["scores_loop"] = [res["loop"]["scores"][k] for k in res["loop"]["scores"]]
out["max_rel_diff"] = max(abs(a - b) / abs(b) for a, b in zip(out["scores_one_pass"], out["scores_loop"]))
out["speedup"] = out["loop_s"] / out["one_pass_s"]
print(json.dumps(out))
