import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


def cons_from_array(arr):
    """Inverse of gen_golden.cons_to_array: (n, w) int array padded with -1 -> set of frozensets."""
    return {frozenset(int(x) for x in row if x >= 0) for row in arr}


def cons_in_insertion_order(arr):
    """Rebuild a constraint set exactly as oracle/gen_golden.py:build_cons did (one add per row, in
    row order): the reference's id_feat label order depends on the set's hash-table layout."""
    cons = set()
    for row in arr:
        cons.add(frozenset(int(x) for x in row if x >= 0))
    return cons


def need_hbm(gib: float) -> None:
    """The full-size tests carry the evidence for BASELINE's configurations: on a device that is big enough they must
    RUN.  Cached blocks and workspaces of earlier tests are released first; if the memory still is not free, something
    leaked (or another process holds the device) and the test FAILS instead of turning into a silent skip.  Only a
    device that is too small as a whole (not an MI355X) skips."""
    import gc

    import torch

    from aggforce_amd import _lib

    gc.collect()
    _lib.free_workspaces()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if total < gib * 2**30:
        pytest.skip(f"device has {total / 2**30:.0f} GiB in all; the test needs {gib:.0f} GiB")
    if free < gib * 2**30:
        pytest.fail(f"only {free / 2**30:.0f} of {total / 2**30:.0f} GiB of HBM are free, the test needs {gib:.0f}: an earlier "
                    "test leaked device memory or another process holds the GPU")
