import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


# ---- launch coverage of the dispatch tables (tests/test_gpu_zz_coverage.py) ------------------------------------
# kernel (mangled handle name) -> node ids of the tests during which this process, or a child process it started,
# launched it.  The library counts every launch under its template instantiation (aggf_coverage_dump); the hooks below
# attribute the launches to the test that was running.
COVERAGE = {"by_kernel": {}, "last": {}, "children_file": None, "gpu_items_run": 0, "gpu_items_collected": 0}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import tempfile

    COVERAGE["children_file"] = os.path.join(tempfile.mkdtemp(prefix="aggf_cov_"), "children.tsv")
    os.environ["AGGF_COVERAGE_FILE"] = COVERAGE["children_file"]


def pytest_unconfigure(config):
    # (this process's own launches are read in-process; it must not append itself to the children's file at exit)
    os.environ.pop("AGGF_COVERAGE_FILE", None)
    os.environ.pop("AGGF_COVERAGE_LABEL", None)


def pytest_collection_modifyitems(config, items):
    # the coverage test judges the whole session: it runs last
    last = [it for it in items if "test_gpu_zz_coverage" in it.nodeid]
    rest = [it for it in items if "test_gpu_zz_coverage" not in it.nodeid]
    items[:] = rest + last
    COVERAGE["gpu_items_collected"] = sum(1 for it in rest if it.get_closest_marker("gpu"))


def pytest_runtest_setup(item):
    os.environ["AGGF_COVERAGE_LABEL"] = item.nodeid


def pytest_runtest_teardown(item):
    if not item.get_closest_marker("gpu") or "test_gpu_zz_coverage" in item.nodeid:
        return
    COVERAGE["gpu_items_run"] += 1
    try:
        from aggforce_amd import _lib

        now = _lib.coverage(total=True)  # (tests may reset the since-reset counters)
    except Exception:  # library not built: the tests themselves say so
        return
    for name, cnt in now.items():
        if cnt > COVERAGE["last"].get(name, 0):
            COVERAGE["by_kernel"].setdefault(name, set()).add(item.nodeid)
    COVERAGE["last"] = now


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
