"""GPU: the solve and the slab sums under ADVERSARIAL DISPATCH ORDER.

Round 4 shipped a wrong-result race in K2's step kernel for most of the round: workgroup 0 overwrote a block that
late-dispatched workgroups of the same launch still read; 265 green tests could not see it, because on an idle GPU the
workgroups of a launch start together.  The class of bug -- a launch whose workgroups read and write one buffer and
whose result depends on which of them runs first -- is tested here deterministically instead of by luck: a TEST-ONLY
build of the same sources (`make -C aggforce_amd/csrc order`, -DAGGF_ORDER_TEST, never shipped) runs the workgroups of
every such launch (DESIGN.md section 5b lists them) strictly one after the other, in ascending or in descending block
order.  The solve tests -- oracle comparisons, bit-identity of batched against separate solves -- must pass under both
orders, and the two orders must agree bit for bit: the replicated solve of the multi-GPU path (qp/qplinear.py:79-86
on every rank) relies on it."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aggforce_amd", "csrc")
ORDER_LIB = os.path.join(CSRC, "build_order", "libaggf_order.so")

_BITS = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from aggforce_amd import _kernels as K
from aggforce_amd.constraints import group_layout, groups_csr
rng = np.random.default_rng(23)
h = hashlib.sha256()
for n, m in ((500, 12), (1000, 40), (130, 130)):
    R = rng.standard_normal((3 * n, n))
    G = torch.from_numpy(R.T @ R).cuda()
    A = torch.from_numpy(rng.standard_normal((m, n))).cuda()
    X, st = K.eq_qp_solve(G, 1e-3, None, A, None, schur_reg=1e-12, n_refine=2)
    h.update(X.cpu().numpy().tobytes()); h.update(st.cpu().numpy().tobytes())
    if m < n:
        pins = torch.arange(0, n, n // m, dtype=torch.int32)[:m].cuda()
        Xp, stp = K.eq_qp_solve_pinned(G, 1e-3, None, pins)
        h.update(Xp.cpu().numpy().tobytes())
    Xb, stb = K.eq_qp_solve_batched(torch.stack([G, 2.0 * G, G]), 1e-3, None, torch.stack([A, A, A]), None)
    h.update(Xb.cpu().numpy().tobytes())
    P = K.sym_pack_upper(G)
    h.update(K.sym_unpack_upper(P, torch.empty_like(G)).cpu().numpy().tobytes())
# slab sums with `accumulate` (tile kernel and streaming kernel), constraint groups
for N, T in ((300, 2051), (90, 3000), (700, 900)):
    f = torch.from_numpy(rng.standard_normal((T, N, 3))).cuda()
    cons = {frozenset([3 * i, 3 * i + 1]) for i in range(N // 3)}
    goa, n_red = group_layout(N, cons)
    p, a = groups_csr(goa, n_red)
    gp, ga = torch.from_numpy(p).cuda(), torch.from_numpy(a).cuda()
    acc = K.gram(f[: T // 3].contiguous(), gp, ga, n_red, torch.float64)
    K.gram(f[T // 3:].contiguous(), gp, ga, n_red, torch.float64, out=acc, accumulate=True)
    h.update(acc.cpu().numpy().tobytes())
print("BITS", h.hexdigest())
"""


def test_solve_and_slab_sums_under_forward_and_reverse_workgroup_order():
    build = subprocess.run(["make", "-C", CSRC, "-j", "16", "order"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert build.returncode == 0 and os.path.exists(ORDER_LIB), build.stdout.decode(errors="replace")[-3000:]
    digests = {}
    for order in ("forward", "reverse"):
        env = {k: v for k, v in os.environ.items() if k not in ("AGGF_COVERAGE_FILE", "AGGF_COVERAGE_LABEL")}
        env.update(AGGF_LIB_PATH=ORDER_LIB, AGGF_ORDER=order)
        # 1. the solve's own parity tests, run ONCE under this order (a wrong answer under either order is a bug today)
        run = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_solve_batched.py"), "-q", "-x",
                              "-m", "gpu", "-p", "no:cacheprovider"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                             cwd=ROOT, timeout=1200)
        text = run.stdout.decode(errors="replace") + run.stderr.decode(errors="replace")
        assert run.returncode == 0, f"AGGF_ORDER={order}:\n" + text[-4000:]
        summary = re.findall(r"AGGF_ORDER_SUMMARY mode=(\d) gated=(\d+) ungated=(\d+) timeouts=(\d+)", text)
        assert summary, text[-2000:]
        mode, gated, ungated, timeouts = (int(v) for v in summary[-1])
        assert mode == (1 if order == "forward" else 2) and timeouts == 0, summary
        assert gated > 500 and ungated < gated, summary  # the gate really ordered most such launches (grids beyond what is surely resident run ungated)
        # 2. bit pattern of a fixed set of solves, packed triangles and accumulating Gram builds
        bits = subprocess.run([sys.executable, "-c", _BITS, ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                              cwd=ROOT, timeout=600)
        out = bits.stdout.decode(errors="replace")
        assert bits.returncode == 0 and "BITS " in out, bits.stderr.decode(errors="replace")[-3000:]
        assert "timeouts=0" in bits.stderr.decode(errors="replace")
        digests[order] = out.split("BITS ")[1].split()[0]
    # 3. ... and with the shipped library, whose workgroups run concurrently in whatever order the dispatcher picks
    env = {k: v for k, v in os.environ.items() if k not in ("AGGF_COVERAGE_FILE", "AGGF_COVERAGE_LABEL", "AGGF_LIB_PATH")}
    plain = subprocess.run([sys.executable, "-c", _BITS, ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT,
                           timeout=600)
    assert plain.returncode == 0, plain.stderr.decode(errors="replace")[-3000:]
    digests["concurrent"] = plain.stdout.decode().split("BITS ")[1].split()[0]
    assert digests["forward"] == digests["reverse"] == digests["concurrent"], digests
