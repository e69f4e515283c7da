"""GPU: the C-ABI collective (aggf_comm_* / aggf_allreduce_sum = RCCL) -- a world of one in process, and two
processes on the same GPU box exchanging the unique id through a file (the channel a torch-less host would use)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from aggforce_amd import _lib  # noqa: E402
from conftest import ROOT  # noqa: E402


def test_packed_upper_triangle_round_trip():
    """aggf_sym_pack_upper / aggf_sym_unpack_upper (the Gram all-reduce payload): element order of the packed form,
    both triangles restored exactly, batches, sizes that are not multiples of the 32 x 32 tile."""
    from aggforce_amd import _kernels as K

    rng = np.random.default_rng(5)
    for batch, n in ((1, 1), (1, 31), (3, 97), (1, 256), (2, 1000)):
        a = rng.standard_normal((batch, n, n))
        sym = a + a.transpose(0, 2, 1)
        G = torch.from_numpy(sym).cuda()
        packed = K.sym_pack_upper(G)
        iu = np.triu_indices(n)
        assert np.array_equal(packed.cpu().numpy(), np.stack([m[iu] for m in sym]))  # row-major upper triangle
        out = torch.full_like(G, float("nan"))
        K.sym_unpack_upper(packed, out)
        assert np.array_equal(out.cpu().numpy(), sym)
        # the lower triangle of the input is never read: garbage there does not reach the result
        dirty = G.clone()
        il = np.tril_indices(n, -1)
        dirty[:, il[0], il[1]] = 7.0
        assert torch.equal(K.sym_pack_upper(dirty), packed)


def test_allreduce_world_of_one_is_identity():
    l = _lib.lib()
    uid = (C.c_char * 128)()
    _lib.check(l.aggf_comm_unique_id(uid, 128), "aggf_comm_unique_id")
    comm = C.c_void_p()
    _lib.check(l.aggf_comm_init(uid, 128, 0, 1, C.byref(comm)), "aggf_comm_init")
    x = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
    ref = x.clone()
    _lib.check(l.aggf_allreduce_sum(x.data_ptr(), x.numel(), _lib.F64, comm, _lib.stream_ptr()), "aggf_allreduce_sum")
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    y = torch.ones(7, dtype=torch.float32, device="cuda")
    _lib.check(l.aggf_allreduce_sum(y.data_ptr(), 7, _lib.F32, comm, _lib.stream_ptr()), "aggf_allreduce_sum")
    torch.cuda.synchronize()
    assert float(y.sum()) == 7.0
    assert l.aggf_allreduce_sum(None, 4, _lib.F64, comm, None) != 0
    assert l.aggf_comm_init(uid, 8, 0, 1, C.byref(comm)) != 0  # id too short
    _lib.check(l.aggf_comm_destroy(comm), "aggf_comm_destroy")


_RANK_SCRIPT = r"""
import ctypes as C, os, sys, time
sys.path.insert(0, sys.argv[1])
rank, world, idfile = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
import numpy as np, torch
from aggforce_amd import _lib, _kernels as K
from aggforce_amd.distributed import frame_shard
l = _lib.lib()
torch.cuda.set_device(rank)
uid = (C.c_char * 128)()
if rank == 0:
    _lib.check(l.aggf_comm_unique_id(uid, 128))
    with open(idfile + ".tmp", "wb") as f:
        f.write(bytes(uid))
    os.replace(idfile + ".tmp", idfile)
else:
    for _ in range(600):
        if os.path.exists(idfile):
            break
        time.sleep(0.1)
    C.memmove(uid, open(idfile, "rb").read(), 128)
comm = C.c_void_p()
_lib.check(l.aggf_comm_init(uid, 128, rank, world, C.byref(comm)), "aggf_comm_init")
T, N = 2000, 200
b, e = frame_shard(T, rank, world)
f = K.synth_normal(e - b, N, torch.float64, 5, frame_offset=b, sigma=30.0)
G = K.gram(f, None, None, N, torch.float64)
_lib.check(l.aggf_allreduce_sum(G.data_ptr(), G.numel(), _lib.F64, comm, _lib.stream_ptr()), "aggf_allreduce_sum")
torch.cuda.synchronize()
np.save(sys.argv[5] + f"/G{rank}.npy", G.cpu().numpy())
_lib.check(l.aggf_comm_destroy(comm))
"""


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL refuses two ranks on one GPU; needs 2 GPUs")
def test_two_process_gram_allreduce_through_the_c_abi(tmp_path):
    """Path B of INTEGRATION.md: no torch.distributed; the Gram shards of two processes (one GPU each) summed by RCCL."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    idfile = str(tmp_path / "rccl_id.bin")
    procs = [subprocess.Popen([sys.executable, "-c", _RANK_SCRIPT, ROOT, str(r), "2", idfile, str(tmp_path)], env=env,
                              stderr=subprocess.PIPE) for r in range(2)]
    errs = [p.communicate(timeout=300)[1].decode()[-1500:] for p in procs]
    assert all(p.returncode == 0 for p in procs), errs
    from aggforce_amd import _kernels as K

    G0, G1 = np.load(tmp_path / "G0.npy"), np.load(tmp_path / "G1.npy")
    assert np.array_equal(G0, G1)
    full = K.gram(K.synth_normal(2000, 200, torch.float64, 5, sigma=30.0), None, None, 200, torch.float64).cpu().numpy()
    assert np.max(np.abs(G0 - full)) < 1e-10 * np.max(np.abs(full))
