"""GPU: bench.py itself as the driver runs it -- launcher -> torch.distributed.run -> ranks -> ONE line.

The pool gives one GPU per box and RCCL refuses two ranks on one device, so the N = 2 case runs as a REHEARSAL
(`--rehearse-on-one-gpu`: both ranks on cuda:0, gloo between them): everything of the multi-rank path except the
transport -- frame shards, the packed all-reduce of G, the replicated solve, max-over-ranks timing, the single
result line -- is exercised exactly as at N = 8."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, env=env, timeout=timeout, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_two_ranks_rehearsal_on_one_gpu():
    line = run_bench("--gpus", "2", "--rehearse-on-one-gpu", "--workload", "tiny", "--steps", "2", "--warmup", "1")
    cfg = line["config"]
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["ms_per_step"] > 0 and line["scaling"] == "strong"
    assert cfg["replicated_solve_max_abs_diff_across_ranks"] == 0.0
    assert cfg["world_size_seen"] == 2 and cfg["backend"] == "gloo"
    assert not any("rccl" in k.lower() for k in cfg)              # nothing is called RCCL under gloo
    assert "REHEARSAL" in cfg["collective"] and cfg["frames_per_gpu"] == 2048
    assert cfg["constraint_residual"] < 1e-8
    assert "cpu_baseline" not in line                               # rank 0 at N = 1 only
    # the strong-scaling terms are measured, not projected: a HIP-event "allreduce" stage (pack + collective + unpack)
    # and the replicated part (solve + host time between the stages)
    assert cfg["stage_ms_per_step"]["allreduce"] > 0 and cfg["allreduce_ms_per_step"] == cfg["stage_ms_per_step"]["allreduce"]
    assert cfg["replicated_ms_per_step"] >= cfg["stage_ms_per_step"]["solve"] > 0
    assert cfg["replicated_ms_per_step"] < line["ms_per_step"]
    # the same workload in one process: same residual (the Gram matrix is summed over the shards)
    one = run_bench("--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert one["n_gpus"] == 1 and one["config"]["backend"] is None and one["config"]["allreduce_ms_per_step"] == 0.0
    assert "allreduce" not in one["config"]["stage_ms_per_step"]
    assert abs(one["config"]["residual"] / cfg["residual"] - 1.0) < 1e-9


@pytest.mark.parametrize("variant", ["pairs", "zeronet", "dense"])
def test_bench_variants_of_the_linear_workload(variant):
    """SURVEY 8(d)'s synthetic-input variants, small: the line is well formed, the map is feasible, and the
    CPU port beside it ran the same variant."""
    line = run_bench("--workload", "tiny", "--variant", variant, "--steps", "1", "--warmup", "1", "--cpu-frames", "512")
    cfg = line["config"]
    assert cfg["variant"] == variant and ("+" + variant) in cfg["workload"]
    assert cfg["constraint_residual"] < 1e-8 and line["value"] > 0
    assert cfg["n_red"] == (256 - 256 // 3 if variant == "pairs" else 256)
    assert line["roofline"]["frac"] > 0 and line["roofline"]["traffic"] is None
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    if variant == "pairs":
        # 171 reduced columns: the streaming kernel sums the groups on the way through LDS -- no pack pass, and the label
        # (read from the launch table) says so
        assert "n_red 171" in line["cpu_baseline"]["sample"]
        assert "gram_small_kernel<" in line["roofline"]["kernel"] and "pack_groups_kernel" not in line["roofline"]["kernel"]


def test_bench_under_torchrun_one_rank_uses_rccl():
    """The driver's N > 1 form -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- with the
    one rank this box can hold: the process group is RCCL (backend nccl), the Gram matrix goes through the (world of
    one) all-reduce path, and the line says so."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                           "--gpus", "1", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, proc.stdout.decode()[-2000:]
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["backend"] == "nccl (RCCL)" and cfg["world_size_seen"] == 1
    assert "RCCL all-reduce" in cfg["collective"] and cfg["replicated_solve_max_abs_diff_across_ranks"] == 0.0
    assert cfg["constraint_residual"] < 1e-8 and line["value"] > 0


def test_bench_kernel_label_is_the_string_rocprofv3_prints(tmp_path):
    """`roofline.kernel` is read from the library's own launch table (aggf_coverage_dump), not typed into bench.py: under
    rocprofv3's kernel trace of the same command the label must be the name of a kernel the profiler saw, template
    arguments included (round 4's hand-written label was one template argument short)."""
    import glob
    import shutil

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        pytest.skip("rocprofv3 not installed")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TMPDIR"] = str(tmp_path)
    for workload, family in (("tiny", "gram_small_kernel"), ("c2", "gram_tile_dma_kernel")):
        out_dir = tmp_path / workload
        proc = subprocess.run([rocprof, "--kernel-trace", "--stats", "--output-format", "csv", "-d", str(out_dir), "--", sys.executable,
                               os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "2", "--warmup", "1",
                               "--no-cpu-baseline"] + (["--frames", "20000"] if workload == "c2" else []),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600, cwd=str(tmp_path))
        assert proc.returncode == 0, proc.stderr.decode(errors="replace")[-3000:]
        lines = [ln for ln in proc.stdout.decode().splitlines() if ln.strip().startswith('{"metric"')]
        assert len(lines) == 1, proc.stdout.decode()[-2000:]
        label = json.loads(lines[0])["roofline"]["kernel"]
        name = label.split(" = ")[-1] if label.startswith("aggf_gram = ") else label
        name = name.split(" (")[0].strip()
        assert family in name, label
        stats = glob.glob(str(out_dir / "**" / "*kernel_stats.csv"), recursive=True)
        assert stats, "rocprofv3 wrote no kernel_stats.csv"
        seen = [row.split('","')[0].strip('"') for f in stats for row in open(f).read().splitlines()[1:]]
        assert any(s.startswith("void " + name + "(") for s in seen), (name, [s for s in seen if family in s])
