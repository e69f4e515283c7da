"""Copy the judged parts of a tools/profile_gpu.sh run from gpurun_out/ (scratch) into profiles/ (tracked):
    python tools/commit_profile.py <tag> "<bench args used>"
writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), profiles/<tag>_rocprof_summary.json
(top kernels + per-dispatch PMC counters) and profiles/<tag>_mfma_counters.json (MFMA utilisation per kernel)."""
import glob
import json
import os
import shutil
import sys

tag, bench_args = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
# (gpurun_out/ keeps the files of earlier runs of the same tag: the newest one is this run's)
stats = sorted(glob.glob(os.path.join(src, f"{tag}_trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(dst, f"{tag}_kernel_stats.csv"))
summary = json.load(open(os.path.join(src, f"{tag}_summary.json")))
summary["commands"] = {
    "trace": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline {bench_args}",
    "pmc": f"rocprofv3 --pmc <counters> -- python3 bench.py --no-cpu-baseline {bench_args} --steps 1 --warmup 1  (one run per "
           "counter group: FETCH_SIZE | WRITE_SIZE | TCC_* | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...)",
    "units": "FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 a 128-byte request is counted as 64 bytes: HBM-side read bytes = "
             "2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md); per_dispatch = mean over the dispatches of the run",
}
try:  # which tree the profiled command RAN FROM: the commit whose sources hash like the copy on the GPU box did
    import hashlib
    import subprocess

    def source_sha1():
        h = hashlib.sha1()
        files = []
        for pat in ("aggforce_amd/csrc/*.hip", "aggforce_amd/csrc/*.h", "include/*.h", "aggforce_amd/*.py", "aggforce_amd/*/*.py", "bench.py"):
            files += glob.glob(os.path.join(root, pat))
        for f in sorted(files):
            h.update(os.path.relpath(f, root).encode())
            h.update(open(f, "rb").read())
        return h.hexdigest()

    head = subprocess.run(["git", "-C", root, "log", "-1", "--format=%h %cs"], stdout=subprocess.PIPE).stdout.decode().split()
    dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "aggforce_amd", "include", "bench.py"],
                           stdout=subprocess.PIPE).stdout.decode().strip()
    same = summary.get("source_sha1") == source_sha1()
    summary["profiled_from"] = {
        "tree": head[0] if same and not dirty else None, "date": head[1],
        "note": ("the commit the profiled commands ran from: the sources on the GPU box hash (source_sha1) like this commit's"
                 if same and not dirty else
                 "the working tree differed from HEAD %s when the profile was committed; source_sha1 identifies the sources" % head[0])}
except Exception:
    pass
trace_log = os.path.join(src, f"{tag}_trace.log")
if os.path.exists(trace_log):
    lines = [l for l in open(trace_log).read().splitlines() if l.startswith("{")]
    if lines:
        summary["bench_line_of_the_traced_run"] = json.loads(lines[-1])
json.dump(summary, open(os.path.join(dst, f"{tag}_rocprof_summary.json"), "w"), indent=1)
mf = {"command": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES -- python3 "
                 f"bench.py --no-cpu-baseline {bench_args} --steps 1 --warmup 1",
      "note": "per-dispatch averages; GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA utilisation = "
              "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
      "kernels": {}}
for e in summary.get("pmc_mfma", []):
    busy, act = e.get("SQ_VALU_MFMA_BUSY_CYCLES"), e.get("GRBM_GUI_ACTIVE")
    if busy and act and busy["per_dispatch"] > 0:
        vals = {k: v["per_dispatch"] for k, v in e.items() if isinstance(v, dict)}
        vals["dispatches"] = busy["dispatches"]
        vals["mfma_utilisation"] = busy["per_dispatch"] / (1024.0 * act["per_dispatch"] / 8.0)
        mf["kernels"][e["kernel"]] = vals
if mf["kernels"]:
    json.dump(mf, open(os.path.join(dst, f"{tag}_mfma_counters.json"), "w"), indent=1)
print("profiles/%s_*: " % tag, [os.path.basename(p) for p in glob.glob(os.path.join(dst, f"{tag}_*"))])
