"""VGPRs / spills / occupancy of the kernels of one source (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
    python tools/kernel_resources.py aggforce_amd/csrc/aggf_gram.hip [name filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                      "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[.*?\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for mangled, pretty in zip(rows, names):
    if flt and flt not in pretty:
        continue
    r = rows[mangled]
    print(f"{pretty.split('(')[0][:80]:80s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', 0):4d} spill {r.get('VGPRs Spill', 0):4d} "
          f"scratch {r.get('ScratchSize', 0):5d} occ {r.get('Occupancy', -1)} SGPR {r.get('SGPRs', -1)}")
