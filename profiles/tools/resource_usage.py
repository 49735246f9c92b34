#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of libvoxelhip's kernels, from hipcc -Rpass-analysis=kernel-resource-usage
(cross-compiles gfx950 without a GPU). Usage: python profiles/tools/resource_usage.py [substring filter]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
       "-fno-fast-math", f"-I{ROOT}/include", f"-I{ROOT}/voxel-rs_amd/csrc/hip", "-c", "-o", "/tmp/vx_resource_usage.o",
       str(ROOT / "voxel-rs_amd/csrc/hip/kernels_render.hip"), "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(.+?): (\S+) \[-Rpass", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r, n in zip(rows, names):
    n = re.sub(r"^void \(anonymous namespace\)::", "", n).split("(")[0]
    if flt in n:
        print(f"{n:70s} vgpr {r.get('VGPRs','?'):>4} sgpr {r.get('TotalSGPRs','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>5} "
              f"occ {r.get('Occupancy [waves/SIMD]','?'):>2} spill v {r.get('VGPRs Spill','?'):>3} s {r.get('SGPRs Spill','?'):>3}")
