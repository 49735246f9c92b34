#!/usr/bin/env python3
"""Dumps the traversal loop (the innermost loop around the image's 8-byte entry load) of one render_persistent variant from the
compiler's assembly, with source lines, and counts its instructions. (Counted: the instructions between the loop's head and its back
edge. A path of the loop that the compiler lays out behind it -- since the load-ahead change the PUSH path, 16 VALU -- is not.) Usage: python profiles/tools/hot_loop.py [mangled-substring] [--asm]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
want = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "render_persistentILi3ELb0ELb0ELi0ELi13ELb0E"  # <IMAGE, image-only, no counters, no walks, 13 stack levels>: the ESVO image kernel
show = "--asm" in sys.argv
out = "/tmp/vx_hot_loop.s"
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
       "-fno-fast-math", "-gline-tables-only", f"-I{ROOT}/include", f"-I{ROOT}/voxel-rs_amd/csrc/hip", "-S", "--cuda-device-only", "-o", out,
       str(ROOT / "voxel-rs_amd/csrc/hip/kernels_render.hip")]
subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
s = open(out).read().split("\n")
files = {}
for l in s:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = [i for i, l in enumerate(s) if want in l and l.startswith("_Z")][0]
end = [i for i in range(start, len(s)) if s[i].strip().startswith(".Lfunc_end")][0]
k = s[start:end + 1]
loads = [i for i, l in enumerate(k) if "buffer_load_dwordx2" in l or "global_load_dwordx2" in l]
i0 = loads[0]
labels = {l.split(":")[0]: i for i, l in enumerate(k) if re.match(r"^\.LBB\d+_\d+:", l)}
back = []
for i, l in enumerate(k):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.search(r"s_branch (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        back.append((labels[m.group(1)], i))
cands = sorted([(a, b) for a, b in back if a <= i0 <= b], key=lambda x: x[1] - x[0])
a, b = cands[0]
body = k[a:b + 1]
cur = None
per_line = {}
n = {"valu": 0, "salu": 0, "ds": 0, "vmem": 0, "mov": 0}
for l in body:
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    t = l.strip()
    if re.match(r"v_", t):
        n["valu"] += 1
        per_line[cur] = per_line.get(cur, 0) + 1
        if t.startswith("v_mov_b"):
            n["mov"] += 1
    elif re.match(r"s_", t):
        n["salu"] += 1
    elif t.startswith("ds_"):
        n["ds"] += 1
    elif "buffer_" in t or "global_" in t or "scratch_" in t:
        n["vmem"] += 1
    if show and t and not t.startswith(";") and not t.startswith("."):
        print(f"{cur[0] if cur else '?':16s}:{cur[1] if cur else 0:5d}  {t}")
print(want, n)
for key, v in sorted(per_line.items(), key=lambda x: (x[0][0], x[0][1])):
    print(f"  {key[0]}:{key[1]}  {v}")
