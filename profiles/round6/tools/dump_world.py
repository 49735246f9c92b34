#!/usr/bin/env python3
"""The depth-14 terrain's frame, written to /tmp/world_{csvo,esvo}.bin ([u64 arena bytes in use][the frame]) for tools/imgbench.cpp and tools/walkbench.cpp.
python profiles/round6/tools/dump_world.py [csvo] [esvo]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from _pkg import load_package
vra = load_package()
want = sys.argv[1:] or ["csvo", "esvo"]
for name, fmt in (("csvo", vra.SVO_CSVO), ("esvo", vra.SVO_ESVO)):
    if name not in want:
        continue
    t0 = time.time()
    w = vra.World(fmt)
    st = w.build_heightfield(14)
    f = w.frame()
    print(name, st, w.size_in_bytes, f.size * 4, round(time.time() - t0, 1), flush=True)
    with open(f"/tmp/world_{name}.bin", "wb") as fh:
        fh.write(int(w.size_in_bytes).to_bytes(8, "little"))
        f.tofile(fh)
    del w, f
