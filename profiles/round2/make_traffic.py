#!/usr/bin/env python3
"""profiles/round2/traffic.json from the outputs of profile_r2.sh (gpurun_out/prof_r2_{csvo,esvo}/pmc.json + kernel_stats.csv):
    python profiles/round2/make_traffic.py <commit>"""
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
out = {"_comment": "HBM-side traffic of the render kernel per launch (C3 workload, one frame at a time: bench.py --frames-in-flight 1), from separate "
                   "rocprofv3 --pmc passes (profiles/round2/profile_r2.sh; summaries next to this file). FETCH_SIZE/WRITE_SIZE are in KB (x 1024); no gfx950 "
                   "doubling applied: the reads are scattered 8-byte gathers (TCC_EA0_RDREQ x 64 B agrees with FETCH_SIZE within 3 %), an access width "
                   "MI355X_MICROARCH.md calls uncalibrated. WRITE_SIZE against 33.2 MB of RGBA32F pixels, image of an ESVO world: 48.6 MB from the pixel "
                   "stores (16-byte stores of lanes that finish at different times write partial lines) + 12 MB from the cost notes of 'expensive "
                   "sub-tiles first' (one atomic max per ray of 64 iterations and more: atomics are carried out at the memory side, 32 bytes each; "
                   "with the floor at 32 iterations they were 46 MB -- profiles/round2/run_p.sh: VX_HOT_FIRST=0 48.6 MB, =2 95.1 MB); the kernel for the image of a CSVO world (the build that lists its "
                   "inside-voxel rays) keeps a few registers in scratch in its service phases: the rest of its reads and writes.",
       "commit": sys.argv[1] if len(sys.argv) > 1 else "?"}
for fmt in ("csvo", "esvo"):
    d = ROOT / "gpurun_out" / f"prof_r2_{fmt}"
    pmc = json.loads((d / "pmc.json").read_text())
    k = [name for name in pmc if "render_persistent<3" in name or "render_persistent<4" in name][0]  # the image kernel (not the instrumented one)
    c = pmc[k]
    ns, calls = None, 0
    for r in csv.DictReader(open(d / "kernel_stats.csv")):
        if "render_persistent" in r["Name"] and int(r["Calls"]) > calls:
            ns, calls = float(r["AverageNs"]), int(r["Calls"])
    row = {"FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"], "TCC_EA0_RDREQ": c.get("TCC_EA0_RDREQ_sum"), "kernel_avg_ns_rocprof": ns,
           "bytes_per_launch": int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_INSTS_SALU": c["SQ_INSTS_SALU"],
           "SQ_THREAD_CYCLES_VALU": c["SQ_THREAD_CYCLES_VALU"], "valu_lane_utilisation": round(c["SQ_THREAD_CYCLES_VALU"] / 64.0 / c["SQ_INSTS_VALU"], 3),
           "TCC_HIT": c.get("TCC_HIT_sum"), "TCC_MISS": c.get("TCC_MISS_sum"), "kernel": k}
    out[fmt] = row
(ROOT / "profiles" / "round2" / "traffic.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
