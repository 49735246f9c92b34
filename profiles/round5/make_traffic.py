#!/usr/bin/env python3
"""profiles/round5/traffic.json from the outputs of profile_r5.sh (gpurun_out/prof_r5_{csvo,esvo}/pmc.json + kernel_stats.csv):
    python profiles/round5/make_traffic.py <commit>"""
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from _pkg import csrc_hash  # noqa: E402
out = {"_comment": "HBM-side traffic and instruction counts of the render kernel per launch (C3 workload, one frame at a time: bench.py --frames-in-flight 1), "
                   "from separate rocprofv3 --pmc passes (profiles/round5/profile_r5.sh; summaries next to this file). FETCH_SIZE/WRITE_SIZE are in KB (x 1024); no gfx950 "
                   "doubling applied: the reads are scattered 8-byte gathers (TCC_EA0_RDREQ x 64 B agrees with FETCH_SIZE within a few %), an access width "
                   "MI355X_MICROARCH.md calls uncalibrated. WRITE_SIZE is to be read against 33.2 MB of RGBA32F pixels.",
       "commit": sys.argv[1] if len(sys.argv) > 1 else "?",
       # what the counters were measured on: bench.py quotes them only while the library's sources are these (_pkg.csrc_hash)
       "csrc_sha16": csrc_hash()}
for fmt in ("csvo", "esvo"):
    d = ROOT / "gpurun_out" / f"prof_r5_{fmt}"
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
           "TCC_HIT": c.get("TCC_HIT_sum"), "TCC_MISS": c.get("TCC_MISS_sum"), "kernel": k,
           "SQ_WAVE_CYCLES": c.get("SQ_WAVE_CYCLES"), "SQ_WAIT_ANY": c.get("SQ_WAIT_ANY"), "SQ_WAIT_INST_ANY": c.get("SQ_WAIT_INST_ANY"), "SQ_ACTIVE_INST_ANY": c.get("SQ_ACTIVE_INST_ANY"),
           "SQ_INSTS_VMEM_RD": c.get("SQ_INSTS_VMEM_RD"), "SQ_INSTS_LDS": c.get("SQ_INSTS_LDS"), "SQ_INSTS_SMEM": c.get("SQ_INSTS_SMEM"), "GRBM_GUI_ACTIVE": c.get("GRBM_GUI_ACTIVE")}
    out[fmt] = row
(ROOT / "profiles" / "round5" / "traffic.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
