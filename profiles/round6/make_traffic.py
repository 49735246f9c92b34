#!/usr/bin/env python3
"""profiles/round6/traffic.json from the outputs of profile_r6.sh (gpurun_out/prof_r6_{csvo,esvo}/pmc.json + kernel_stats.csv: C3, one frame at a time) and of
c4_counters.sh (gpurun_out/r6_c4_{csvo,esvo}/pmc.json + the trace's kernel_stats.csv: C4, 3840x2160 on the depth-14 terrain, one frame at a time):
    python profiles/round6/make_traffic.py <commit>"""
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from _pkg import csrc_hash  # noqa: E402

out = {"_comment": "HBM-side traffic and instruction counts of the render kernel per launch, from separate rocprofv3 --pmc passes (profiles/round6/profile_r6.sh: "
                   "C3, bench.py --frames-in-flight 1; profiles/round6/c4_counters.sh: C4, deep_frames.py, VX_FRAMES_IN_FLIGHT=1; summaries next to this file). "
                   "read_bytes = 128 x TCC_EA0_RDREQ_128B + 64 x TCC_EA0_RDREQ_64B + 32 x TCC_EA0_RDREQ_32B: the memory side's read requests by their size -- nearly "
                   "all of this kernel's are 128-byte requests, which FETCH_SIZE (KB, x 1024) tallies at 64 bytes each: MI355X_MICROARCH.md's gfx950 correction "
                   "(double it) applies to these 8-byte gathers too (round 5 argued it did not; the request-size counters settle it). WRITE_SIZE (KB) is exact "
                   "and is to be read against 33.2 MB (C3) / 132.7 MB (C4) of RGBA32F pixels. bytes_per_launch = read_bytes + WRITE_SIZE x 1024.",
       "commit": sys.argv[1] if len(sys.argv) > 1 else "?",
       # what the counters were measured on: bench.py quotes them only while the library's sources are these (_pkg.csrc_hash)
       "csrc_sha16": csrc_hash()}


def row_of(pmc, kernel_ns):
    k = [name for name in pmc if ("render_persistent<3" in name or "render_persistent<4" in name) and "SQ_INSTS_VALU" in pmc[name]][0]  # the image kernel
    c = pmc[k]
    g = lambda n: c.get(n) if c.get(n) is not None else c.get(n + "_sum")
    read_bytes = 128 * (g("TCC_EA0_RDREQ_128B") or 0) + 64 * (g("TCC_EA0_RDREQ_64B") or 0) + 32 * (g("TCC_EA0_RDREQ_32B") or 0)
    return {"kernel": k, "kernel_avg_ns_rocprof": kernel_ns, "FETCH_SIZE_KB": c.get("FETCH_SIZE"), "WRITE_SIZE_KB": c.get("WRITE_SIZE"),
            "TCC_EA0_RDREQ": g("TCC_EA0_RDREQ"), "TCC_EA0_RDREQ_128B": g("TCC_EA0_RDREQ_128B"), "TCC_EA0_RDREQ_64B": g("TCC_EA0_RDREQ_64B"),
            "TCC_EA0_RDREQ_32B": g("TCC_EA0_RDREQ_32B"), "read_bytes": int(read_bytes),
            "bytes_per_launch": int(read_bytes + (c.get("WRITE_SIZE") or 0) * 1024),
            "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_INSTS_SALU": c["SQ_INSTS_SALU"], "SQ_THREAD_CYCLES_VALU": c.get("SQ_THREAD_CYCLES_VALU"),
            "valu_lane_utilisation": round(c["SQ_THREAD_CYCLES_VALU"] / 64.0 / c["SQ_INSTS_VALU"], 3) if c.get("SQ_THREAD_CYCLES_VALU") else None,
            "TCC_HIT": g("TCC_HIT"), "TCC_MISS": g("TCC_MISS"), "TCC_REQ": g("TCC_REQ"), "TCP_TOTAL_CACHE_ACCESSES": g("TCP_TOTAL_CACHE_ACCESSES"),
            "TCP_TCC_READ_REQ": g("TCP_TCC_READ_REQ"), "TCP_TCC_READ_REQ_LATENCY": g("TCP_TCC_READ_REQ_LATENCY"), "TCC_EA0_RDREQ_LEVEL": g("TCC_EA0_RDREQ_LEVEL"),
            "TCP_UTCL1_REQUEST": g("TCP_UTCL1_REQUEST"), "TCP_UTCL1_TRANSLATION_MISS": g("TCP_UTCL1_TRANSLATION_MISS"),
            "SQ_WAVE_CYCLES": c.get("SQ_WAVE_CYCLES"), "SQ_WAIT_ANY": c.get("SQ_WAIT_ANY"), "SQ_WAIT_INST_ANY": c.get("SQ_WAIT_INST_ANY"),
            "SQ_ACTIVE_INST_ANY": c.get("SQ_ACTIVE_INST_ANY"), "SQ_INSTS_VMEM_RD": c.get("SQ_INSTS_VMEM_RD"), "SQ_INSTS_LDS": c.get("SQ_INSTS_LDS"),
            "SQ_INSTS_SMEM": c.get("SQ_INSTS_SMEM"), "GRBM_GUI_ACTIVE": c.get("GRBM_GUI_ACTIVE")}


def kernel_ns(path):
    ns, calls = None, 0
    for r in csv.DictReader(open(path)):
        if "render_persistent" in r["Name"] and int(r["Calls"]) > calls:
            ns, calls = float(r["AverageNs"]), int(r["Calls"])
    return ns


for fmt in ("csvo", "esvo"):
    d = ROOT / "gpurun_out" / f"prof_r6_{fmt}"
    out[fmt] = row_of(json.loads((d / "pmc.json").read_text()), kernel_ns(d / "kernel_stats.csv"))
out["C4"] = {"workload": "3840x2160 primary + shadow, static depth-14 terrain, a still view, one frame at a time (profiles/round6/deep_frames.py)"}
for fmt in ("csvo", "esvo"):
    d = ROOT / "gpurun_out" / f"r6_c4_{fmt}"
    stats = glob.glob(str(d / "trace" / "**" / "*kernel_stats.csv"), recursive=True)
    out["C4"][fmt] = row_of(json.loads((d / "pmc.json").read_text()), kernel_ns(stats[0]) if stats else None)
(ROOT / "profiles" / "round6" / "traffic.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
