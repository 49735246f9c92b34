#!/bin/bash
# round 4, pass S6: what the tile order changes in the counters -- tiles along the rows (VX_TILE_NUMBERING=0) against strips (default), C3 CSVO, two frames in
# flight (the timed mode) and one at a time; PMC passes only (never with a trace)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4s/pmc_order; mkdir -p $O
for num in 0 1; do for fif in 2 1; do
  i=0
  for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    VX_TILE_NUMBERING=$num rocprofv3 --pmc $pmc --output-format csv -d "$O/n${num}_f${fif}_$i" -- python3 bench.py --format csvo --no-cpu-baseline --no-extras --frames-in-flight $fif --steps 50 --warmup 10 --repeats 3 > "$O/n${num}_f${fif}_$i.log" 2>&1
  done
done; done
python3 - "$O" <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for num in (0, 1):
    for fif in (2, 1):
        tot = {}
        for d in sorted(glob.glob(f"{out}/n{num}_f{fif}_*/")):
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                acc = collections.defaultdict(list)
                for r in csv.DictReader(open(f)):
                    if "render_persistent<3, false, false, 3" in r["Kernel_Name"]:
                        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                for c, v in acc.items():
                    tot[c] = sum(v) / len(v)
        print(f"numbering {num} frames in flight {fif}:", {k: round(v) for k, v in sorted(tot.items())})
        if "SQ_WAVE_CYCLES" in tot:
            print("   wait share", round(tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 3), "issue-stall share", round(tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"], 3), "active share", round(tot["SQ_ACTIVE_INST_ANY"] / tot["SQ_WAVE_CYCLES"], 3),
                  "L2 miss rate", round(tot.get("TCC_MISS_sum", 0) / max(1, tot.get("TCC_MISS_sum", 0) + tot.get("TCC_HIT_sum", 0)), 3), "L1 miss rate", round(tot.get("TCP_TCC_READ_REQ_sum", 0) / max(1, tot.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1)), 3))
PY
