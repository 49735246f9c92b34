#!/bin/bash
# round 6, step 1: where the depth-14 frames wait. C4 (3840x2160, primary + shadow, static depth-14 terrain), one frame at a time, and C3's frame
# (1920x1080, depth 12) in the same passes for comparison. (a) VX_WAVES_PER_CU x frames-in-flight sweep in the measurement build; (b) rocprofv3
# kernel trace + separate --pmc passes: SQ (issue, VMEM latency), TA / TCP (address processing, L1), UTCL1 (translation), TCC (L2) and its
# memory side. Every pass under its own timeout (a group the hardware cannot collect together aborts inside the profiler).
#   usage: profiles/round6/c4_counters.sh <csvo|esvo> [sweep|pmc|all]
set -u
fmt=$1
what=${2:-all}
out=gpurun_out/r6_c4_$fmt
mkdir -p "$out"
export TMPDIR=/tmp
if [ "$what" = sweep ] || [ "$what" = all ]; then
  VX_LIB_DIR=voxel-rs_amd/lib/lib_tl timeout -k 10 900 python3 profiles/round6/deep_frames.py --format $fmt --frames 12 \
     --sweep "16:1 16:2 16:3 16:4 12:1 12:2 12:3 12:4 8:1 8:2 8:4" > "$out/sweep.txt" 2>&1
  cat "$out/sweep.txt" | grep '^{' | cut -c1-220
fi
if [ "$what" = pmc ] || [ "$what" = all ]; then
  export VX_FRAMES_IN_FLIGHT=1
  run="python3 profiles/round6/deep_frames.py --format $fmt --frames 6 --warmup 2"
  run3="python3 profiles/round6/deep_frames.py --format $fmt --depth 12 --size 1920x1080 --frames 6 --warmup 2"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $run > "$out/trace.log" 2>&1
  i=0
  for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
             "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
             "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum" \
             "TCC_TAG_STALL_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" \
             "GRBM_UTCL2_BUSY TCP_GATE_EN1_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_LFIFO_FULL_sum TD_TC_STALL TD_TD_BUSY" \
             "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 420 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc$i" -- $run > "$out/pmc$i.log" 2>&1
    [ -n "${SKIP_C3:-}" ] || timeout -k 10 240 rocprofv3 --pmc $pmc --output-format csv -d "$out/c3_pmc$i" -- $run3 > "$out/c3_pmc$i.log" 2>&1
  done
  python3 profiles/round6/pmc_summary.py "$out" > "$out/summary.txt"
  cat "$out/summary.txt"
fi
