#!/bin/bash
# round 5: rocprofv3 evidence for the deep configurations (C4: 4K on the depth-14 terrain; C5: 8K + resolve), one frame at a time
# (VX_FRAMES_IN_FLIGHT=1: a launch has the device to itself). Kernel trace and each PMC group are separate passes, each under a timeout: a pass whose
# counters the hardware cannot collect together aborts inside the profiler's handler and would otherwise sit there until gpurun's limit (it did: 50 minutes).
#   usage: profiles/round5/profile_configs.sh <csvo|esvo>
set -u
fmt=$1
out=gpurun_out/prof_r5_configs_$fmt
mkdir -p "$out"
export TMPDIR=/tmp VX_FRAMES_IN_FLIGHT=1
args="--format $fmt --configs C4 C5 --steps 8"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 profiles/configs_bench.py $args > "$out/trace.log" 2>&1
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc$i" -- python3 profiles/configs_bench.py $args > "$out/pmc$i.log" 2>&1
done
python3 - "$out" "$fmt" <<'PY' > "$out/summary.txt"
import csv, glob, sys, collections
out, fmt = sys.argv[1], sys.argv[2]
for f in glob.glob(out+'/trace/**/*kernel_stats.csv', recursive=True):
    print('== kernel stats (rocprofv3 --kernel-trace --stats; profiles/configs_bench.py --configs C4 C5, VX_FRAMES_IN_FLIGHT=1)')
    print(open(f).read())
print('== bench lines of the traced run')
print(''.join(l for l in open(out+'/trace.log') if l.startswith('{"config"'))[:3000])
# per launch, by launch size: C4's frames are 3840x2160, C5's 7680x4320 (the same kernel: told apart by their durations / wave counts)
for d in sorted(glob.glob(out+'/pmc*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'render_persistent' not in r['Kernel_Name']: continue
            acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            print('== pmc', k, '(the launches of the run in order: warm-up, counted and timed frames of C4, then of C5; means of the first and the second half)')
            for c,vals in v.items():
                h=len(vals)//2
                print('   %-28s n=%d first half mean=%.6g second half mean=%.6g' % (c, len(vals), sum(vals[:h])/max(h,1), sum(vals[h:])/max(len(vals)-h,1)))
PY
cat "$out/summary.txt"
