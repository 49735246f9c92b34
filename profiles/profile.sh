#!/bin/bash
# Collects the rocprofv3 evidence for one bench configuration (run on the GPU box via gpurun).
#   usage: scripts_profile.sh <tag> <bench args...>
# kernel-trace/stats and each PMC group run as separate passes (never combined), outputs under gpurun_out/prof_<tag>/.
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args="--no-cpu-baseline --steps 50 --warmup 10 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py $args > "$out/trace.log" 2>&1
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc$i" -- python3 bench.py $args > "$out/pmc$i.log" 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
for f in glob.glob(out+'/trace/**/*kernel_stats.csv', recursive=True):
    print('== kernel stats', f)
    print(open(f).read())
for d in sorted(glob.glob(out+'/pmc*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            if 'render_' not in k: continue
            print('== pmc', k)
            for c,vals in v.items():
                print('   %-32s n=%d mean=%.6g' % (c, len(vals), sum(vals)/len(vals)))
PY
