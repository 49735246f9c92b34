#!/bin/bash
# round 2, GPU pass O: instruction counters of the batched-service build against the lane-service build (ESVO, C3, one frame at a time)
set -u
O=gpurun_out/r2o; mkdir -p $O; rm -rf $O/*
export TMPDIR=/tmp
for b in 0 1; do
VX_BATCH=$b rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_b$b -- python3 bench.py --format esvo --no-cpu-baseline --frames-in-flight 1 --steps 30 --warmup 5 --repeats 2 > $O/pmc_b$b.log 2>&1
VX_BATCH=$b rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc2_b$b -- python3 bench.py --format esvo --no-cpu-baseline --frames-in-flight 1 --steps 30 --warmup 5 --repeats 2 > $O/pmc2_b$b.log 2>&1
python3 - $O/pmc_b$b $O/pmc2_b$b $b <<'PY' >> $O/batch_pmc.txt
import csv, glob, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:3]:
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:100]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'render_persistent<3' in k:
        print('VX_BATCH='+sys.argv[3], k)
        for c,vals in sorted(v.items()): print('   %-24s n=%d mean=%.6g' % (c, len(vals), sum(vals)/len(vals)))
PY
done
cat $O/batch_pmc.txt
