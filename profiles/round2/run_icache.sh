#!/bin/bash
# round 2: instruction-cache counters of the render kernel (C3 ESVO, one frame at a time)
set -u
O=gpurun_out/icache; mkdir -p $O; rm -rf $O/*
export TMPDIR=/tmp
i=0
for pmc in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc$i -- python3 bench.py --format esvo --no-cpu-baseline --frames-in-flight 1 --steps 30 --warmup 5 --repeats 2 > $O/pmc$i.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'render_persistent<3' in k:
        print(k)
        for c,vals in sorted(v.items()): print('   %-28s n=%d mean=%.6g' % (c, len(vals), sum(vals)/len(vals)))
PY
