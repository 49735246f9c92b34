#!/bin/bash
# round 2, GPU pass J: experiment X1 (top two levels of the image in LDS) -- parity, then A/B
set -u
mkdir -p gpurun_out/r2j; rm -f gpurun_out/r2j/*
timeout 600 python -m pytest tests -m gpu -x -q -k "kernel_versions" > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2j/pytest.log
timeout 600 python profiles/sweep.py --format esvo --depth 12 --configs "X=0,f=2" "X=1,f=2" "X=0,f=1,h=0" "X=1,f=1,h=0" --rounds 5 --steps 20 > gpurun_out/r2j/sweep_x1_esvo.txt 2>&1
export TMPDIR=/tmp
for x in 0 1; do
VX_HOT_LEVELS=$x rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/r2j/pmc_x$x -- python3 bench.py --format esvo --no-cpu-baseline --frames-in-flight 1 --steps 30 --warmup 5 --repeats 2 > gpurun_out/r2j/pmc_x$x.log 2>&1
python3 - gpurun_out/r2j/pmc_x$x $x <<'PY' >> gpurun_out/r2j/x1_pmc.txt
import csv, glob, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'render_persistent<3' in k:
        print('VX_HOT_LEVELS='+sys.argv[2], k)
        for c,vals in sorted(v.items()): print('   %-24s n=%d mean=%.6g' % (c, len(vals), sum(vals)/len(vals)))
PY
done
tail -n 3 gpurun_out/r2j/pytest.log; tail -n 4 gpurun_out/r2j/sweep_x1_esvo.txt; cat gpurun_out/r2j/x1_pmc.txt
