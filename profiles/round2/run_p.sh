# round 2, GPU pass P: HBM writes of the render kernel with and without the cost notes of "expensive sub-tiles first" (VX_HOT_FIRST bits)
set -u
O=gpurun_out/r2p; mkdir -p $O; rm -rf $O/*
export TMPDIR=/tmp
for h in 0 2 7; do
VX_HOT_FIRST=$h timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_h$h -- python3 bench.py --format esvo --no-cpu-baseline --frames-in-flight 1 --steps 30 --warmup 5 --repeats 2 > $O/pmc_h$h.log 2>&1
python3 - $O/pmc_h$h $h <<'PY' >> $O/wr.txt
import csv, glob, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'render_persistent<3' in k:
        print('VX_HOT_FIRST='+sys.argv[2], {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
cat $O/wr.txt
