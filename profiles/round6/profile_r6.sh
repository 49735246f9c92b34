#!/bin/bash
# round 6: rocprofv3 evidence for the C3 bench, ONE FRAME AT A TIME (--frames-in-flight 1: every launch of the render kernel has the device to itself, so the
# profiler's average duration is the kernel's own -- the figure bench.py reports as roofline.kernel_exclusive_ms). Kernel trace and each PMC group are
# separate passes (never combined). The memory side by request size (TCC_EA0_RDREQ_{32B,64B,128B}): what FETCH_SIZE, which tallies every request at 64 bytes,
# has to be corrected by on gfx950.
#   usage: profiles/round6/profile_r6.sh <csvo|esvo>
set -u
fmt=$1
out=gpurun_out/prof_r6_$fmt
mkdir -p "$out"
export TMPDIR=/tmp
args="--format $fmt --no-cpu-baseline --no-extras --frames-in-flight 1 --steps 50 --warmup 10 --repeats 3 --sustained-seconds 0"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py $args > "$out/trace.log" 2>&1
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc$i" -- python3 bench.py $args > "$out/pmc$i.log" 2>&1
done
python3 - "$out" "$fmt" <<'PY' > "$out/summary.txt"
import csv, glob, sys, collections, json
out, fmt = sys.argv[1], sys.argv[2]
for f in glob.glob(out+'/trace/**/*kernel_stats.csv', recursive=True):
    print('== kernel stats (rocprofv3 --kernel-trace --stats; bench.py --frames-in-flight 1)')
    print(open(f).read())
    import shutil; shutil.copy(f, out + '/kernel_stats.csv')
tot = collections.defaultdict(dict)
for d in sorted(glob.glob(out+'/pmc*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        per = collections.OrderedDict()  # (kernel, dispatch) -> counter -> sum over the counter's instances
        for r in csv.DictReader(open(f)):
            per.setdefault((r['Kernel_Name'][:70], r['Dispatch_Id']), collections.defaultdict(float))[r['Counter_Name']] += float(r['Counter_Value'])
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for (k, _), v in per.items():
            for c, x in v.items():
                acc[k][c].append(x)
        for k, v in acc.items():
            if 'render_' not in k: continue
            print('== pmc', k)
            for c, vals in v.items():
                print('   %-32s n=%d mean=%.6g' % (c, len(vals), sum(vals)/len(vals)))
                tot[k][c] = sum(vals)/len(vals)
json.dump(tot, open(out + '/pmc.json', 'w'), indent=1)
PY
tail -70 "$out/summary.txt"
grep -h '"metric"' "$out/trace.log" | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under the tracer:', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'])"
