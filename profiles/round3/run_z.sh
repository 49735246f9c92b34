#!/bin/bash
# round 3, final pass: the bench lines at HEAD (both formats, the default command), forced-sharded lines, the rocprofv3 evidence, the configurations
set -u
export TMPDIR=/tmp
O=gpurun_out/r3z; mkdir -p $O; rm -f $O/*
for f in csvo esvo; do timeout 900 python3 bench.py --format $f > $O/final_bench_$f.json 2> $O/final_bench_$f.err; tail -c 600 $O/final_bench_$f.json; echo; done
timeout 600 python3 bench.py --no-cpu-baseline --force-sharded > $O/final_bench_csvo_forced_sharded.json 2> $O/sharded.err; tail -c 300 $O/final_bench_csvo_forced_sharded.json; echo
timeout 600 python3 bench.py --no-cpu-baseline --force-sharded --gather-format rgba32f > $O/final_bench_csvo_forced_sharded_rgba32f.json 2>> $O/sharded.err
[ -n "${SKIP_RUN_J:-}" ] || { bash profiles/round3/run_j.sh > $O/run_j.log 2>&1; tail -n 4 $O/run_j.log | cut -c1-300; }
