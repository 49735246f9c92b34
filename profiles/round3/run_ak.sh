#!/bin/bash
# round 3, pass AK: whole CUs reserved for the exchange and the assembly (VX_COMM_RESERVE_CUS: the render streams get a CU mask) instead of four wave
# slots on every CU: what the mask costs a plain render, what bench.py --force-sharded costs with it (one rank: all this pool can measure)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3ak; mkdir -p $O; rm -f $O/*
VX_COMM_RESERVE_CUS=8 timeout 600 python3 -m pytest tests -m gpu -x -q -k "kernel_versions or sharded or gather or full_size" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=" $O/pytest.log | cut -c1-200
for r in 0 4 8 16; do
  VX_COMM_RESERVE_CUS=$r timeout 300 python3 bench.py --no-cpu-baseline --no-sd500 --repeats 9 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('reserve $r plain  ', d['value'], d['ms_per_step'])"
  VX_COMM_RESERVE_CUS=$r timeout 300 python3 bench.py --no-cpu-baseline --no-sd500 --repeats 9 --force-sharded 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('reserve $r sharded', d['value'], d['ms_per_step'], d['config'].get('gather'), d.get('per_rank_ms'), d.get('exchange_ms'))"
done | tee $O/reserve.txt
VX_WAVES_PER_CU=12 timeout 300 python3 bench.py --no-cpu-baseline --no-sd500 --repeats 9 --force-sharded 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('12 waves per CU sharded', d['value'], d['ms_per_step'])" | tee -a $O/reserve.txt
