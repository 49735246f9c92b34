#!/bin/bash
# round 3, the final pass at HEAD: whole GPU suite, smoke, the rocprofv3 / PMC evidence (run_j.sh -> traffic.json, which the bench line's issue model
# reads), then the bench lines (run_z.sh without its own run_j), the configurations (run_f.sh)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | cut -c1-200 | tee gpurun_out/final/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke | tee gpurun_out/final/smoke.txt
bash profiles/round3/run_j.sh > gpurun_out/final/run_j.log 2>&1; tail -n 4 gpurun_out/final/run_j.log | cut -c1-300
SKIP_RUN_J=1 bash profiles/round3/run_z.sh > gpurun_out/final/run_z.log 2>&1; tail -n 3 gpurun_out/final/run_z.log | cut -c1-300
bash profiles/round3/run_f.sh > gpurun_out/final/run_f.log 2>&1; tail -n 12 gpurun_out/final/run_f.log | cut -c1-200
