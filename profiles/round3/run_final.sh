#!/bin/bash
# round 3, the final pass at HEAD: whole GPU suite, smoke, the bench lines + rocprofv3 evidence (run_z.sh), the configurations (run_f.sh)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -n 3 | cut -c1-200 | tee gpurun_out/final/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2 | tee gpurun_out/final/smoke.txt
bash profiles/round3/run_z.sh > gpurun_out/final/run_z.log 2>&1; tail -n 3 gpurun_out/final/run_z.log | cut -c1-300
bash profiles/round3/run_f.sh > gpurun_out/final/run_f.log 2>&1; tail -n 12 gpurun_out/final/run_f.log | cut -c1-200
