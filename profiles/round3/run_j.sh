#!/bin/bash
# round 3, pass J: rocprofv3 evidence at HEAD: one frame at a time (kernel trace + PMC groups, both formats) and the timed mode's trace
set -u
export TMPDIR=/tmp
for f in csvo esvo; do bash profiles/round3/profile_r3.sh $f > gpurun_out/prof_r3_$f.log 2>&1; tail -n 3 gpurun_out/prof_r3_$f.log; done
python3 profiles/round3/make_traffic.py "$(cat gpurun_out/.head 2>/dev/null || echo HEAD)" > /dev/null; cp profiles/round3/traffic.json gpurun_out/traffic_r3.json
for f in csvo esvo; do bash profiles/round3/profile_fif2.sh $f > gpurun_out/prof_r3_${f}_fif2.log 2>&1; tail -n 2 gpurun_out/prof_r3_${f}_fif2.log; done
python3 -c "
import json; t=json.load(open('gpurun_out/traffic_r3.json'))
for f in ('csvo','esvo'):
    r=t[f]; print(f, 'fetch MB', r['FETCH_SIZE_KB']/1024, 'write MB', r['WRITE_SIZE_KB']/1024, 'valu', r['SQ_INSTS_VALU'], 'salu', r['SQ_INSTS_SALU'], 'lanes', r['valu_lane_utilisation'], 'ns', r['kernel_avg_ns_rocprof'], 'wave_cycles', r['SQ_WAVE_CYCLES'], 'wait', r['SQ_WAIT_ANY'], 'vmem', r['SQ_INSTS_VMEM_RD'], 'lds', r['SQ_INSTS_LDS'])
"
