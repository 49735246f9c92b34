#!/bin/bash
# Where the streamer's initial fill spends its commits' time: VX_COMMIT_TIMING lines summed per run (image update by part, allocation + upload).
for i in 1 2 3 4; do
  for f in csvo esvo; do
    VX_COMMIT_TIMING=1 python3 profiles/stream_bench.py --format $f --scene-depth 14 --radius 40 --width 3840 --height 2160 --frames 2 2>/tmp/fill_err.txt | FMT=$f python3 -c "
import json,sys,os,re
fill=None
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        fill=json.loads(line)['initial_fill']
tot=[0.0]*6; n=0; worst=[]
for line in open('/tmp/fill_err.txt'):
    m=re.search(r'root walk ([\d.]+) s, chunk walk ([\d.]+), place ([\d.]+), encode ([\d.]+), root \+ header ([\d.]+); allocation \+ upload ([\d.]+)', line)
    if m:
        v=[float(x) for x in m.groups()]; n+=1
        for k in range(6): tot[k]+=v[k]
        worst.append(v[5])
worst.sort()
print(os.environ['FMT'], fill['seconds'], 'commit_s', fill['commit_s'], 'lines', n, 'image parts', [round(x,3) for x in tot[:5]], 'upload', round(tot[5],3), 'upload median/p99/max us', round(worst[len(worst)//2]*1e6), round(worst[int(len(worst)*0.99)]*1e6), round(worst[-1]*1e6))
"
  done
done
