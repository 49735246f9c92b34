#!/usr/bin/env python3
"""Summary of c4_counters.sh's passes: per render kernel and counter, the mean over the pass's last launches (the timed frames), every
instance / dimension of a counter summed per dispatch first."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats; deep_frames.py, VX_FRAMES_IN_FLIGHT=1)")
    print(open(f).read())
for log in sorted(glob.glob(out + "/*.log")):
    for line in open(log):
        if line.startswith("{"):
            print("== %s: %s" % (log.split("/")[-1], line.strip()[:260]))
merged = collections.OrderedDict()  # kernel -> counter -> mean of the timed launches (every pass of the directory)
for d in sorted(glob.glob(out + "/*pmc*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.OrderedDict()  # (kernel, dispatch) -> counter -> sum
        for r in csv.DictReader(open(f)):
            if "render_persistent" not in r["Kernel_Name"]:
                continue
            key = (r["Kernel_Name"][:64], int(r["Dispatch_Id"]))
            per.setdefault(key, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
        kernels = collections.OrderedDict()
        for (k, _), v in per.items():
            kernels.setdefault(k, []).append(v)
        for k, launches in kernels.items():
            last = launches[-4:] if len(launches) >= 6 else launches
            print("== %s %s (%d launches; mean of the last %d)" % (d.rstrip("/").split("/")[-1], k, len(launches), len(last)))
            for c in last[0]:
                print("   %-44s %.6g" % (c, sum(l[c] for l in last) / len(last)))
                if not d.rstrip("/").split("/")[-1].startswith("c3_"):
                    merged.setdefault(k, collections.OrderedDict())[c] = sum(l[c] for l in last) / len(last)
import json

json.dump(merged, open(out + "/pmc.json", "w"), indent=1)
