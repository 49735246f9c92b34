import subprocess, sys, json, re, os
def stat():
    d = {}
    for line in open("/sys/fs/cgroup/cpu.stat"):
        k, v = line.split()
        d[k] = int(v)
    return d
for i in range(4):
    for fmt in ("csvo", "esvo"):
        for workers in ("16", "6"):
            a = stat()
            env = dict(os.environ, VXH_EXP_STREAM_WORKERS=workers)
            out = subprocess.run([sys.executable, "profiles/stream_bench.py", "--format", fmt, "--scene-depth", "14", "--radius", "40", "--width", "3840", "--height", "2160", "--frames", "2"],
                                 capture_output=True, text=True, env=env).stdout
            b = stat()
            m = re.search(r'"initial_fill": (\{[^}]*\})', out)
            fill = json.loads(m.group(1)) if m else None
            print(fmt, "workers", workers, "fill", fill and fill["seconds"], "commit_s", fill and fill["commit_s"], "| whole process: usage_s", round((b["usage_usec"] - a["usage_usec"]) / 1e6, 2),
                  "nr_throttled", b.get("nr_throttled", 0) - a.get("nr_throttled", 0), "throttled_s", round((b.get("throttled_usec", 0) - a.get("throttled_usec", 0)) / 1e6, 3), flush=True)
