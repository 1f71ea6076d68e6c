#!/usr/bin/env python3
"""One short line per workload: ms per iteration and every kernel's average (a quick A/B aid).  usage: bench_brief.py workload [workload ...] [-- extra bench.py args]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
for wl in args or ["hex100"]:
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", wl, "--no-cpu-baseline", "--no-configs"] + extra, capture_output=True, text=True, env=dict(os.environ, SMOOTHMESH_BENCH_FULL="1"))
    try:
        d = json.loads(r.stdout.strip().split("\n")[-1])
    except Exception:
        print(wl, "FAILED", r.stderr[-400:]); continue
    ks = "  ".join("%s %.1f" % (k["name"], k["avg_us"]) for k in d.get("kernels", []))
    print("%-11s %.4f ms (cold %s)  parity %s | %s" % (wl, d["ms_per_step"], ("%.4f" % d["ms_per_step_cold"]) if d.get("ms_per_step_cold") else "-", d.get("parity_check", {}).get("ok", "-"), ks), flush=True)
