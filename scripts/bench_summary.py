#!/usr/bin/env python3
"""Print a compact per-kernel table from bench.py JSON lines on stdin."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if line.startswith("[smgpu]"):
        print(line)
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print(f"{d['config']['workload'][:60]}...  value={d['value']/1e9:.3f} Gpts/s  ms/step={d['ms_per_step']:.4f}  (with events {d['ms_per_step_with_events']:.4f})")
    for k in d["kernels"]:
        print(f"   {k['name']:22s} {k['avg_us']:9.1f} us  {k['algo_GBps']:8.1f} GB/s algorithmic")
