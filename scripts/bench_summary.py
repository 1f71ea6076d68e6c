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
    if d["config"].get("device_bytes_per_gpu"):
        c = d["config"]
        print(f"   {c['cells_per_gpu']} cells, {c['points_per_gpu']} points, {c['device_bytes_per_gpu'] / 2**30:.2f} GiB on the device "
              f"({c['device_bytes_per_gpu'] / c['cells_per_gpu']:.0f} B per cell); host max RSS {d.get('host_max_rss_gib', 0):.1f} GiB; phases {d.get('phases')}")
    for k in d["kernels"]:
        gb = f"{k['algo_GBps']:8.1f} GB/s algorithmic" if k.get("algo_GBps") else (f"{k['algo_f64_Tops']:8.2f} T FP64 instr/s algorithmic" if k.get("algo_f64_Tops") else "")
        print(f"   {k['name'][:40]:40s} {k['avg_us']:9.1f} us  {gb}")
        for kn, kv in (k.get("kernels") or {}).items():
            print(f"        {kn:35s} {kv:9.1f} us")
    ch = d.get("chains")
    if ch:
        print(f"   chains (us, kernels timed alone): serial {ch['serial_part_us']:.0f}  main {ch['main_stream_chain_us']:.0f}  side {ch['side_stream_chain_us']:.0f}  "
              f"critical path {ch['critical_path_us']:.0f}  sum {ch['sum_all_alone_us']:.0f}  measured step {ch['ms_per_step_measured'] * 1e3:.0f}")
    g = d.get("roofline_centroid_gather") or {}
    for name, a in (g.get("accountings") or {}).items():
        print(f"   centroid gather, {name}: {a['achieved_GBps']:.0f} GB/s = {a['frac']:.3f} of the HBM peak")
