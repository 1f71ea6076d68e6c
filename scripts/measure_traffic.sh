#!/bin/bash
# usage (on the GPU box): scripts/measure_traffic.sh <workload> [steps]
# HBM traffic per kernel launch from rocprofv3 PMC counters, as MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (3 + 2 TCC slots), kernel dispatch tracing only.
# Writes profiles/r1/traffic_<workload>.json (bytes per launch; corrections applied by bench.py's reader).
wl=${1:-hex100}; steps=${2:-10}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/traffic_$wl
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export SMGPU_SIDE_STREAM=0   # counter collection serialises kernels: no cross-stream waits (see scripts/pmc_kernels.sh)
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py --no-cpu-baseline --no-configs --workload $wl --steps $steps --warmup 1 > /dev/null 2>$out/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py --no-cpu-baseline --no-configs --workload $wl --steps $steps --warmup 1 > /dev/null 2>$out/write.err
cd $root
python3 - "$out" "$wl" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out, wl = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("smgpu::", "").replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in acc.items():
    if k.startswith("__amd"):
        continue
    f = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1)
    w = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1)
    res[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches_sampled": len(c.get("FETCH_SIZE", []))}
doc = {"workload": wl, "units": "FETCH_SIZE/WRITE_SIZE as reported by rocprofv3 (KB per launch, mean over launches)",
       "correction": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B -> read bytes = 2 * FETCH_SIZE * 1024 for coalesced streams "
                     "(calibrated here on k_apply: known 50 B/point read, 24 B/point written; see profiles/r1/hex215c_pmc_*): "
                     "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024; an upper bound for kernels whose misses are 64-B requests",
       "kernels": res}
import os

json.dump(doc, open(f"gpurun_out/traffic_{wl}.json", "w"), indent=1)
print(json.dumps(doc)[:600])
PY
