#!/bin/bash
# One-GPU size sweep of the hot path: the same workload from Infinity-Cache resident to tens of GiB on the device.
# usage: scripts/size_sweep.sh [steps] workload...     (GPU legs only: no oracle, no configs[])
export SMOOTHMESH_BENCH_FULL=1   # bench.py prints its full document (not the compact driver line) on stdout
steps=${1:-50}; shift
for w in "$@"; do
    echo "== $w"
    timeout 1500 python bench.py --workload "$w" --steps "$steps" --warmup 5 \
        --no-cpu-baseline --no-configs 2>&1 | python scripts/bench_summary.py
done
