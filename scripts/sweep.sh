#!/bin/bash
# usage: scripts/sweep.sh "ENV1=.. ENV2=.." "ENV..." ...   (each arg = one bench configuration)
export SMOOTHMESH_BENCH_FULL=1   # bench.py prints its full document (not the compact driver line) on stdout
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg SMGPU_VERBOSE=1 python bench.py --no-cpu-baseline $BENCH_ARGS 2>&1 | grep -E "^\{|\[smgpu\]" | python scripts/bench_summary.py
done
