#!/bin/bash
# usage (on the GPU box): scripts/ab_bench.sh <workload> <steps> [variant.so ...]
# one bench.py line per library build (the in-tree libsmgpu.so first, then every variant given), printed as
# "<lib> ms_per_step [kernel avg_us ...]" -- A/B of build-time switches (variants are built into smoothmesh_amd/csrc/variants/).
export SMOOTHMESH_BENCH_FULL=1   # bench.py prints its full document (not the compact driver line) on stdout
wl=$1; steps=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/ab
for v in default "$@"; do
  name=$(basename "$v" .so)
  if [ "$v" = default ]; then unset SMOOTHMESH_SMGPU_LIB; else export SMOOTHMESH_SMGPU_LIB=$root/$v; fi
  timeout 600 python $root/bench.py --workload $wl --no-cpu-baseline --no-configs --steps $steps --warmup 5 > $root/gpurun_out/ab/${wl}_$name.json 2> $root/gpurun_out/ab/${wl}_$name.err
  python - "$root/gpurun_out/ab/${wl}_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[2], "ms/step %.4f cold %.4f" % (d["ms_per_step"], d.get("ms_per_step_cold", 0)), [(k["name"][:22], round(k["avg_us"], 1)) for k in d["kernels"]])
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
done
