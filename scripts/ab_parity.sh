#!/bin/bash
# usage (on the GPU box): scripts/ab_parity.sh <workload> <steps> variant.so ...: like ab_bench.sh, but WITH the oracle leg -- a variant that
# is faster and wrong must say so (parity_check of every line is printed)
export SMOOTHMESH_BENCH_FULL=1
wl=$1; steps=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/ab
for v in default "$@"; do
  name=$(basename "$v" .so)
  if [ "$v" = default ]; then unset SMOOTHMESH_SMGPU_LIB; else export SMOOTHMESH_SMGPU_LIB=$root/$v; fi
  timeout 900 python $root/bench.py --workload $wl --no-configs --steps $steps --warmup 5 > $root/gpurun_out/ab/${wl}_$name.json 2> $root/gpurun_out/ab/${wl}_$name.err
  python - "$root/gpurun_out/ab/${wl}_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    pc = d.get("parity_check") or {}
    print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], "parity ok", pc.get("ok"), "bitwise", pc.get("bitwise_equal"), "iters", pc.get("iters"),
          [(k["name"][:22], round(k["avg_us"], 1)) for k in d["kernels"][:6]])
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
done
