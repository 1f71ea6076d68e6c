#!/bin/bash
# usage (on the GPU box): scripts/ab_env_parity.sh <workload> <steps> "ENV=.. ENV2=.." ...: one bench.py line per environment (first: defaults),
# WITH the oracle leg (parity_check printed): A/B of run-time knobs where a wrong result must not pass as a fast one
export SMOOTHMESH_BENCH_FULL=1
wl=$1; steps=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/ab
i=0
for e in "" "$@"; do
  i=$((i+1))
  env $e timeout 900 python $root/bench.py --workload $wl --no-configs --steps $steps --warmup 5 > $root/gpurun_out/ab/${wl}_env$i.json 2> $root/gpurun_out/ab/${wl}_env$i.err
  python - "$root/gpurun_out/ab/${wl}_env$i.json" "${e:-defaults}" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    pc = d.get("parity_check") or {}
    print(sys.argv[2], "| ms/step %.4f" % d["ms_per_step"], "parity ok", pc.get("ok"), "bitwise", pc.get("bitwise_equal"), "iters", pc.get("iters"),
          [(k["name"][:22], round(k["avg_us"], 1)) for k in d["kernels"][:6]])
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
done
