#!/bin/bash
# usage (on the GPU box): scripts/ab_env.sh <workload> <steps> "VAR=val VAR2=val" ["..." ...]
# one bench.py line per environment setting (the first argument set may be "" = defaults): A/B of run-time knobs
export SMOOTHMESH_BENCH_FULL=1   # bench.py prints its full document (not the compact driver line) on stdout
wl=$1; steps=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/ab
i=0
for e in "$@"; do
  i=$((i+1))
  out=$root/gpurun_out/ab/${wl}_env$i
  env $e timeout 600 python $root/bench.py --workload $wl --no-cpu-baseline --no-configs --no-parity --steps $steps --warmup 5 > $out.json 2> $out.err
  python - "$out.json" "$e" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print("[%s]" % sys.argv[2], "ms/step %.4f" % d["ms_per_step"], [(k["name"][:22], round(k["avg_us"], 1)) for k in d["kernels"]])
except Exception as ex:
    print("[%s]" % sys.argv[2], "FAILED", ex)
PY
done
