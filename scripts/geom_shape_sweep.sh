#!/bin/bash
# usage (GPU box): scripts/geom_shape_sweep.sh <workload> <steps>   -- k_geom_tile against the tile shape: cells per tile (the tiles
# are bricks along the Morton curve; faces on a tile's surface are computed by both neighbours), with the duplication factor of the
# staged faces / points printed beside the kernel time (SMGPU_VERBOSE=1), LDS per tile and the time of the whole iteration.
export SMOOTHMESH_BENCH_FULL=1   # bench.py prints its full document (not the compact driver line) on stdout
wl=${1:-hex100}; steps=${2:-50}
root=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "SMGPU_GEOM_CELLS=64" "SMGPU_GEOM_CELLS=96" "SMGPU_GEOM_CELLS=128" "SMGPU_GEOM_CELLS=160 SMGPU_GEOM_CAPF=768 SMGPU_GEOM_CAPP=1100" "SMGPU_GEOM_CELLS=192 SMGPU_GEOM_CAPF=768 SMGPU_GEOM_CAPP=1300" "SMGPU_GEOM_CELLS=256 SMGPU_GEOM_CAPF=1024 SMGPU_GEOM_CAPP=1400" "SMGPU_GEOM_T=128 SMGPU_GEOM_CELLS=64" "SMGPU_GEOM_T=128 SMGPU_GEOM_CELLS=128 SMGPU_GEOM_CAPF=512 SMGPU_GEOM_CAPP=768"; do
  echo "== $cfg"
  env $cfg SMGPU_VERBOSE=1 timeout 600 python $root/bench.py --no-cpu-baseline --no-configs --workload $wl --steps $steps --warmup 5 2>&1 | grep -E "^\{|\[smgpu\] tiles" | python $root/scripts/bench_summary.py 2>/dev/null | grep -E "tiles:|ms/step|k_geom_tile"
done
