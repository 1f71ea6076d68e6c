run() { # lib workload steps
  if [ -n "$1" ]; then export SMOOTHMESH_SMGPU_LIB=$1; else unset SMOOTHMESH_SMGPU_LIB; fi
  timeout 300 python bench.py --no-configs --workload $2 --steps $3 --warmup 10 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', '${1:-main}', d['ms_per_step'], d['roofline']['avg_launch_us'], [ (k,v.get('avg_launch_us')) for k,v in d.items() if k.startswith('roofline_')])"
}
V=smoothmesh_amd/csrc/variants/libsmgpu_head.so
for rep in 1 2; do for v in "" $V; do run "$v" hex100 300; done; done
for w in hex215 cavity215; do for v in "" $V; do run "$v" $w 100; done; done
unset SMOOTHMESH_SMGPU_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_multirank.py -m gpu -x -q 2>&1 | tail -3
