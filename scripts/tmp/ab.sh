run() { # envspec workload steps
  env $1 timeout 300 python bench.py --no-configs --workload $2 --steps $3 --warmup 10 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', '$1', d['ms_per_step'])"
}
for w in hex100c cavity215c cavity100c; do for v in SMGPU_FA_SIDE_EXACT=1 SMGPU_FA_SIDE_EXACT=0; do run "$v" $w 100; done; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
