#!/bin/bash
# PMC counters of the boundary smoothing kernels on hex100B (one pass per counter group; kernel-trace only).
# Counter collection serialises kernels: a stream that waits for a value written behind a kernel of ANOTHER stream never
# wakes up (observed: 25 minutes until the outer limit), so the side streams are switched off for these passes and every
# pass has its own time limit.
export SMGPU_SIDE_STREAM=0   # (the engine also does this by itself when it sees ROCPROF_COUNTER_COLLECTION=1)
root=${GRAFT_REPO_ROOT:-$(pwd)}
wl=${1:-hex100B}
out=$root/gpurun_out/pmc_$wl
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 $root/bench.py --workload $wl --steps 20 --warmup 2 --no-cpu-baseline --no-configs > $out/p$i.log 2>&1
done
cd $root
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = k.split("(")[0].split("::")[-1].split("<")[0]
        if name.startswith("k_"):
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in acc.items():
    print(name)
    for c, v in sorted(d.items()):
        print("   %-28s mean %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
