#!/bin/bash
# What a launch of each kernel EXECUTES, by instruction type (the dynamic counterpart of scripts/isa_mix.py): wave instructions per
# launch from the SQ_INSTS_* counters, one rocprofv3 pass per group of four (kernel trace only; side streams off, see pmc_kernels.sh).
#   usage: scripts/pmc_inst_mix.sh [workload] [steps]     ->  gpurun_out/inst_mix_<workload>/summary.txt
export SMGPU_SIDE_STREAM=0
root=${GRAFT_REPO_ROOT:-$(pwd)}
wl=${1:-hex100}
steps=${2:-20}
out=$root/gpurun_out/inst_mix_$wl
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 $root/bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --no-configs > $out/p$i.log 2>&1
done
cd $root
python3 - > $out/summary.txt <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = k.split("(")[0].replace("void ", "").replace("smgpu::", "")
        if name.startswith("k_"):
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    valu = m.get("SQ_INSTS_VALU", 0.0)
    if valu <= 0:
        continue
    f64 = sum(m.get("SQ_INSTS_VALU_" + t + "_F64", 0.0) for t in ("ADD", "MUL", "FMA", "TRANS"))
    f32 = sum(m.get("SQ_INSTS_VALU_" + t + "_F32", 0.0) for t in ("ADD", "MUL", "FMA", "TRANS"))
    i32 = m.get("SQ_INSTS_VALU_INT32", 0.0) + m.get("SQ_INSTS_VALU_INT64", 0.0)
    cvt = m.get("SQ_INSTS_VALU_CVT", 0.0)
    other = valu - f64 - f32 - i32 - cvt
    tot = valu + m.get("SQ_INSTS_SALU", 0.0) + m.get("SQ_INSTS_LDS", 0.0) + m.get("SQ_INSTS_VMEM_RD", 0.0) + m.get("SQ_INSTS_VMEM_WR", 0.0) + m.get("SQ_INSTS_SMEM", 0.0) + m.get("SQ_INSTS_BRANCH", 0.0)
    print(name)
    print("   wave instructions per launch: total %.3g = VALU %.3g (%.0f%%) + SALU %.3g + LDS %.3g + VMEM rd %.3g wr %.3g + SMEM %.3g + branch %.3g   [waves %.0f]" % (
        tot, valu, 100 * valu / tot, m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_VMEM_RD", 0), m.get("SQ_INSTS_VMEM_WR", 0), m.get("SQ_INSTS_SMEM", 0),
        m.get("SQ_INSTS_BRANCH", 0), m.get("SQ_WAVES", 0)))
    print("   VALU: f64 arithmetic %.3g (%.0f%%: add %.3g mul %.3g fma %.3g trans %.3g)  int %.3g (%.0f%%)  cvt %.3g  f32 %.3g  moves / selects / compares / rest %.3g (%.0f%%)" % (
        f64, 100 * f64 / valu, m.get("SQ_INSTS_VALU_ADD_F64", 0), m.get("SQ_INSTS_VALU_MUL_F64", 0), m.get("SQ_INSTS_VALU_FMA_F64", 0), m.get("SQ_INSTS_VALU_TRANS_F64", 0),
        i32, 100 * i32 / valu, cvt, f32, other, 100 * other / valu))
    if m.get("SQ_BUSY_CYCLES"):
        print("   SQ_ACTIVE_INST_VALU %.3g  SQ_BUSY_CYCLES %.3g  SQ_WAVE_CYCLES %.3g" % (m.get("SQ_ACTIVE_INST_VALU", 0), m.get("SQ_BUSY_CYCLES", 0), m.get("SQ_WAVE_CYCLES", 0)))
PY
cat $out/summary.txt
