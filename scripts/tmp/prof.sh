root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_probe
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -o p -- python3 $root/scripts/probe_rank_of_8.py > $out/log.txt 2>&1
python3 - <<'PY' > $out/timeline.txt
import csv,glob,os
f=glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/prof_probe/t/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find a steady-state window: take kernels 3000..3060
n=len(rows)
print(n,'kernels')
i0=n//3
prev=None
for r in rows[i0:i0+40]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap=(s-prev)/1000 if prev else 0
    print(f"{r['Kernel_Name'][:60]:60s} dur {(e-s)/1000:7.1f} us gap {gap:7.1f} us  q {r.get('Queue_Id','')}")
    prev=e
PY
rm -rf $out/t
