#!/bin/bash
# run on the GPU box: absolute kernel timeline (rocprofv3 --kernel-trace: start / end of every kernel relative to the window's
# first, all streams) of the one-rank-of-eight probe -- a window in the in-order loop and one in the exchange-stream loop
# -> gpurun_out/probe_timeline2/{inorder,overlap}.txt.  $1: extra arguments for probe_rank_of_8.py
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/probe_timeline2
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/t -o p -- python3 $root/scripts/probe_rank_of_8.py $1 > $out/probe.txt 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/t/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
g = [i for i, r in enumerate(rows) if 'k_geom_halo' in r['Kernel_Name'] or 'k_geom_tile' in r['Kernel_Name']]
def dump(name, at):
    if at >= len(g): return
    i0 = g[at]
    t0 = int(rows[i0]['Start_Timestamp'])
    with open(out + '/' + name, 'w') as o:
        for r in rows[i0:i0 + 26]:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            o.write(f"{(s - t0) / 1000:8.1f} .. {(e - t0) / 1000:8.1f}  ({(e - s) / 1000:6.1f} us)  {r['Kernel_Name'].split('(')[0][:70]}\n")
dump('inorder.txt', 100)
dump('overlap.txt', 230)
PY
rm -rf $out/t
