#!/bin/bash
# run on the GPU box: kernel timeline (rocprofv3 --kernel-trace) of the one-rank-of-eight probe, plain and with boundary point
# smoothing -> gpurun_out/probe_timeline/{probe.txt,timeline.txt,timeline_boundary.txt}: per kernel start-to-end duration and the
# idle gap in front of it, two iterations each (copy to profiles/<round>/ what is to be judged)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/probe_timeline
rm -rf $out; mkdir -p $out
cd $root
timeout 600 python3 scripts/probe_rank_of_8.py --boundary 2>&1 | grep -E "rank 0 of 8|inorder|overlap|serial|boundary" > $out/probe.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/t -o p -- python3 $root/scripts/probe_rank_of_8.py --boundary > /dev/null 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/t/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def dump(name, anchor, skip):
    idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
    i0 = idx[skip]
    prev = None
    with open(out + '/' + name, 'w') as o:
        n = 0
        for r in rows[i0:]:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            gap = (s - prev) / 1000 if prev else 0.0
            o.write(f"{r['Kernel_Name'].split('(')[0][:60]:60s} dur {(e - s) / 1000:7.1f} us   idle before {max(gap, 0.0):6.1f} us\n")
            prev = e
            n += 1
            if n >= 24: break
first_bnd = next(i for i, r in enumerate(rows) if 'k_bnd_fix' in r['Kernel_Name'])
plain = [i for i, r in enumerate(rows[:first_bnd]) if 'k_shared_fix' in r['Kernel_Name']]
rows_plain = rows[:first_bnd]
# plain probe: a window in the middle of the first (in-order) loop
i0 = plain[len(plain) // 4]
prev = None
with open(out + '/timeline.txt', 'w') as o:
    for r in rows_plain[i0:i0 + 22]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev) / 1000 if prev else 0.0
        o.write(f"{r['Kernel_Name'].split('(')[0][:60]:60s} dur {(e - s) / 1000:7.1f} us   idle before {max(gap, 0.0):6.1f} us\n")
        prev = e
dump('timeline_boundary.txt', 'k_bnd_fix', 40)
PY
rm -rf $out/t
