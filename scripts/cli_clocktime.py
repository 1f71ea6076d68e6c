#!/usr/bin/env python3
"""The tool a user runs, timed end to end at BASELINE size: write the cavity<N> case to disk (ascii and
binary), run `smoothMesh -case ... -centroidalIters 200 -relTol 0 -minAngle 35 -maxAngle 160` (BASELINE configs[3]) and print its
`ClockTime breakdown` line beside the wall time of the whole process; with --serial-io also with SMHOST_IO_THREADS=1 (the
round-5 readers / writers).  usage: cli_clocktime.py [N=215] [--iters 200] [--serial-io] [--dir /tmp/cli_case]"""
import argparse
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "smoothmesh_amd", "bin", "smoothMesh")

ap = argparse.ArgumentParser()
ap.add_argument("n", nargs="?", type=int, default=215)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--serial-io", action="store_true")
ap.add_argument("--dir", default="/tmp/cli_case")
ap.add_argument("--formats", default="ascii,binary")
ap.add_argument("--timeline", action="store_true", help="SMOOTHMESH_TIMELINE=1 SMGPU_VERBOSE=2 SMHOST_IO_VERBOSE=1: the stages' stderr lines")
args = ap.parse_args()

from smoothmesh_amd.polymesh import cavity_mesh, write_case  # noqa: E402

t0 = time.perf_counter()
mesh = cavity_mesh(args.n, jitter=0.2, seed=12345)
print(f"# cavity{args.n}: {mesh.nPoints} points, {mesh.nCells} cells, {len(mesh.owner)} faces, generated in {time.perf_counter() - t0:.1f} s", flush=True)
for fmt in args.formats.split(","):
    case = os.path.join(args.dir, fmt)
    shutil.rmtree(case, ignore_errors=True)
    t0 = time.perf_counter()
    write_case(case, mesh, binary=(fmt == "binary"), writeFormat=fmt, precision=17)
    d = os.path.join(case, "constant", "polyMesh")
    mb = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) / 1e6
    print(f"# {fmt}: constant/polyMesh written in {time.perf_counter() - t0:.1f} s, {mb:.0f} MB", flush=True)
    for env_io in ([None, "1"] if args.serial_io else [None]):
        env = dict(os.environ)
        if env_io:
            env["SMHOST_IO_THREADS"] = env_io
        if args.timeline:
            env.update(SMOOTHMESH_TIMELINE="1", SMGPU_VERBOSE="2", SMHOST_IO_VERBOSE="1")
        for d_ in os.listdir(case):
            if d_.isdigit() and d_ != "0":
                shutil.rmtree(os.path.join(case, d_))
        t0 = time.perf_counter()
        r = subprocess.run([BIN, "-case", case, "-centroidalIters", str(args.iters), "-relTol", "0", "-minAngle", "35", "-maxAngle", "160"],
                           capture_output=True, text=True, env=env)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            print(r.stdout[-1500:], r.stderr[-1500:])
            raise SystemExit(f"smoothMesh failed on the {fmt} case")
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("ClockTime") or "Smoothing iteration=%d " % args.iters in ln]
        out_dir = os.path.join(case, str(args.iters), "polyMesh")
        omb = sum(os.path.getsize(os.path.join(out_dir, f)) for f in os.listdir(out_dir)) / 1e6
        print(f"{fmt:6s} io_threads={'default' if not env_io else env_io:7s} process wall {wall:.2f} s, points written {omb:.0f} MB")
        for ln in lines:
            print("       " + ln)
        if args.timeline:
            print("\n".join("       | " + ln for ln in r.stderr.splitlines()))
        sys.stdout.flush()
shutil.rmtree(args.dir, ignore_errors=True)
