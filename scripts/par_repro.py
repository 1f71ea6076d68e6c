import os, sys, subprocess, tempfile, time, re
sys.path.insert(0, os.getcwd())
from smoothmesh_amd.polymesh import cavity_subdomain, write_decomposed_case
BIN = os.path.join(os.getcwd(), "smoothmesh_amd", "bin", "smoothMesh")
grid = (2, 2, 1)
subs = [cavity_subdomain(12, grid, r, jitter=0.2, seed=4) for r in range(4)]
LINE = re.compile(r"Iteration (\d+): nFrozenPoints=(\d+)")
ref = None
for t in range(int(sys.argv[1])):
    d = tempfile.mkdtemp()
    write_decomposed_case(d, subs, binary=True, writeFormat="binary")
    t0 = time.time()
    r = subprocess.run([BIN, "-case", d, "-parallel", "-centroidalIters", "9", "-relTol", "0", "-writeInterval", "100"], capture_output=True, text=True, timeout=600)
    dt = time.time() - t0
    frz = re.findall(r"nFrozenPoints[ =:]+(\d+)", r.stdout) or re.findall(r"frozen[^0-9]*(\d+)", r.stdout)
    if ref is None: ref = frz
    print(t, "rc", r.returncode, "%.1fs" % dt, "same" if frz == ref else "DIFF", frz[:10], flush=True)
    for l in (r.stdout + r.stderr).splitlines():
        if "[smgpu]" in l or "rror" in l: print("   ", l[:200])
