"""usage: check_bench_line.py <file with bench.py's stdout> <nGPUs>: exit 1 unless the JSON line is one a judge can accept --
n_gpus as asked, parity_check ok (copies of shared points identical across ranks, the down-scaled case equal to the oracle's
MultiDomain through the same transport), the records really travelled over RCCL with all ranks in the communicator, and the same
for every configs[] entry (BASELINE configs[4] rides there)."""
import json
import sys

line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if len(line) != 1:
    sys.exit(f"expected one JSON line, found {len(line)}")
d = json.loads(line[0])
n = int(sys.argv[2])
bad = []


def check(e, what):
    if e.get("n_gpus") != n:
        bad.append(f"{what}: n_gpus {e.get('n_gpus')} != {n}")
    pc = e.get("parity_check") or {}
    if not pc.get("ok"):
        bad.append(f"{what}: parity_check not ok: {json.dumps(pc)[:600]}")
    rc = e.get("rccl") or {}
    if rc.get("ranks_seen") != n:
        bad.append(f"{what}: communicator saw {rc.get('ranks_seen')} ranks")
    if "RCCL" not in rc.get("backend", "") or rc.get("transport") not in ("direct", "push", "torch"):
        bad.append(f"{what}: records did not travel over RCCL: {rc}")
    if rc.get("transport") == "torch":
        print(f"note: {what}: torch all_to_all_single was used ({rc.get('self_check')})")
    if not (e.get("cpu_baseline") or {}).get("value"):
        bad.append(f"{what}: no cpu_baseline")
    print(f"{what}: {e['value'] / 1e9:.2f} G points/s, {e['ms_per_step']:.4f} ms/step, transport {rc.get('transport')}, parity ok {pc.get('ok')}")


check(d, "headline")
for c in d.get("configs", []):
    if "error" in c:
        bad.append(f"configs[{c.get('workload')}]: {c['error']}")
    else:
        check(c, f"configs[{c['workload']}]")
if n > 1 and not d.get("configs"):
    bad.append("no configs[] entry (BASELINE configs[4])")
if bad:
    print("\n".join(bad))
    sys.exit(1)
