"""bench.py --gpus N starts its own ranks (one process per GPU through torch.distributed.run) when no launcher did."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=900):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout)


def test_self_launch_reaches_the_ranks_without_a_gpu():
    """no GPU here: the two ranks must have been started (each refuses loudly), nothing falls back to a CPU path"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test below")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "hex6"], timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("no GPU visible") >= 1
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["hex16", "cavity12c"])
def test_bench_two_ranks_self_launched(workload):
    """`python bench.py --gpus 2` as the driver calls it (no WORLD_SIZE): two ranks on this box's one GPU through the gloo debug
    transport; ONE JSON line with n_gpus = 2 and both ranks' points"""
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", workload],
             env={"SMOOTHMESH_BACKEND": "gloo", "SMOOTHMESH_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["points_per_gpu"] > 0 and "roofline" in d
    if workload.startswith("cavity"):
        assert "15^3 base grid" in d["config"]["workload"] and "configs[4]" in d["config"]["workload"]


@pytest.mark.gpu
def test_bench_single_gpu_line_carries_the_other_configs():
    r = _run(["--steps", "5", "--warmup", "1", "--workload", "hex12", "--configs", "hex12c,cavity10c", "--config-steps", "3"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["cpu_baseline"]["kind"] == "port" and d["roofline"]["frac"] > 0
    assert [c["workload"] for c in d["configs"]] == ["hex12c", "cavity10c"]
    assert d["parity_check"]["ok"] and d["parity_check"]["rel_linf"] <= 1e-10 and d["ms_per_step_cold"] > 0
    for c in d["configs"]:
        assert "error" not in c, c
        assert c["ms_per_step"] > 0 and c["roofline"]["frac"] > 0 and c["kernels"]
        # every configuration is compared with the oracle on its own mesh by the same run and carries its own CPU baseline
        assert c["parity_check"]["ok"] and c["parity_check"]["nFrozen_equal"] and c["parity_check"]["rel_linf"] <= 1e-10
        assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["value"] > 0
        # no impossible bandwidths in the per-kernel table (filter + exact kernels are accounted as one unit)
        assert all(k["algo_GBps"] is None or k["algo_GBps"] < 8000 for k in c["kernels"])
