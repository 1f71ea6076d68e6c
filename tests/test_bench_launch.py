"""bench.py --gpus N starts its own ranks (one process per GPU through torch.distributed.run) when no launcher did."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=900, detail=None):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    if detail is not None:
        e["SMOOTHMESH_BENCH_DETAIL"] = str(detail)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout)


def _compact(r):
    """stdout carries exactly ONE JSON line, it is the last line, shorter than 4 KB and strict JSON"""
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0]
    assert len(lines[0].encode()) < 4096

    def no_const(x):
        raise ValueError(f"non-strict JSON constant {x}")
    return json.loads(lines[0], parse_constant=no_const)


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
@pytest.mark.parametrize("workload,configs", [("hex16", "cavity10c"), ("cavity12c", "")])
def test_bench_two_ranks_self_launched(workload, configs, tmp_path):
    """`python bench.py --gpus 2` as the driver calls it (no WORLD_SIZE): two ranks on this box's one GPU through the gloo debug
    transport; ONE JSON line with n_gpus = 2 and both ranks' points, and everything a judge needs to accept an N > 1 line:
    parity_check (copies of shared points identical across the ranks after the timed steps; a down-scaled case through the same
    transport against the oracle's MultiDomain), cpu_baseline (rank 0's oracle on its own sub-domain), how the records travelled
    (transport, communicator size, self-check), and configs[] = the polyhedral constraints-on workload (BASELINE configs[4]) beside
    the hex headline"""
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", workload, "--configs", configs, "--config-steps", "3"],
             env={"SMOOTHMESH_BACKEND": "gloo", "SMOOTHMESH_SHARE_GPU": "1"}, detail=tmp_path / "detail.json")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    c_ = _compact(r)
    assert c_["n_gpus"] == 2 and c_["steps"] == 4 and c_["parity_check"]["ok"] and c_["rccl"]["ranks_seen"] == 2 and c_["cpu_baseline"]["value"] > 0
    d = json.load(open(tmp_path / "detail.json"))          # the full document
    assert d["value"] == pytest.approx(c_["value"], rel=1e-4)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["points_per_gpu"] > 0 and "roofline" in d
    if workload.startswith("cavity"):
        assert "15^3 base grid" in d["config"]["workload"] and "configs[4]" in d["config"]["workload"]

    def judgeable(e):
        pc = e["parity_check"]
        assert pc["ok"] is True
        a, b = pc["shared_point_copies"], pc["small_case"]
        assert a["ok"] and a["mismatching_points"] == 0 and a["shared_points"] > 0 and a["copies"] >= 2 * a["shared_points"]
        assert b["ok"] and b["rel_linf_max_over_ranks"] <= 1e-10 and b["bitwise_equal_on_every_rank"] and "MultiDomain" in b["against"]
        cb = e["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "RANK 0's sub-domain" in cb["sample"]
        assert e["speedup_vs_cpu_baseline"] > 0
        # the cost of the multi-rank code path against the rank's sub-domain as a serial mesh, and the bound it puts on weak scaling
        assert e["halo_cost"]["serial_ms_per_step"] > 0 and 0 < e["weak_efficiency_bound"] < 1.5 and e["halo_overhead_us"] == e["halo_cost"]["halo_overhead_us"]
        rc = e["rccl"]
        assert rc["ranks_seen"] == 2 and rc["transport"] in ("direct", "torch", "push") and rc["self_check"]
        assert rc["transport"] == b["transport"]            # the small case went through the transport of the timed run
        # gloo here: the line must SAY that this was not RCCL
        assert rc["transport"] == "torch" and "gloo" in rc["backend"] and rc["self_check"].startswith("off: process group backend gloo")

    judgeable(d)
    if configs:
        (c,) = d["configs"]
        assert c["workload"] == "cavity10c" and c["n_gpus"] == 2 and c["steps"] == 3 and c["value"] > 0
        assert "configs[4]" in c["config"] and "constraints on" in c["config"] and c["nFrozenPoints_last"] > 0
        judgeable(c)
    else:
        assert "configs" not in d


def test_default_multi_gpu_line_names_configs4():
    """bench.py --gpus 8 without --workload: the hex100 weak-scaling headline AND, as configs[], cavity215c per GPU = the
    430^3-base (~80 M-cell) polyhedral mesh, constraints on, 200 iterations (BASELINE configs[4]) -- read off the source, since
    neither eight GPUs nor a GPU at all exist where this test runs"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '(["cavity215c"] if args.workload == "hex100" else [])' in src
    assert 'args.config_steps or (100 if k_ == "hex" else 200)' in src
    sys.path.insert(0, ROOT)
    import bench
    assert int(round(215 * 8 ** (1.0 / 3.0))) == 430 and bench.proc_grid(8) == (2, 2, 2)
    assert "configs[4]" in bench.workload_text("cavity", 215, True, False, False, world=8, n_global=430)


def _line(**kw):
    par = {"ok": True, "bitwise_equal": True, "rel_linf": 0.0}
    d = {"metric": "m", "workload_name": "hex100", "value": 1e10, "steps": 100, "ms_per_step": 0.08, "parity_check": dict(par),
         "speedup_vs_cpu_baseline": 3000.0,
         "configs": [{"workload": "hex100c", "steps": 100, "ms_per_step": 0.2, "value": 5e9, "parity_check": dict(par), "speedup_vs_cpu_baseline": 4000.0},
                     {"workload": "cavity215c", "steps": 200, "ms_per_step": 3.0, "value": 3e9, "parity_check": dict(par), "speedup_vs_cpu_baseline": 5000.0}]}
    d.update(kw)
    return d


def test_finalize_line_fails_closed_and_is_tail_safe():
    """the line's LAST key is the compact per-configuration summary (a reader of the tail sees configs[1..3]); a parity_check
    that is not ok, or a configs[] entry with an error, turns into a non-zero exit code"""
    sys.path.insert(0, ROOT)
    import bench
    out, code = bench.finalize_line(_line())
    assert code == 0 and out["exit_code"] == 0 and "failures" not in out
    assert list(out)[-1] == "configs_summary"
    assert [c["workload"] for c in out["configs_summary"]] == ["hex100", "hex100c", "cavity215c"]
    assert all(c["parity_ok"] is True and c["ms_per_step"] > 0 and c["points_per_s"] > 0 and c["cpu_ratio"] > 0 for c in out["configs_summary"])
    assert len(json.dumps(out["configs_summary"])) < 1200          # stays inside a short tail
    # a falsified parity object in a configs[] entry
    bad = _line()
    bad["configs"][1]["parity_check"]["ok"] = False
    out, code = bench.finalize_line(bad)
    assert code == 1 and out["failures"] == ["cavity215c: parity_check not ok"] and out["configs_summary"][2]["parity_ok"] is False
    # ... in the headline
    bad = _line()
    bad["parity_check"] = {"ok": False}
    assert bench.finalize_line(bad)[1] == 1
    # a sub-run that raised
    bad = _line()
    bad["configs"][0] = {"workload": "hex100c", "error": "RuntimeError: boom"}
    out, code = bench.finalize_line(bad)
    assert code == 1 and "hex100c: RuntimeError: boom" in out["failures"] and out["configs_summary"][1]["error"].endswith("boom")
    assert list(out)[-1] == "configs_summary"
    # a line without an oracle leg (--no-cpu-baseline) is not a failure, and says that nothing was compared
    out, code = bench.finalize_line({"workload_name": "hex100", "value": 1.0, "steps": 1, "ms_per_step": 1.0})
    assert code == 0 and out["configs_summary"][0]["parity_ok"] is None


def test_exit_path_carries_a_falsified_parity_object():
    """through the real exit path (emit -> SystemExit) in a process of its own: the line is printed AND the exit code is 1"""
    prog = ("import sys, json; sys.path.insert(0, %r); import bench; sys.path.insert(0, %r); from test_bench_launch import _line; "
            "raise SystemExit(bench.emit(_line()))" % (ROOT, os.path.join(ROOT, "tests")))
    ok = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=dict(os.environ, SMOOTHMESH_BENCH_FALSIFY="", SMOOTHMESH_BENCH_DETAIL=""))
    assert ok.returncode == 0 and _compact(ok)["exit_code"] == 0
    bad = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=dict(os.environ, SMOOTHMESH_BENCH_FALSIFY="cavity215c", SMOOTHMESH_BENCH_DETAIL=""))
    d = _compact(bad)
    assert bad.returncode == 1 and d["exit_code"] == 1 and d["failures"] == ["cavity215c: parity_check not ok"]
    assert list(d)[-1] == "configs_summary"
    # the full document went to stderr, one line, for a reader without the detail file
    full = [l for l in bad.stderr.splitlines() if l.startswith("BENCH_DETAIL {")]
    assert len(full) == 1 and json.loads(full[0][len("BENCH_DETAIL "):])["configs"][1]["parity_check"]["falsified_by_test_hook"]


def test_compact_line_is_short_strict_and_complete():
    """what the driver parses: the LAST stdout line, < 4096 bytes, strict JSON, carrying the contract's keys, the dominant kernel's
    roofline (with the ceiling that binds it), the centroid-gather kernel's fractions, cpu_baseline, parity and one short object per
    configuration (round 5's 20 KB line came back with parsed = null)"""
    sys.path.insert(0, ROOT)
    import bench
    roof = {"kernel": "k_geom_tile", "avg_launch_us": 47.7, "bound": "hbm", "achieved": 2877.9, "peak": 8000.0, "unit": "GB/s", "frac": 0.3597,
            "algorithmic_bytes_per_launch": 137327232, "valu_f64": {"frac": 0.5769}, "traffic": 112907499, "note": "x" * 900}
    gather = {"kernel": "k_smooth<final>", "avg_launch_us": 33.7, "bound": "hbm", "frac": 0.5156, "achieved": 4125.0,
              "accountings": {"fused (this kernel's algorithmic bytes)": {"frac": 0.5156, "achieved_GBps": 4125.0},
                              "K_cg only (SURVEY 8d)": {"frac": 0.4097, "achieved_GBps": 3277.0}}}
    big = _line(n_gpus=1, warmup=5, unit="points/s", dtype="f64", data="synthetic", scaling="weak", higher_is_better=True, vs_baseline=None,
                ms_per_step_cold=float("nan"), roofline=roof, roofline_centroid_gather=gather,
                config={"workload": "w" * 2000, "points_per_gpu": 1030301, "cells_per_gpu": 1000000, "parallelism": "1 GPU"},
                cpu_baseline={"value": 3.5e6, "unit": "points/s", "cores": 1, "kind": "port", "sample": "s" * 3000, "host_cpus": 256},
                kernels=[{"name": "k", "avg_us": 1.0}] * 400)
    for c in big["configs"]:
        c["roofline"] = dict(roof)
        c["roofline_centroid_gather"] = dict(gather)
    big["configs"] += [dict(big["configs"][0], workload=f"extra{i}") for i in range(3)]
    out, code = bench.finalize_line(big)
    line = bench.compact_line(out, "bench_detail.json")
    assert len(line.encode()) < 4096 and "\n" not in line

    def no_const(x):
        raise ValueError(x)
    d = json.loads(line, parse_constant=no_const)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "roofline_centroid_gather", "cpu_baseline", "parity_check", "exit_code", "configs_summary"):
        assert k in d, k
    assert d["ms_per_step_cold"] is None                      # NaN never reaches the line
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["binding"] == "valu_f64" and r["frac"] == pytest.approx(0.3597) and r["valu_f64_frac"] == pytest.approx(0.5769)
    assert r["traffic"] == 112907499 and r["algorithmic_bytes_per_launch"] == 137327232 and r["peak"] == 8000.0
    g = d["roofline_centroid_gather"]
    assert g["frac_K_cg"] == pytest.approx(0.4097) and g["frac_fused"] == pytest.approx(0.5156)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and len(d["cpu_baseline"]["sample"]) <= 200
    assert d["parity_check"] == {"ok": True, "bitwise_equal": True, "iters": None, "rel_linf": 0.0, "tolerance": 1e-10,
                                 "against": "CPU oracle (unpinned restatement of the reference)"}
    assert [c["workload"] for c in d["configs_summary"]][:3] == ["hex100", "hex100c", "cavity215c"]
    assert all(c["dominant_kernel"] == "k_geom_tile" and c["roofline_frac"] > 0 and c["binding"] == "valu_f64" for c in d["configs_summary"])


@pytest.mark.gpu
def test_bench_exits_nonzero_when_a_comparison_fails(tmp_path):
    r = _run(["--steps", "3", "--warmup", "1", "--workload", "hex10", "--configs", "hex10c", "--config-steps", "2"],
             env={"SMOOTHMESH_BENCH_FALSIFY": "hex10c"}, detail=tmp_path / "detail.json")
    d = _compact(r)
    assert r.returncode == 1 and d["exit_code"] == 1 and d["failures"] == ["hex10c: parity_check not ok"]
    assert d["parity_check"]["ok"] and list(d)[-1] == "configs_summary"
    assert d["configs_summary"][1]["parity_ok"] is False


@pytest.mark.gpu
def test_bench_single_gpu_line_carries_the_other_configs(tmp_path):
    r = _run(["--steps", "5", "--warmup", "1", "--workload", "hex12", "--configs", "hex12c,cavity10c", "--config-steps", "3"],
             detail=tmp_path / "detail.json")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    c_ = _compact(r)
    # what the driver's parsed record will hold
    assert c_["n_gpus"] == 1 and c_["steps"] == 5 and c_["warmup"] == 1 and c_["value"] > 0 and c_["dtype"] == "f64"
    assert c_["roofline"]["frac"] > 0 and c_["roofline"]["bound"] in ("hbm", "valu_f64") and c_["roofline"]["binding"]
    assert c_["cpu_baseline"]["kind"] == "port" and c_["cpu_baseline"]["cores"] == 1 and c_["cpu_baseline"]["value"] > 0
    assert c_["parity_check"]["ok"] and c_["parity_check"]["bitwise_equal"] and c_["detail"]
    assert [c["workload"] for c in c_["configs_summary"]] == ["hex12", "hex12c", "cavity10c"]
    assert all(c["dominant_kernel"] and c["roofline_frac"] > 0 for c in c_["configs_summary"])
    assert all(c["near_ties"] == 0 for c in c_["configs_summary"])      # no decision of these runs hung on a last bit of an angle
    d = json.load(open(tmp_path / "detail.json"))          # the full document
    assert d["n_gpus"] == 1 and d["cpu_baseline"]["kind"] == "port" and d["roofline"]["frac"] > 0
    assert [c["workload"] for c in d["configs"]] == ["hex12c", "cavity10c"]
    # tail-safe: the compact summary of every configuration is the line's last key
    assert list(d)[-1] == "configs_summary" and d["exit_code"] == 0
    assert [c["workload"] for c in d["configs_summary"]] == ["hex12", "hex12c", "cavity10c"]
    assert all(c["parity_ok"] and c["bitwise_equal"] and c["cpu_ratio"] > 0 for c in d["configs_summary"])
    assert d["parity_check"]["ok"] and d["parity_check"]["rel_linf"] <= 1e-10 and d["ms_per_step_cold"] > 0
    for c in d["configs"]:
        assert "error" not in c, c
        assert c["ms_per_step"] > 0 and c["roofline"]["frac"] > 0 and c["kernels"]
        # every configuration is compared with the oracle on its own mesh by the same run and carries its own CPU baseline
        assert c["parity_check"]["ok"] and c["parity_check"]["nFrozen_equal"] and c["parity_check"]["rel_linf"] <= 1e-10
        assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["value"] > 0
        # no impossible bandwidths in the per-kernel table (filter + exact kernels are accounted as one unit)
        assert all(k["algo_GBps"] is None or k["algo_GBps"] < 8000 for k in c["kernels"])
