"""bench.py's contract with the driver: one JSON line with the agreed keys; no GPU -> a loud refusal, never a CPU path."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_to_run_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "MI355X" in (p.stderr + p.stdout)
    assert not [line for line in p.stdout.splitlines() if line.startswith("{")]  # no number is ever printed from a CPU


def test_bench_starts_its_own_ranks_when_given_gpus_without_a_launcher():
    """`python bench.py --gpus N` as the driver may invoke it: the parent must spawn N ranks (torch.distributed.run child)
    before touching any GPU API and forward their output and exit code.  RTK_BENCH_LAUNCH_ONLY makes a rank report its
    environment and stop before the GPU check, so the launcher is testable here."""
    env = dict(os.environ, RTK_BENCH_LAUNCH_ONLY="1")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(key, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    ranks = [json.loads(line) for line in p.stdout.splitlines() if line.startswith("{")]
    assert sorted(r["rank"] for r in ranks) == [0, 1, 2]
    assert all(r["world"] == 3 and r["gpus"] == 3 and r["master"] == "127.0.0.1" for r in ranks)
    assert sorted(r["local_rank"] for r in ranks) == [0, 1, 2]
    assert "torch.distributed.run" in p.stderr
    # a failing rank must fail the command (no GPU here: the ranks refuse to run)
    if not torch.cuda.is_available():
        env.pop("RTK_BENCH_LAUNCH_ONLY")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode != 0
        assert not [line for line in p.stdout.splitlines() if line.startswith("{")]


def test_roofline_arithmetic_and_stale_profile_refusal(tmp_path, monkeypatch):
    """The VALU-issue roofline: pipe-cycles = 2 x f32-class + 4 x f64 wave-instructions, over 1024 SIMDs x 2.4 GHz x kernel
    time; a PMC file taken on other kernel sources, another kernel or another workload is refused."""
    sys.path.insert(0, ROOT)
    import bench

    counters = {"SQ_INSTS_VALU": 10e9, "SQ_INSTS_VALU_ADD_F64": 1e9, "SQ_INSTS_VALU_MUL_F64": 1e9, "SQ_INSTS_VALU_FMA_F64": 0.5e9, "SQ_INSTS_VALU_TRANS_F64": 0.0,
                "SQ_THREAD_CYCLES_VALU": 32.0 * 11e9, "SQ_ACTIVE_INST_VALU": 11e9}
    r = bench.valu_roofline({"counters": counters}, kernel_ms=20.0)
    pipe = 4 * 2.5e9 + 2 * 7.5e9
    assert abs(r["achieved"] - pipe / 0.020 / 1e9) < 0.1 and r["peak"] == 1024 * 2.4
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] <= 1
    assert abs(r["valu_lane_utilisation"] - 0.5) < 1e-6 and abs(r["useful_lane_frac"] - r["frac"] * 0.5) < 1e-3
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    for rel in bench.KERNEL_SOURCES:
        os.makedirs(os.path.dirname(tmp_path / rel), exist_ok=True)
        (tmp_path / rel).write_text("// " + rel)
    good = {"kernel": "k", "workload": "w 8x8x1", "n_gpus": 1, "source_hash": bench.kernel_source_hash(), "counters": counters}
    path = tmp_path / "profiles" / f"{bench.PMC_ROUND}_pmc_c2.json"
    path.write_text(json.dumps(good))
    rec, src = bench.load_pmc("c2", "k", "w 8x8x1", 1)
    assert rec is not None and src.endswith("_pmc_c2.json")
    for key, val in (("kernel", "other"), ("workload", "w 9x9x1"), ("n_gpus", 2)):
        args = {"kernel": "k", "workload": "w 8x8x1", "n_gpus": 1}
        args[key] = val
        rec, why = bench.load_pmc("c2", args["kernel"], args["workload"], args["n_gpus"])
        assert rec is None and "stale" in why and key in why
    (tmp_path / bench.KERNEL_SOURCES[0]).write_text("// edited kernel")
    rec, why = bench.load_pmc("c2", "k", "w 8x8x1", 1)
    assert rec is None and "source_hash" in why
    assert bench.load_pmc("c3", "k", "w 8x8x1", 1)[0] is None


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--width", "320", "--height", "200", "--spp", "24",
           "--cpu-seconds", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [line for line in p.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "Msamples/s" and d["dtype"] == "f64" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 320 * 200 * 24 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3  # value IS samples / time
    assert d["config"]["reduced"] is True and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert key in r, key
    assert r["bound"] == "valu" and r["traffic"] is None and r["frac"] is None  # PMC profiles are quoted for the full-size workloads only
    assert "refused" in r["pmc"] and r["hbm_model"]["algorithmic_bytes_per_sample"] > 0 and r["hbm_model"]["box_record_bytes"] == 32
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["other_order"]["identical_framebuffer"] is True and d["other_order"]["fast_order_exact"] is True
    assert d["framebuffer_sha256"] == d["other_order"]["framebuffer_sha256"]
