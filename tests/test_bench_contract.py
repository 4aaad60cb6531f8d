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
    assert r["bound"] in ("hbm", "mfma") and r["traffic"] is None  # PMC traffic is quoted for the full-size workload only
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["other_order"]["identical_framebuffer"] is True and d["other_order"]["fast_order_exact"] is True
    assert d["framebuffer_sha256"] == d["other_order"]["framebuffer_sha256"]
