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


def test_work_based_fraction_arithmetic():
    """roofline.work_frac = sum over units of work (exact counter x ISA-derived instruction cost, priced per class) / 64 lanes
    / (1024 SIMDs x 2.4 GHz x kernel time): a number that depends on the work and the time only -- not on how many
    instructions the kernel executed for it, nor on how full its waves were."""
    sys.path.insert(0, ROOT)
    import bench

    zero = {"f32": 0, "f64": 0, "trans_f32": 0, "trans_f64": 0, "mul_i32": 0, "lds": 0, "vmem": 0, "salu": 0}
    isa = {"costs": {k: dict(zero) for k in ("box", "sphere", "segment", "sample", "miss", "lambertian", "metal", "dielectric", "partial")}}
    isa["costs"]["box"].update(f32=16)
    isa["costs"]["sphere"].update(f32=10, f64=40, trans_f64=1)
    isa["costs"]["lambertian"].update(f64=50)
    isa["costs"]["metal"].update(f64=90)
    isa["costs"]["dielectric"].update(f64=120)
    counters = {"samples": 1000, "segments": 2600, "box_tests": 53000, "sphere_tests": 5400, "surface_hits": 1600}
    cycles = {"f32": 2.0, "f64": 4.0, "trans_f32": 8.0, "trans_f64": 16.0, "mul_i32": 8.0}
    r = bench.work_roofline(counters, isa, cycles, kernel_ms=1e-3)
    lane_cycles = 53000 * 32.0 + 5400 * (20 + 160 + 16) + 1600 * 200.0      # the cheapest material prices a hit
    want = lane_cycles / 64.0 / (1024 * 2.4e9 * 1e-6)
    assert abs(r["work_frac"] - want) < 1e-4 * want + 1e-4
    assert abs(sum(r["work_by_unit_share"].values()) - 1.0) < 1e-3
    assert abs(bench.work_roofline(counters, isa, cycles, kernel_ms=2e-3)["work_frac"] - want / 2) < 1e-4   # twice the time, half the fraction
    # measured issue rates replace the nominal prices class by class; missing ones fall back
    got, src = bench.issue_cycles({"issue_v_fma_f32": 0.5, "issue_v_fma_f64": 0.25, "issue_v_mul_f64": 0.25, "issue_v_add_f64": 0.25, "issue_v_rcp_f64": 0.0625,
                                   "issue_v_rcp_f32": 0.125, "issue_v_mul_lo_u32": 0.125})
    assert got == {"f32": 2.0, "f64": 4.0, "trans_f32": 8.0, "trans_f64": 16.0, "mul_i32": 8.0} and "measured" in src
    assert bench.issue_cycles(None)[0] == bench.NOMINAL_ISSUE_CYCLES
    # the VALU-issue utilisation prices transcendentals at their own class when given the table
    pmc = {"SQ_INSTS_VALU": 100.0, "SQ_INSTS_VALU_ADD_F64": 10.0, "SQ_INSTS_VALU_MUL_F64": 10.0, "SQ_INSTS_VALU_FMA_F64": 10.0, "SQ_INSTS_VALU_TRANS_F64": 5.0,
           "SQ_INSTS_VALU_TRANS_F32": 5.0, "SQ_THREAD_CYCLES_VALU": 64.0, "SQ_ACTIVE_INST_VALU": 1.0}
    u = bench.valu_roofline({"counters": pmc}, kernel_ms=1e-6, cycles=cycles)
    assert abs(u["achieved"] - (30 * 4 + 5 * 16 + 5 * 8 + 60 * 2) / 1e-9 / 1e9) < 1e-3 and u["frac_of_measured_issue_ceiling"] is None
    # ... and against the measured ceiling: a dense v_fma_f32 stream at the kernel's waves per SIMD (768 threads per CU = 3 waves:
    # between the 2- and the 4-wave measurement) at the clock the profiled launch held (GRBM_GUI_ACTIVE / 8 XCDs / its duration)
    pmc2 = dict(pmc, GRBM_GUI_ACTIVE=8 * 2.0e9 * 1e-9)
    m = bench.valu_roofline({"counters": pmc2, "workgroup": "768", "kernel_ms_profiled": 1e-6}, kernel_ms=1e-6, cycles=cycles, fma_by_waves={"1": 0.2, "2": 0.4, "4": 0.44, "8": 0.45})
    c = m["frac_of_measured_issue_ceiling"]
    assert c["waves_per_simd"] == 3.0 and abs(c["v_fma_f32_per_cycle_at_that_occupancy"] - 0.42) < 1e-9 and abs(c["clock_ghz"] - 2.0) < 1e-9
    assert abs(c["value"] - m["achieved"] / (1024 * 2.0 * 2.0 * 0.42)) < 1e-3


def test_isa_cost_table_is_derived_from_the_kernel_sources():
    """tools/isa_costs.py: the per-unit instruction costs behind work_frac come from the ISA of probe kernels that run the
    product's own step functions (compile-only: no GPU).  The table must be sane -- a box step is a couple of dozen f32-class
    instructions and no f64 arithmetic, a sphere test is dominated by f64 -- and the committed table, when it was derived
    from the present sources, must be exactly what the present sources give."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, ROOT)
    import bench
    import isa_costs

    table = isa_costs.derive()
    c = table["costs"]
    assert set(c) == {"box", "sphere", "segment", "sample", "miss", "lambertian", "metal", "dielectric", "partial"}
    assert 10 <= c["box"]["f32"] <= 30 and c["box"]["f64"] == 0 and c["box"]["trans_f64"] == 0
    assert c["sphere"]["f64"] >= 30 and c["sphere"]["trans_f64"] >= 1          # discriminant, square root, the corrected quotients
    assert c["lambertian"]["f64"] < c["metal"]["f64"] and c["lambertian"]["f64"] < c["dielectric"]["f64"]
    assert isa_costs.pipe_cycles(c["box"]) < isa_costs.pipe_cycles(c["sphere"]) < isa_costs.pipe_cycles(c["sample"]) + isa_costs.pipe_cycles(c["lambertian"])
    assert isa_costs.NOMINAL_CYCLES == bench.NOMINAL_ISSUE_CYCLES
    path = os.path.join(ROOT, "profiles", f"{bench.PMC_ROUND}_isa_costs.json")
    if os.path.exists(path):
        committed = json.load(open(path))
        if committed.get("source_hash") == bench.kernel_source_hash():
            assert committed["costs"] == c


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
    ceil = r["ceilings"]    # measured on this box in this run: a stream copy below the HBM spec, LDS far above it, f64 dearer than f32
    assert 1000 < ceil["hbm_copy_GBps"] < 8000 and ceil["lds_read_b128_GBps"] > 10000 and ceil["lds_read_b128_random_records_GBps"] > 1000
    ic = ceil["issue_cycles_per_wave_instruction"]
    assert ic["f32"] == 2.0 and ic["f64"] > ic["f32"] and ic["trans_f32"] > ic["f32"] and ic["trans_f64"] > ic["f64"] and "measured" in ceil["issue_cycles_source"]
    assert 1.0 < ceil["shader_clock_GHz_under_valu_load"] <= 2.45 and ceil["clock_light_load_GHz"] >= ceil["clock_dense_valu_GHz"] - 0.05
    by_waves = ceil["issue_v_fma_f32_by_waves_per_simd"]     # one wave alone cannot fill the issue port; eight nearly do (spec: 0.5 per cycle)
    assert by_waves["1"] < by_waves["2"] <= by_waves["8"] + 0.02 and 0.3 < by_waves["8"] <= 0.52
    assert "work_frac" in r and (r["work_frac"] is None or 0 < r["work_frac"] < 1)
    assert r["hbm_model"]["model_vs_lds_ceiling"] is not None
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["other_order"]["identical_framebuffer"] is True and d["other_order"]["fast_order_exact"] is True
    assert d["framebuffer_sha256"] == d["other_order"]["framebuffer_sha256"]


@pytest.mark.gpu
def test_bench_multi_abi_mode_reproduces_the_one_gpu_framebuffer():
    """`bench.py --multi abi`: ONE process driving the product's own multi-GPU path (rtk_render_multi_enqueue / rtk_multi_wait,
    two frames in flight) -- rehearsed on this one-GPU box with the device listed four times -- must print the same
    framebuffer digest as the N = 1 line of the same workload, and the contract's keys."""
    common = ["--steps", "3", "--warmup", "1", "--width", "320", "--height", "200", "--spp", "24"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common, "--no-cpu-baseline", "--no-f32", "--no-other-configs"],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([line for line in one.stdout.splitlines() if line.startswith("{")][-1])
    abi = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--multi", "abi", "--gpus", "4", "--multi-devices", "0,0,0,0", *common],
                         capture_output=True, text=True, timeout=600)
    assert abi.returncode == 0, abi.stderr[-2000:]
    lines = [line for line in abi.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    d4 = json.loads(lines[0])
    assert d4["multi"] == "abi" and d4["n_gpus"] == 4 and d4["devices"] == [0, 0, 0, 0] and d4["uses_rccl"] is False
    assert d4["framebuffer_sha256"] == d1["framebuffer_sha256"]
    assert d4["value"] > 0 and d4["blocking_render_multi"]["value"] > 0
    assert abs(d4["value"] - 320 * 200 * 24 / (d4["ms_per_step"] * 1e-3) / 1e6) / d4["value"] < 1e-3
