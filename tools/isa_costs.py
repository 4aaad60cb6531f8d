"""Per-unit-of-work instruction costs of the lean MIXED render kernel, read from the ISA (no GPU needed).

Compiles csrc/rtk_trace.hip with -DRTK_ISA_PROBES (one probe kernel per unit of work: a box step, a sphere step, the start of
a segment, the start of a sample, a miss, a hit on each of the three materials, a partial-sum store -- each running the
product's own step function once on a lane state loaded from memory), counts every probe's VALU instructions by class and
subtracts the empty probe (load state, store state).  The result is what one wave executing that step ONCE issues:

  python3 tools/isa_costs.py [--write profiles/r03_isa_costs.json]

bench.py multiplies the exact work counters of a frame by these costs (priced with the issue rates measured by
csrc/rtk_microbench.hip) to get roofline.work_frac; tests/test_bench_contract.py re-derives the table and compares.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracingoneweekendapplication_amd", "csrc")
PROBES = ["EMPTY", "BOX", "SPHERE", "SEGMENT", "SAMPLE", "MISS", "LAMBERTIAN", "METAL", "DIELECTRIC", "PARTIAL"]  # enum ProbeKind order
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_exp_", "v_sin_", "v_cos_")
# nominal issue cost of a wave64 instruction in SIMD cycles per class, used when no measurement is at hand
# (MI355X_MICROARCH.md: v_fma_f32 2, transcendental f32 8 one wave alone; f64 at half the f32 rate)
NOMINAL_CYCLES = {"f32": 2.0, "f64": 4.0, "trans_f32": 8.0, "trans_f64": 16.0, "mul_i32": 8.0}


def classify(mnemonic):
    """VALU class of an instruction mnemonic, or None for scalar / memory / LDS / control instructions."""
    if not mnemonic.startswith("v_"):
        return None
    if mnemonic.startswith(TRANS):
        return "trans_f64" if "_f64" in mnemonic else "trans_f32"
    if mnemonic.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mul_lo_i32", "v_mad_u64_u32", "v_mad_i64_i32")):
        return "mul_i32"
    if "_f64" in mnemonic and not mnemonic.startswith("v_cvt_f32_f64") and not mnemonic.startswith("v_cmp"):
        return "f64"
    if mnemonic.startswith(("v_cvt_f64", "v_cmp_") ) and "_f64" in mnemonic:
        return "f64"
    return "f32"


def count_kernel(lines):
    counts = {"f32": 0, "f64": 0, "trans_f32": 0, "trans_f64": 0, "mul_i32": 0, "lds": 0, "vmem": 0, "salu": 0}
    for line in lines:
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        m = t.split()[0]
        c = classify(m)
        if c:
            counts[c] += 1
        elif m.startswith("ds_"):
            counts["lds"] += 1
        elif m.startswith(("global_", "buffer_", "flat_", "scratch_")):
            counts["vmem"] += 1
        elif m.startswith("s_") and not m.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_branch", "s_cbranch")):
            counts["salu"] += 1
    return counts


def derive(extra_flags=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "probes.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-S", "--cuda-device-only",
                               "-DRTK_ISA_PROBES", "-DRTK_DEV_ONLY_ALL", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *extra_flags,
                               os.path.join(CSRC, "rtk_trace.hip"), "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    kernels = {}
    name = None
    for line in text:
        m = re.match(r"^(_ZN3rtk13rtk_isa_probeILi(\d+)EE\w+):", line)
        if m:
            name = PROBES[int(m.group(2))]
            kernels[name] = []
            continue
        if name and line.startswith(".Lfunc_end"):
            name = None
        if name:
            kernels[name].append(line)
    missing = [p for p in PROBES if p not in kernels]
    if missing:
        raise SystemExit(f"probe kernels not found in the ISA: {missing}")
    raw = {k: count_kernel(v) for k, v in kernels.items()}
    base = raw["EMPTY"]
    costs = {}
    for k in PROBES[1:]:
        costs[k.lower()] = {c: max(0, raw[k][c] - base[c]) for c in base}
    return {"kernel_family": "rtk_render_kernel<double, 256u, *, *> (lean MIXED program: f32 centre / half-extent boxes, f64 spheres)",
            "unit": "wave64 instructions per step, one wave executing the step once (probe kernel minus the empty probe)",
            "costs": costs, "raw_empty": base}


def pipe_cycles(cost, cycles=None):
    """SIMD issue cycles of one wave executing the step once."""
    cycles = cycles or NOMINAL_CYCLES
    return sum(cost[c] * cycles[c] for c in ("f32", "f64", "trans_f32", "trans_f64", "mul_i32"))


def source_hash():
    sys.path.insert(0, ROOT)
    import bench
    return bench.kernel_source_hash()


if __name__ == "__main__":
    table = derive()
    table["source_hash"] = source_hash()
    for k, c in table["costs"].items():
        print(f"{k:12s} {c}  -> {pipe_cycles(c):7.1f} nominal SIMD cycles")
    if "--write" in sys.argv:
        path = sys.argv[sys.argv.index("--write") + 1]
        with open(path, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
        print("wrote", path)
