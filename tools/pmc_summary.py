"""Summarise the rocprofv3 passes of tools/pmc_collect.sh for one config into profiles/<round>_pmc_<config>.json
(what bench.py's roofline reads) and profiles/<round>_kernel_stats_<config>.csv.

  python3 tools/pmc_summary.py <outdir> <config> <round>

Per-launch values = mean over the dispatches of the TIMED kernel (its name is taken from the bench line in the pass
logs).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-byte requests at 64 bytes);
FETCH_SIZE / WRITE_SIZE are in KiB.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out, cfg, rnd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "r02")


def bench_line(log):
    try:
        for line in open(log):
            if line.startswith("{") and '"metric"' in line:
                return json.loads(line)
    except Exception:
        pass
    return None


line = None
for log in sorted(glob.glob(os.path.join(out, "*.log"))):
    line = bench_line(log) or line
if line is None:
    raise SystemExit(f"no bench line in {out}/*.log")
kernel = line["roofline"]["kernel"]                       # e.g. rtk_render_kernel<double, 256u, false, true>
short = kernel.replace("rtk_render_kernel", "rtk::rtk_render_kernel")


def is_timed(name):
    return short in name.replace("void ", "")


def newest_per_pass(pattern):
    """One file per pass directory: the most recent one (gpurun merges a new collection over an older one's files)."""
    best = {}
    for f in glob.glob(pattern, recursive=True):
        key = os.path.relpath(f, out).split(os.sep)[0]
        if key not in best or os.path.getmtime(f) > os.path.getmtime(best[key]):
            best[key] = f
    return list(best.values())


vals, meta, n_disp = {}, {}, {}
for f in newest_per_pass(os.path.join(out, "*", "**", "*counter_collection.csv")):
    with open(f) as fh:
        rows = [r for r in csv.DictReader(fh) if is_timed(r.get("Kernel_Name", ""))]
    per_counter = {}
    for r in rows:
        per_counter.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        per_counter[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
            if k in r:
                meta[k] = r[k]
    for name, by_dispatch in per_counter.items():
        vals[name] = sum(by_dispatch.values()) / len(by_dispatch)
        n_disp[name] = len(by_dispatch)

# kernel durations from the --kernel-trace --stats run of the same command
dur, stats_rows = [], []
for f in newest_per_pass(os.path.join(out, "stats", "**", "*kernel_trace.csv")):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if is_timed(r.get("Kernel_Name", "")):
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for f in newest_per_pass(os.path.join(out, "stats", "**", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats_{cfg}.csv"))
kernel_ms_profiled = sum(dur) / len(dur) if dur else None
# A frame with more than 21 sample chunks is SEVERAL launches of the timed kernel (passes over the chunks, csrc/rtk_api.cpp):
# bench.py's kernel_ms brackets the frame, so the additive counters and the duration are scaled from "mean dispatch" to "frame".
launches = int(line["roofline"].get("launches_per_frame") or 1)
if launches > 1:
    vals = {k: v * launches for k, v in vals.items()}
    if kernel_ms_profiled is not None:
        kernel_ms_profiled *= launches

import bench  # noqa: E402  (kernel_source_hash)

hash_file = os.path.join(out, "source_hash.txt")   # written at collection time (tools/pmc_collect.sh)
source_hash = open(hash_file).read().strip() if os.path.exists(hash_file) else bench.kernel_source_hash()

g = vals.get
rec = {
    "config": cfg, "kernel": kernel, "workload": line["config"]["workload"].split("spp")[0], "n_gpus": 1, "dtype": "f64",
    "order": line["config"]["order"], "source_hash": source_hash,
    "method": "rocprofv3 --pmc passes over `python3 bench.py --config %s ...` (tools/pmc_collect.sh), mean over the dispatches of the timed kernel; "
              "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); durations from a separate --kernel-trace --stats run" % cfg,
    "counters": vals, "dispatches_averaged": n_disp, "launches_per_frame": launches,
    "counters_are": "per frame = mean over the timed kernel's dispatches x launches_per_frame", "kernel_ms_profiled": kernel_ms_profiled, "kernel_ms_bench_events": line["roofline"]["kernel_ms"],
    "vgpr": meta.get("VGPR_Count"), "sgpr": meta.get("SGPR_Count"), "scratch": meta.get("Scratch_Size"), "lds_block": meta.get("LDS_Block_Size"),
    "workgroup": meta.get("Workgroup_Size"), "grid": meta.get("Grid_Size"),
}
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    rec["hbm_bytes_per_launch"] = int((2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024)
if g("SQ_WAVE_CYCLES"):
    rec["wait_share"] = {k: round(g(k) / g("SQ_WAVE_CYCLES"), 4) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                                                            "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA") if g(k) is not None}
if g("SQ_LDS_IDX_ACTIVE"):
    rec["lds_bank_conflict_share"] = round(g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE"), 4)
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
    rec["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{cfg}.json")
json.dump(rec, open(path, "w"), indent=1, sort_keys=True)
print(f"wrote {path}")
if all(k in vals for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU")) and kernel_ms_profiled:
    print(json.dumps(bench.valu_roofline(rec, kernel_ms_profiled), indent=1))
print(json.dumps({k: rec.get(k) for k in ("kernel", "kernel_ms_profiled", "kernel_ms_bench_events", "vgpr", "scratch", "hbm_bytes_per_launch", "wait_share",
                                          "lds_bank_conflict_share", "l2_hit_rate")}, indent=1))
