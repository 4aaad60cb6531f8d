"""Summarise rocprofv3 --pmc csv output for the render kernel: per-dispatch counter values (last dispatch)."""
import csv, glob, os, sys, json
out = sys.argv[1]
vals = {}
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        rows = [r for r in csv.DictReader(fh) if "rtk_render_kernel" in r.get("Kernel_Name", "")]
    if not rows:
        continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            vals["_kernel"] = r["Kernel_Name"][:90]
            for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
                if k in r:
                    vals["_" + k] = r[k]
print(json.dumps(vals, indent=1, sort_keys=True))
g = vals.get
if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
    print("VALU lane utilisation = %.3f" % (g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0)))
if g("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS"):
        if g(k) is not None:
            print("%-22s / SQ_WAVE_CYCLES = %.3f" % (k, g(k) / g("SQ_WAVE_CYCLES")))
if g("SQ_LDS_IDX_ACTIVE"):
    print("LDS bank-conflict share = %.3f" % (g("SQ_LDS_BANK_CONFLICT", 0) / g("SQ_LDS_IDX_ACTIVE")))
if g("FETCH_SIZE") is not None:
    print("FETCH_SIZE KB = %.1f (x2 for wide streaming reads on gfx950), WRITE_SIZE KB = %s" % (g("FETCH_SIZE"), g("WRITE_SIZE")))
