"""rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (one pass, csv) ->
profiles/<out>.json: per kernel, the share of SIMD cycles with the MFMA pipe busy and the LDS bank-conflict share.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace \
        --output-format csv -d gpurun_out/pmc_mfma -o mfma -- python3 tools/profile_ops.py --repeats 1
    python tools/pmc_mfma.py gpurun_out/pmc_mfma/mfma_counter_collection.csv profiles/r02_pmc_mfma_busy.json <commit>

GRBM_GUI_ACTIVE is summed over the 8 XCDs: mfma_busy_frac = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs); the clock the part held
during a dispatch = GUI_ACTIVE / 8 / (End_Timestamp - Start_Timestamp), computed over dispatches of >= 100 us ONLY (the
counter window's start / drain ramps inflate it on short ones: 3.4-3.7 "GHz" on 20 us kernels in round 2; null when a kernel
has no long dispatch); profiled passes run a few percent below un-profiled ones)."""
import csv
import json
import re
import sys

path, out = sys.argv[1:3]
commit = sys.argv[3] if len(sys.argv) > 3 else None
agg = {}
with open(path) as f:
    for row in csv.DictReader(f):
        name = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").strip()
        a = agg.setdefault(name, {})
        c = a.setdefault(row["Counter_Name"], [0, 0.0])
        c[0] += 1
        c[1] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("End_Timestamp"):
            dt = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            t = a.setdefault("_ns", [0, 0.0])
            t[0] += 1
            t[1] += dt
            if dt >= 100e3:    # the clock estimate is only valid for long dispatches (>= 100 us): start / drain ramps of the
                lg = a.setdefault("_long", [0, 0.0, 0.0])   # counter window inflate it on short ones (3.4-3.7 "GHz" on 20 us kernels)
                lg[0] += 1
                lg[1] += dt
                lg[2] += float(row["Counter_Value"])
res = {"commit": commit, "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
                                   "over tools/profile_ops.py --repeats 1 (eager U-Net evaluations @ latent (1,8,48,128,128))",
       "kernels": {}}
for name, a in agg.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in a or "GRBM_GUI_ACTIVE" not in a:
        continue
    busy, gui = a["SQ_VALU_MFMA_BUSY_CYCLES"][1], a["GRBM_GUI_ACTIVE"][1]
    if busy <= 0 or gui <= 0:
        continue
    conf = a.get("SQ_LDS_BANK_CONFLICT", [0, 0.0])[1]
    act = a.get("SQ_LDS_IDX_ACTIVE", [0, 1.0])[1]
    ns = a.get("_ns", [0, 0.0])[1]
    res["kernels"][name] = {"launches": a["GRBM_GUI_ACTIVE"][0], "mfma_busy_frac": busy / (gui / 8.0 * 1024.0),
                            "lds_bank_conflict_frac": conf / act if act > 0 else None,
                            "avg_dispatch_us": ns / a["GRBM_GUI_ACTIVE"][0] / 1e3 if ns > 0 else None,
                            "clock_ghz": (a["_long"][2] / 8.0 / a["_long"][1]) if a.get("_long") else None,
                            "clock_ghz_dispatches": a["_long"][0] if a.get("_long") else 0}
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["mfma_busy_frac"]):
    print(f"{k[:70]:70s} launches {v['launches']:4d}  MFMA busy {100 * v['mfma_busy_frac']:5.1f} %  "
          f"clock {v['clock_ghz'] or 0:.2f} GHz  {v['avg_dispatch_us'] or 0:7.1f} us")
