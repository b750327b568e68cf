"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --output-format csv) into
profiles/<out>.json: HBM bytes per launch of every kernel.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o fetch -- python3 tools/profile_ops.py --repeats 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o write -- python3 tools/profile_ops.py --repeats 1
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/fetch_counter_collection.csv gpurun_out/pmc_write/write_counter_collection.csv profiles/r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE count kilobytes; on gfx950 FETCH_SIZE counts the 128-byte requests of wide coalesced reads as
64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is used as is."""
import csv
import json
import re
import sys


def per_kernel(path, counter):
    agg = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").strip()
            a = agg.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return agg


fetch, write, out = sys.argv[1:4]
commit = sys.argv[4] if len(sys.argv) > 4 else None     # git is not available on the GPU box: pass `git rev-parse --short HEAD`
workload = sys.argv[5] if len(sys.argv) > 5 else ("`python tools/profile_ops.py --repeats 1` (2 eager U-Net evaluations @ latent "
                                                  "(1,8,48,128,128))")
fa, wa = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
res = {"commit": commit, "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes; 2*FETCH_SIZE + WRITE_SIZE",
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over " + workload + ". FETCH_SIZE is doubled per "
               "MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B for wide coalesced reads); WRITE_SIZE is used "
               "as is. hbm_bytes_per_launch = 2*FETCH + WRITE.", "kernels": {}}
for name in sorted(set(fa) | set(wa)):
    f, w = fa.get(name, [0, 0.0]), wa.get(name, [0, 0.0])
    fk = f[1] / f[0] if f[0] else 0.0
    wk = w[1] / w[0] if w[0] else 0.0
    res["kernels"][name] = {"FETCH_SIZE_KB_avg_per_launch": fk, "launches_FETCH_SIZE": f[0],
                            "WRITE_SIZE_KB_avg_per_launch": wk, "launches_WRITE_SIZE": w[0],
                            "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
    print(f"{k[:60]:60s} launches {v['launches_FETCH_SIZE']:4d}  {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")
