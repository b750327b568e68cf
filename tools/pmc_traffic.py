"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --output-format csv) into
profiles/<out>.json: HBM bytes per launch of every kernel.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o fetch -- python3 tools/profile_ops.py --repeats 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o write -- python3 tools/profile_ops.py --repeats 1
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/fetch_counter_collection.csv gpurun_out/pmc_write/write_counter_collection.csv profiles/r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE count kilobytes; on gfx950 FETCH_SIZE counts the 128-byte requests of wide coalesced reads as
64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is used as is: `hbm_bytes_per_launch`.
`hbm_bytes_calibrated` replaces the 2 by the factor measured for the kernel's access shape on this part
(profiles/r03_fetch_calibration.json, tools/fetch_calib.hip: kernels that read a known number of unique bytes once): 4.0 for the
halo DMA of conv3_halo_k32_kernel (lane pairs fetching 32-byte pieces of rows 256 B apart: one tallied 64-B request per row),
1.67 for 16-byte-per-lane streaming reads (gn_apply, attention passes, sampler update); kernels whose shape was not
calibrated keep 2.0 (`fetch_factor_source` says which)."""
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
CALIBRATED = [("conv3_halo_k32_kernel", 4.0, "dma_rows32_all"), ("gn_apply_kernel", 1.67, "stream16"), ("gn_colsum_kernel", 1.67, "stream16"),
              ("attn_broadcast_add", 1.67, "stream16"), ("attn_depthsum", 1.67, "stream16"), ("sampler_step", 1.67, "stream16"),
              ("add_bf16", 1.67, "stream16"), ("gn_bwd", 1.67, "stream16")]


def fetch_factor(name):
    for prefix, f, shape in CALIBRATED:
        if name.startswith(prefix):
            return f, "profiles/r03_fetch_calibration.json: " + shape
    return 2.0, "uncalibrated (MI355X_MICROARCH.md rule)"


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
                            "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
                            "fetch_factor_calibrated": fetch_factor(name)[0], "fetch_factor_source": fetch_factor(name)[1],
                            "hbm_bytes_calibrated": (fetch_factor(name)[0] * fk + wk) * 1024.0}
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
    print(f"{k[:60]:60s} launches {v['launches_FETCH_SIZE']:4d}  {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  "
          f"calibrated (x{v['fetch_factor_calibrated']:.2f}) {v['hbm_bytes_calibrated'] / 1e6:9.1f} MB")
