"""rocprofv3 --pmc FETCH_SIZE over tools/fetch_calib.hip -> profiles/<out>.json: the factor by which FETCH_SIZE has to be
multiplied to give the bytes actually read, per access shape (see the header of tools/fetch_calib.hip)."""
import csv
import json
import sys

path, out = sys.argv[1:3]
ROWS = 1224
expected = [("stream16", 2 << 30), ("dma_rows32_one (C=128: 32 of every 256 B)", 8192 * ROWS * 32),
            ("dma_rows32_all (C=128: 8 chunks, chunk outer)", 8192 * ROWS * 256), ("dma_rows32_c16 (dense 32-B rows)", 65536 * ROWS * 32)]
rows = [r for r in csv.DictReader(open(path)) if r.get("Counter_Name") == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
res = {"method": "rocprofv3 --pmc FETCH_SIZE over tools/fetch_calib (every kernel reads a known number of unique bytes once, "
                 "3 GiB buffer); factor = bytes / (FETCH_SIZE_KB * 1024)", "shapes": {}}
per = len(expected)
for i, r in enumerate(rows):
    name, nbytes = expected[i % per]
    kb = float(r["Counter_Value"])
    ent = res["shapes"].setdefault(name, {"bytes": nbytes, "FETCH_SIZE_KB": [], "factor": []})
    ent["FETCH_SIZE_KB"].append(kb)
    ent["factor"].append(nbytes / (kb * 1024.0) if kb else None)
json.dump(res, open(out, "w"), indent=1)
for k, v in res["shapes"].items():
    print(f"{k:50s} bytes {v['bytes'] / 1e6:9.1f} MB  FETCH_SIZE {[round(x / 1024, 1) for x in v['FETCH_SIZE_KB']]} MB  factor {[round(f, 3) for f in v['factor']]}")
