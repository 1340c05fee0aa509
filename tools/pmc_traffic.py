#!/usr/bin/env python3
"""HBM-side traffic per launch and kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_profile.sh.
usage: python tools/pmc_traffic.py gpurun_out/pmc_<tag> profiles/rNN_traffic.json
FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 counts 128-byte fabric
read requests as 64 bytes)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summarize import short  # noqa: E402

root, out = sys.argv[1], sys.argv[2]
workload = [int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]] if len(sys.argv) > 6 else [64, 640, 640, "f16"]
acc = defaultdict(lambda: {"fetch": 0.0, "write": 0.0, "n_fetch": 0, "n_write": 0})
for g, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in sorted(glob.glob(os.path.join(root, g, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:   # newest run only
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != cname:
                continue
            k = short(row["Kernel_Name"])
            acc[k][g] += float(row["Counter_Value"]) * 1024.0 * (2.0 if g == "fetch" else 1.0)
            acc[k]["n_" + g] += 1
kern = {}
for k, v in acc.items():
    n = max(v["n_fetch"], v["n_write"], 1)
    kern[k] = {"launches": n, "fetch_bytes_per_launch": v["fetch"] / max(v["n_fetch"], 1),
               "write_bytes_per_launch": v["write"] / max(v["n_write"], 1),
               "traffic_bytes_per_launch": v["fetch"] / max(v["n_fetch"], 1) + v["write"] / max(v["n_write"], 1)}
json.dump({"workload": workload, "source": "tools/pmc_profile.sh + tools/pmc_traffic.py (bench.py --steps 2 --warmup 1, workload = [batch, H, W, dtype]); "
                     "separate --pmc passes for FETCH_SIZE and WRITE_SIZE; FETCH_SIZE doubled (gfx950 correction)",
           "kernels": kern}, open(out, "w"), indent=1)
print("wrote", out, len(kern), "kernels")
