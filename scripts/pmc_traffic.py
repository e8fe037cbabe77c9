#!/usr/bin/env python3
"""Aggregate HBM traffic of the dominant kernel family from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE must be collected separately: TCC has 4 slots, MI355X_MICROARCH.md).

  pmc_traffic.py <fetch_dir> <write_dir> <workload> <out.json>

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE (KiB) counts 128-B requests at 64 B, so
read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE (KiB) is exact for 16-B-per-lane streaming stores."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    assert files, d
    per_kernel = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        fam = "conv" if ("igemm_kernel" in name or "wgrad_kernel" in name) else name.split("(")[0].split("::")[-1]
        per_kernel[fam][0] += 1
        per_kernel[fam][1] += float(r["Counter_Value"])
    return per_kernel


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
n = fetch["conv"][0]
assert n == write["conv"][0] and n > 0
rd = 2.0 * fetch["conv"][1] * 1024.0
wr = write["conv"][1] * 1024.0
out = {"workload": sys.argv[3], "kernel_family": "igemm_kernel + wgrad_kernel (conv fprop/dgrad/wgrad)",
       "launches": n, "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / n,
       "traffic_bytes_per_launch": (rd + wr) / n,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py; "
                 "read = 2*FETCH_SIZE*1024 (gfx950 correction), write = WRITE_SIZE*1024",
       "other_kernels_total_MB": {k: round((2.0 * fetch[k][1] + write.get(k, [0, 0])[1]) * 1024 / 1e6, 1)
                                  for k in fetch if k != "conv"}}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out)[:600])
