#!/usr/bin/env python3
"""Aggregate HBM traffic of the dominant kernel family from two rocprofv3 --pmc passes over scripts/conv_pass.py
(FETCH_SIZE and WRITE_SIZE must be collected separately: TCC has 4 slots, MI355X_MICROARCH.md).

  pmc_traffic.py <fetch_dir> <write_dir> <workload> <out.json> [kernels: fp32mfma|split]

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE (KiB) counts 128-B requests at 64 B, so
read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE (KiB) is exact for 16-B-per-lane streaming stores."""
import collections
import csv
import glob
import json
import sys


CONV_KERNELS = ("igemm_kernel", "wgrad_kernel", "igemm_fixup_kernel", "wgrad_reduce_kernel", "dgrad_empty_class_kernel",
                "igemm_split16_kernel", "wgrad_split_kernel", "dgrad_empty_class_split_kernel")
MAIN_KERNELS = ("igemm_kernel<", "wgrad_kernel<", "igemm_split16_kernel<", "wgrad_split_kernel<")


def load(d, counter):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    assert files, d
    per_kernel = collections.defaultdict(lambda: [0, 0.0])
    ops = 0
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        conv = any(k in name for k in CONV_KERNELS)
        fam = "conv" if conv else name.split("(")[0].split("::")[-1]
        per_kernel[fam][0] += 1
        per_kernel[fam][1] += float(r["Counter_Value"])
        ops += any(k in name for k in MAIN_KERNELS)      # one main kernel per conv op
    return per_kernel, ops


(fetch, n), (write, n2) = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
assert n == n2 and n > 0
rd = 2.0 * fetch["conv"][1] * 1024.0
wr = write["conv"][1] * 1024.0
out = {"workload": sys.argv[3], "kernels": sys.argv[5] if len(sys.argv) > 5 else "fp32mfma",
       "kernel_family": "conv fprop/dgrad/wgrad ops = the implicit-GEMM kernels (fp32-MFMA or split-operand) + their fix-up / slab-reduce kernels "
                        "(the layout passes that split operands are NOT conv ops: the training step's BatchNorm passes write sp directly)",
       "ops": n, "kernel_launches": fetch["conv"][0], "read_bytes_per_op": rd / n, "write_bytes_per_op": wr / n,
       "traffic_bytes_per_launch": (rd + wr) / n,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over scripts/conv_pass.py (every conv "
                 f"op of one {sys.argv[3]} step once); read = 2*FETCH_SIZE*1024 (gfx950 correction), write = WRITE_SIZE*1024; "
                 "summed over the op's kernels, divided by the number of ops",
       "other_kernels_total_MB": {k: round((2.0 * fetch[k][1] + write.get(k, [0, 0])[1]) * 1024 / 1e6, 1)
                                  for k in fetch if k != "conv"}}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out)[:700])
