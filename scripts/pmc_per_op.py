#!/usr/bin/env python3
"""Per conv op: HBM-side bytes from the PMC counters against the op's algorithmic bytes - where the family's over-fetch sits.

  CONV_PASS_LOG=ops.json rocprofv3 --pmc FETCH_SIZE ... -- python3 scripts/conv_pass.py 50 128 4 split      (and a WRITE_SIZE pass)
  pmc_per_op.py <fetch_dir> <write_dir> ops.json <out.txt>

Joins by launch order: the main kernels of the pass (igemm_split16_kernel / wgrad_split_kernel) appear in the order conv_pass.py
logged its ops; an op's slab-reduce / finalize launches that follow it are added to it.  read = 2 * FETCH_SIZE * 1024 (gfx950)."""
import csv
import glob
import json
import sys


def dispatches(d, counter):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    assert files, d
    rows = [r for r in csv.DictReader(open(files[0])) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0) for r in rows]


MAIN = ("igemm_split16_kernel<", "wgrad_split_kernel<")
TAIL = ("wgrad_reduce", "bn_bwd_finalize", "bn_bwd_dgamma", "dgrad_empty_class")


def per_op(disp, scale):
    ops, cur = [], None
    for name, v in disp:
        if any(m in name for m in MAIN):
            cur = [name.split("(")[0].replace("void mvg::", ""), v * scale]
            ops.append(cur)
        elif cur is not None and any(t in name for t in TAIL):
            cur[1] += v * scale
    return ops


fetch, write = per_op(dispatches(sys.argv[1], "FETCH_SIZE"), 2.0), per_op(dispatches(sys.argv[2], "WRITE_SIZE"), 1.0)
log = [o for o in json.load(open(sys.argv[3]))]
# the stem's two ops come first in the pass and are not in the log (row-window form): drop the main kernels in front
skip = len(fetch) - len(log)
assert skip >= 0 and len(fetch) == len(write), (len(fetch), len(write), len(log))
fetch, write = fetch[skip:], write[skip:]
agg = {}
lines = [f"{'op':9s} {'shape':24s} {'kernel':34s} | read MB: alg  pmc  ratio | write MB: alg  pmc ratio"]
for (op, shp, ar, aw), (kn, fr), (_, wr) in zip(log, fetch, write):
    lines.append(f"{op:9s} {shp:24s} {kn:34s} | {ar / 1e6:8.1f} {fr / 1e6:8.1f} {fr / ar:5.2f} | {aw / 1e6:8.1f} {wr / 1e6:8.1f} {wr / max(aw, 1):5.2f}")
    a = agg.setdefault(op, [0.0, 0.0, 0.0, 0.0])
    a[0] += ar; a[1] += fr; a[2] += aw; a[3] += wr
lines.append("")
for op, (ar, fr, aw, wr) in agg.items():
    lines.append(f"{op:9s} total: read {ar / 1e9:7.2f} GB algorithmic, {fr / 1e9:7.2f} GB counted ({fr / ar:4.2f}x); "
                 f"write {aw / 1e9:7.2f} / {wr / 1e9:7.2f} GB ({wr / aw:4.2f}x)")
tot = [sum(v[i] for v in agg.values()) for i in range(4)]
lines.append(f"all ops  : read {tot[0] / 1e9:.2f} -> {tot[1] / 1e9:.2f} GB, write {tot[2] / 1e9:.2f} -> {tot[3] / 1e9:.2f} GB; "
             f"(read + write) counted / algorithmic = {(tot[1] + tot[3]) / (tot[0] + tot[2]):.3f}")
open(sys.argv[4], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-6:]))
