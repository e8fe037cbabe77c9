#!/usr/bin/env python3
"""Per-kernel summaries of rocprofv3 --pmc passes (counter_collection.csv).

  pmc_summarize.py traffic <fetch_dir> <write_dir> <steps> <out.json>     whole training steps: HBM bytes per kernel and per step
  pmc_summarize.py mfma <dir> <out.json>                                 GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
                                                                         SQ_ACTIVE_INST_VALU SQ_INSTS_VALU over scripts/conv_pass.py

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE (KiB) counts 128-byte requests at 64 bytes, so read bytes =
2 * FETCH_SIZE * 1024; WRITE_SIZE (KiB) is exact for 16-byte-per-lane streaming stores."""
import collections
import csv
import glob
import json
import re
import sys


def rows(d):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    assert files, d
    for f in files:
        yield from csv.DictReader(open(f))


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("mvg::", "").strip()


def traffic(fetch_dir, write_dir, steps, out):
    rd, wr, calls = collections.Counter(), collections.Counter(), collections.Counter()
    for r in rows(fetch_dir):
        if r["Counter_Name"] == "FETCH_SIZE":
            rd[short(r["Kernel_Name"])] += 2.0 * float(r["Counter_Value"]) * 1024.0
            calls[short(r["Kernel_Name"])] += 1
    for r in rows(write_dir):
        if r["Counter_Name"] == "WRITE_SIZE":
            wr[short(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024.0
    ks = sorted(set(rd) | set(wr), key=lambda k: -(rd[k] + wr[k]))
    per = {k: {"launches_per_step": calls[k] / steps, "read_GB_per_step": round(rd[k] / steps / 1e9, 3),
               "write_GB_per_step": round(wr[k] / steps / 1e9, 3)} for k in ks if (rd[k] + wr[k]) / steps > 5e7}
    tot_r, tot_w = sum(rd.values()) / steps, sum(wr.values()) / steps
    json.dump({"steps": steps, "read_GB_per_step": round(tot_r / 1e9, 2), "write_GB_per_step": round(tot_w / 1e9, 2),
               "total_GB_per_step": round((tot_r + tot_w) / 1e9, 2),
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over bench.py --steps N --warmup 0 "
                         "--no-overlap --no-roofline --no-cpu-baseline; read = 2*FETCH_SIZE*1024 (gfx950), write = WRITE_SIZE*1024; "
                         "kernels below 0.05 GB per step omitted from the table",
               "kernels": per}, open(out, "w"), indent=1)
    print(f"total {(tot_r + tot_w) / 1e9:.1f} GB per step (read {tot_r / 1e9:.1f}, written {tot_w / 1e9:.1f})")


def mfma(d, out):
    acc = collections.defaultdict(collections.Counter)
    launches = collections.Counter()
    for r in rows(d):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            launches[k] += 1
    res = {}

    def summarise(name, c, n):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back): cycles of the launch = / 8;
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (16 cycles per v_mfma_f32_16x16x32_f16, 32 per
        # 32x32x16): fraction of the SIMD-cycles in which the matrix pipe is busy = busy / (1024 * cycles)
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        if cyc <= 0 or c["SQ_INSTS_MFMA"] <= 0:
            return
        res[name] = {"launches": n, "mfma_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4),
                     "valu_insts_per_mfma": round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], 2),
                     "mfma_busy_cycles_per_mfma": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_MFMA"], 1),
                     "gpu_Mcycles": round(cyc / 1e6, 3)}
    # the family total = exactly the kernels that issued MFMAs (the rows below: their launches and cycles add up to it);
    # everything else that ran in the profiled command is listed under "not_counted" so that nothing disappears
    tot = collections.Counter()
    n = 0
    not_counted = {}
    for k, c in acc.items():
        if c["SQ_INSTS_MFMA"] > 0:
            for kk, v in c.items():
                tot[kk] += v
            n += launches[k]
        else:
            not_counted[k] = {"launches": launches[k], "gpu_Mcycles": round(c["GRBM_GUI_ACTIVE"] / 8.0 / 1e6, 3)}
    summarise("all kernels that issue MFMAs (cycle-weighted total of the rows below)", tot, n)
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"]):
        summarise(k, c, launches[k])
    rows_sum = sum(v["gpu_Mcycles"] for k, v in res.items() if not k.startswith("all kernels"))
    assert abs(rows_sum - res[next(iter(res))]["gpu_Mcycles"]) < 1e-2 * max(rows_sum, 1.0), "the table does not add up"
    json.dump({"kernels": res, "not_counted_no_mfma": not_counted, "note": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs): the share of "
               "SIMD-cycles with the matrix pipe busy, at whatever clock the chip held (the time-based roofline fraction in bench.py is "
               "this figure times held clock / 2.4 GHz)"}, open(out, "w"), indent=1)
    for k, v in list(res.items())[:8]:
        print(k, v["mfma_busy_frac"], v["valu_insts_per_mfma"], v["mfma_busy_cycles_per_mfma"])


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5])
    else:
        mfma(sys.argv[2], sys.argv[3])
