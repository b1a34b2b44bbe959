#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of a round (gpurun_out/rNN_stats, rNN_pmc_fetch,
rNN_pmc_write) into the small files committed under profiles/:
  rNN_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (as emitted)
  rNN_pmc_by_kernel.csv    FETCH_SIZE / WRITE_SIZE per launch, aggregated by kernel
  gemm_traffic.json        HBM bytes per GEMM-family launch, read by bench.py ("traffic")
Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in
KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream, so reads are
doubled; WRITE_SIZE is exact for 16-B/lane streaming stores."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(os.path.dirname(here), "gpurun_out")


def one(pattern):
    m = glob.glob(os.path.join(src, pattern))
    return m[0] if m else None


st = one(f"{rnd}_stats/*/*_kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(here, f"{rnd}_kernel_stats.csv"))


BIG_GRID = 100000   # work-items: the (N*S)-row GEMMs; the position-table GEMMs launch < 25k


def agg(path, counter, big_only=False):
    d = collections.defaultdict(lambda: [0, 0.0])
    if not path:
        return d
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        if big_only and int(row["Grid_Size"]) < BIG_GRID:
            continue
        k = row["Kernel_Name"]
        d[k][0] += 1
        d[k][1] += float(row["Counter_Value"])
    return d


f = agg(one(f"{rnd}_pmc_fetch/*/*_counter_collection.csv"), "FETCH_SIZE")
w = agg(one(f"{rnd}_pmc_write/*/*_counter_collection.csv"), "WRITE_SIZE")
if f:
    with open(os.path.join(here, f"{rnd}_pmc_by_kernel.csv"), "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_KiB_per_launch,WRITE_SIZE_KiB_per_launch,"
                 "hbm_bytes_per_launch_corrected\n")
        for k in sorted(f, key=lambda k: -f[k][1]):
            n = f[k][0]
            fe = f[k][1] / n
            wr = w[k][1] / max(1, w[k][0]) if k in w else 0.0
            fh.write(f"\"{k}\",{n},{fe:.1f},{wr:.1f},{(2 * fe + wr) * 1024:.0f}\n")
    is_gemm = lambda k: "gemm_kernel" in k or "gemm_dma_kernel" in k
    fb = agg(one(f"{rnd}_pmc_fetch/*/*_counter_collection.csv"), "FETCH_SIZE", big_only=True)
    wb = agg(one(f"{rnd}_pmc_write/*/*_counter_collection.csv"), "WRITE_SIZE", big_only=True)
    gf = [(v[0], v[1]) for k, v in fb.items() if is_gemm(k)]
    gw = [(v[0], v[1]) for k, v in wb.items() if is_gemm(k)]
    n = sum(a for a, _ in gf)
    fetch = sum(b for _, b in gf) / n
    write = sum(b for _, b in gw) / max(1, sum(a for a, _ in gw))
    json.dump({"round": rnd, "kernel": "qarig::gemm_dma_kernel<*> + qarig::gemm_kernel<*>", "launches_profiled": n,
               "FETCH_SIZE_KiB_per_launch": round(fetch, 1),
               "WRITE_SIZE_KiB_per_launch": round(write, 1),
               "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
               "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE), bench.py --steps 1; "
                       "reads doubled per the gfx950 FETCH_SIZE calibration; launches of >= 100k "
                       "work-items only (the same set bench.py's roofline object averages)"},
              open(os.path.join(here, "gemm_traffic.json"), "w"), indent=1)
print("ok")
