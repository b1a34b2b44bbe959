#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of a build (tools/profile_round.sh TAG: gpurun_out/TAG_stats,
TAG_pmc_fetch, TAG_pmc_write, TAG_pmc_sq1, TAG_pmc_sq2) into the small files committed under
profiles/:
  TAG_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (as emitted)
  TAG_pmc_by_kernel.csv    FETCH_SIZE / WRITE_SIZE per launch, aggregated by kernel
  TAG_sq_by_kernel.csv     SQ / GRBM counters per launch, aggregated by kernel, plus derived
                           MFMA-busy share, VALU instructions per MFMA and effective clock
  gemm_traffic.json        fabric-side bytes per GEMM-family launch and configuration, read by bench.py
                           ("traffic"): FETCH_SIZE / WRITE_SIZE count L2 misses, i.e. traffic that reaches
                           the Infinity Fabric -- Infinity-Cache (MALL) hits included, so an upper bound of
                           the HBM bytes

    python3 profiles/summarize.py TAG --on-box [--config c4]   (on the GPU box: raw CSVs -> gpurun_out/TAG_summary,
                                                    raw per-dispatch counter CSVs deleted: they
                                                    exceed what gpurun merges back)
    python3 profiles/summarize.py TAG              (here: gpurun_out/TAG_summary -> profiles/)

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in
KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream, so reads are
doubled; WRITE_SIZE is exact for 16-B/lane streaming stores.  SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE
are summed over the 8 XCDs by rocprofv3; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (same guide, cycle-constants table)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
on_box = "--on-box" in sys.argv
cfg_name = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "c2"
if "--precision" in sys.argv:
    cfg_name += "_" + sys.argv[sys.argv.index("--precision") + 1]
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(os.path.dirname(here), "gpurun_out")
summ = os.path.join(src, f"{tag}_summary")

BIG_GRID = 100000   # work-items: the (N*S)-row GEMMs; the position-table GEMMs launch < 25k


def one(pattern):
    m = glob.glob(os.path.join(src, pattern), recursive=True)
    return m[0] if m else None


def short(k):
    """Kernel name without its argument list."""
    k = k.replace("void ", "")
    depth, out = 0, []
    for ch in k:          # drop the trailing "(...)" parameter list, keep template arguments
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).strip()


def agg(path, big_only=False):
    """{kernel: {counter: [launches, sum]}} and {kernel: [launches, sum duration ns]}"""
    d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    if not path:
        return d
    for row in csv.DictReader(open(path)):
        if big_only and int(row["Grid_Size"]) < BIG_GRID:
            continue
        c = d[short(row["Kernel_Name"])][row["Counter_Name"]]
        c[0] += 1
        c[1] += float(row["Counter_Value"])
    return d


def durations(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    if not path:
        return d
    for row in csv.DictReader(open(path)):
        k = d[short(row["Kernel_Name"])]
        k[0] += 1
        k[1] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return d


def box():
    os.makedirs(summ, exist_ok=True)
    st = one(f"{tag}_stats/**/*_kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(summ, f"{tag}_kernel_stats.csv"))
    fpath = one(f"{tag}_pmc_fetch/**/*_counter_collection.csv")
    wpath = one(f"{tag}_pmc_write/**/*_counter_collection.csv")
    f, w = agg(fpath), agg(wpath)
    if f:
        with open(os.path.join(summ, f"{tag}_pmc_by_kernel.csv"), "w") as fh:
            fh.write("kernel,launches,FETCH_SIZE_KiB_per_launch,WRITE_SIZE_KiB_per_launch,"
                     "hbm_bytes_per_launch_corrected\n")
            for k in sorted(f, key=lambda k: -f[k]["FETCH_SIZE"][1]):
                n, tot = f[k]["FETCH_SIZE"]
                fe = tot / max(1, n)
                wn, wt = w[k]["WRITE_SIZE"] if k in w else (0, 0.0)
                wr = wt / max(1, wn)
                fh.write(f"\"{k}\",{n},{fe:.1f},{wr:.1f},{(2 * fe + wr) * 1024:.0f}\n")
        is_gemm = lambda k: "gemm_" in k and "reduce" not in k and "skinny" not in k
        fb, wb = agg(fpath, big_only=True), agg(wpath, big_only=True)
        gf = [v["FETCH_SIZE"] for k, v in fb.items() if is_gemm(k)]
        gw = [v["WRITE_SIZE"] for k, v in wb.items() if is_gemm(k)]
        n = sum(a for a, _ in gf)
        if n:
            fetch = sum(b for _, b in gf) / n
            write = sum(b for _, b in gw) / max(1, sum(a for a, _ in gw))
            json.dump({"config": cfg_name, "round": tag,
                       "kernel": "qarig GEMM family (gemm_dma_pf* / gemm_kernel / gemm_lp* / gemm_f8*)",
                       "launches_profiled": n,
                       "FETCH_SIZE_KiB_per_launch": round(fetch, 1),
                       "WRITE_SIZE_KiB_per_launch": round(write, 1),
                       "fabric_bytes_per_launch": int((2 * fetch + write) * 1024),
                       "what": "L2-miss (fabric-side) bytes per launch, Infinity-Cache hits included: an upper "
                               "bound of the HBM bytes",
                       "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py; reads doubled "
                               "per the gfx950 FETCH_SIZE calibration; launches of >= 100k work-items only "
                               "(the same set bench.py's roofline object averages)"},
                      open(os.path.join(summ, f"{tag}_gemm_traffic.json"), "w"), indent=1)
    # SQ / GRBM sets
    rows = collections.defaultdict(dict)
    for p in ("sq1", "sq2"):
        path = one(f"{tag}_pmc_{p}/**/*_counter_collection.csv")
        a = agg(path)
        dur = durations(one(f"{tag}_pmc_{p}/**/*_kernel_trace.csv"))
        for k, cs in a.items():
            for c, (n, tot) in cs.items():
                rows[k][c] = tot / max(1, n)
                rows[k]["launches"] = n
            if k in dur:
                rows[k][f"avg_us_{p}"] = dur[k][1] / dur[k][0] / 1e3
    if rows:
        cols = ["launches", "avg_us_sq1", "SQ_WAVES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES",
                "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_VALU_MFMA_BUSY_CYCLES",
                "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
                "SQ_VALU_MFMA_COEXEC_CYCLES"]
        with open(os.path.join(summ, f"{tag}_sq_by_kernel.csv"), "w") as fh:
            fh.write("kernel," + ",".join(cols) +
                     ",mfma_busy_frac_of_active_cu_cycles,valu_insts_per_mfma,effective_clock_GHz\n")
            order = sorted(rows, key=lambda k: -rows[k].get("SQ_BUSY_CYCLES", 0) * rows[k].get("launches", 0))
            for k in order:
                r = rows[k]
                vals = [f"{r.get(c, float('nan')):.1f}" for c in cols]
                # SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's SIMDs' matrix pipes per SE ...
                # the guide's definition: cycles (not quad-cycles) the MFMA pipe is busy; normalise by
                # the CU-cycles the kernel was resident (SQ_BUSY_CU_CYCLES is not collected: use
                # GRBM_GUI_ACTIVE/8 x 256 CUs x 4 SIMDs as the upper bound of pipe-cycles)
                gui = r.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
                mb = r.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan"))
                frac = mb / (gui * 256 * 4) if gui else float("nan")
                vpm = r.get("SQ_INSTS_VALU", float("nan")) / r["SQ_INSTS_MFMA"] if r.get("SQ_INSTS_MFMA") else float("nan")
                us = r.get("avg_us_sq1", 0.0)
                clk = gui / (us * 1e3) if us else float("nan")
                fh.write(f"\"{k}\"," + ",".join(vals) + f",{frac:.4f},{vpm:.2f},{clk:.3f}\n")
    # drop the raw per-dispatch counter CSVs (tens of MB each)
    for p in glob.glob(os.path.join(src, f"{tag}_pmc_*")) + glob.glob(os.path.join(src, f"{tag}_stats")):
        if os.path.isdir(p):
            shutil.rmtree(p, ignore_errors=True)
    print("summary in", summ)


def local():
    n = 0
    for p in glob.glob(os.path.join(summ, "*")):
        if p.endswith("_gemm_traffic.json"):      # merged into profiles/gemm_traffic.json under its configuration
            one_cfg = json.load(open(p))
            dst = os.path.join(here, "gemm_traffic.json")
            allcfg = json.load(open(dst)) if os.path.exists(dst) else {}
            if "round" in allcfg:                 # the round-2 layout (one configuration, no key)
                allcfg = {}
            allcfg[one_cfg["config"]] = one_cfg
            json.dump(allcfg, open(dst, "w"), indent=1)
        else:
            shutil.copy(p, os.path.join(here, os.path.basename(p)))
        n += 1
    print(f"copied {n} files from {summ}")


if __name__ == "__main__":
    box() if on_box else local()
