#!/usr/bin/env python3
"""Cuts the rocprofv3 kernel trace of tools/copy_probe.py per section and counts copyBuffer launches."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
names = open(sys.argv[2]).read().split("SECTIONS ")[1].strip().split("|")
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
sec, cur, out = -1, None, []
for r in rows:
    k = r["Kernel_Name"]
    if "cumsum" in k.lower() or "scan" in k.lower():
        sec += 1; cur = {}
        continue
    if "flip" in k.lower():
        if cur is not None: out.append((names[sec] if sec < len(names) else str(sec), cur)); cur = None
        continue
    if cur is not None:
        short = k.split("(")[0][-60:]
        cur[short] = cur.get(short, 0) + 1
for name, d in out:
    cp = sum(v for k, v in d.items() if "copyBuffer" in k)
    print(f"{name:45s} copyBuffer {cp:4d}   kernels: " + ", ".join(f"{k.split('::')[-1]}x{v}" for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:6]))
