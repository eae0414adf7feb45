#!/usr/bin/env python3
"""Mean per-dispatch value of every counter collected by tools/pmc_collect.sh, for the local_corr kernels, plus the kernels'
mean duration from the kernel trace of the same pass.  usage: tools/pmc_table.py gpurun_out/r2pmc [kernel-name substring, default local_corr]"""
import csv, glob, os, sys

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "local_corr"
for d in sorted(p for p in glob.glob(root + "/*") if os.path.isdir(p)):
    ctr, dur = {}, {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if pat not in k:
                continue
            key = (k.split("(")[0], r["Counter_Name"])
            e = ctr.setdefault(key, {})
            e[r["Dispatch_Id"]] = e.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if pat in k:
                dur.setdefault(k.split("(")[0], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"== {os.path.basename(d)}")
    for k, v in dur.items():
        print(f"   {k}: {len(v)} dispatches, mean {sum(v) / len(v) / 1e3:.1f} us (under the counter pass)")
    for (k, c), e in sorted(ctr.items()):
        print(f"   {k:32s} {c:28s} {sum(e.values()) / len(e):16.1f}  (n={len(e)})")
