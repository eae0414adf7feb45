#!/usr/bin/env python3
"""Cut the last timed step out of a rocprofv3 kernel trace of bench.py and print per-kernel totals (markdown)."""
import collections, csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else glob.glob("gpurun_out/prof_bench/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
fin = [i for i, r in enumerate(rows) if "finalize_kernel" in r["Kernel_Name"]]
step = rows[fin[-2] + 1: fin[-1] + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[r["Kernel_Name"][:100]][0] += 1
    agg[r["Kernel_Name"][:100]][1] += d
    busy += d
print(f"last step: {(t1-t0)/1e6:.2f} ms on the GPU timeline, {len(step)} kernel launches, sum of kernel durations {busy/1e3:.2f} ms\n")
print("| ms | launches | avg us | kernel |\n|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"| {v[1]/1e3:.3f} | {v[0]} | {v[1]/v[0]:.1f} | `{k}` |")
