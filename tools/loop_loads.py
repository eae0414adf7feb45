#!/usr/bin/env python3
"""Find loops whose global loads are drained every iteration (tools/loop_loads.py file.hip [extra hipcc flags]): compiles the file to
gfx950 assembly and lists, per kernel, every backward branch whose body holds global/buffer loads AND an `s_waitcnt vmcnt(0)` — the
pattern that cost roma_chol_step 10 us per launch (one load per iteration, each a full memory round trip)."""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]
out = tempfile.mktemp(suffix=".s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out, src] + sys.argv[2:],
               check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
os.remove(out)
func, labels = None, {}
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        func, labels = m.group(1), {}
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
        continue
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.search(r"s_branch (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels:
        body = lines[labels[m.group(1)]:i]
        loads = sum(1 for b in body if re.search(r"\b(global_load|buffer_load|flat_load)", b))
        waits0 = sum(1 for b in body if re.search(r"s_waitcnt.*vmcnt\(0\)", b))
        if loads and waits0:
            print(f"{func[:70]:70s} loop {m.group(1)}: {len(body):5d} lines, {loads:3d} loads, {waits0} x vmcnt(0)")
