#!/usr/bin/env python3
"""HBM bytes per local_corr launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

usage: tools/pmc_traffic.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <algorithmic bytes per launch>
Corrections per /opt/skills/guides/MI355X_MICROARCH.md: the counters are in KB (1024 B); on gfx950 FETCH_SIZE counts
128-byte requests as 64 B, so it is doubled.  Counters are summed over XCDs / instances per dispatch by rocprofv3."""
import csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(d, name):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    vals = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != name or "local_corr" not in r.get("Kernel_Name", ""):
                continue
            key = (f, r.get("Dispatch_Id"))
            vals[key] = vals.get(key, 0.0) + float(r["Counter_Value"])
    if not vals:
        raise SystemExit(f"no {name} rows for local_corr under {d}")
    return sum(vals.values()) / len(vals), len(vals)


def source_sha1():
    """Same digest as bench.py's kernel_source_sha1(): all source files of roma_local_corr."""
    h = hashlib.sha1()
    sys.path.insert(0, ROOT)
    from bench import LOCAL_CORR_SOURCES                      # one list, one digest
    for f in LOCAL_CORR_SOURCES:
        h.update(open(os.path.join(ROOT, "roma_amd", "csrc", f), "rb").read())
    return h.hexdigest()


fetch, n1 = mean_counter(sys.argv[1], "FETCH_SIZE")
write, n2 = mean_counter(sys.argv[2], "WRITE_SIZE")
alg = float(sys.argv[3])
out = {
    # bench.py reports this figure only while local_corr.hip is the file it was collected for
    "kernel_source_sha1": source_sha1(),
    "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
    "fetch_size_kb_mean": fetch, "write_size_kb_mean": write, "dispatches": min(n1, n2),
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --no-cpu --no-microbench --steps 3 "
              "--warmup 1`; mean over the in-pipeline local_corr dispatches (5 shapes per step, fp16, B=2, the flows the random-init "
              "pipeline produces); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); KB = 1024 B",
    "algorithmic_bytes_per_launch": alg,
}
print(json.dumps(out, indent=1))
