#!/usr/bin/env python3
"""Reduce the rocprofv3 outputs of scripts/prof_bench.sh to one JSON summary (kept under profiles/)."""
import csv, glob, json, os, re, sys


def kname(full):
    m = re.search(r"(ns2d_\w+|\w+_step_k)(<[^>]*>)?", full)
    return (m.group(1) + (m.group(2) or "")) if m else None

out = sys.argv[1]
res = {"kernels": {}, "pmc": {}}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = kname(r["Name"])
        if name:
            res["kernels"][name] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                    "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
for ctr, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            name = kname(r["Kernel_Name"])
            if not name:
                continue
            a = acc.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        for name, (n, v) in acc.items():
            res["pmc"].setdefault(name, {})[ctr] = {"dispatches": n, "mean_per_dispatch_raw": v / n}
# corrections (MI355X_MICROARCH.md, HBM): counters are in KiB; FETCH_SIZE reads half the bytes of a
# wide coalesced read stream on gfx950 (doubled here as the guide prescribes); WRITE_SIZE is exact.
for name, c in res["pmc"].items():
    fe = c.get("FETCH_SIZE", {}).get("mean_per_dispatch_raw")
    wr = c.get("WRITE_SIZE", {}).get("mean_per_dispatch_raw")
    if fe is not None and wr is not None:
        c["hbm_bytes_per_dispatch"] = (2.0 * fe + wr) * 1024.0
        c["note"] = "(2*FETCH_SIZE + WRITE_SIZE) * 1024"
print(json.dumps(res, indent=1))
