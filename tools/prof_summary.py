#!/usr/bin/env python3
"""Summarises a tools/prof_r02.sh output directory: per-kernel average durations of every trace, per-kernel PMC
averages, and HBM bytes per frame (FETCH_SIZE is reported in KB and counts 64 B per 128-B request of a wide coalesced
read on gfx950: x1024 x2, MI355X_MICROARCH.md "HBM"; WRITE_SIZE x1024).  Writes <dir>/summary.json."""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
out = {"traces": {}, "pmc": {}}


def short(name):
    name = name.replace("vbx::", "")
    i = name.find("(")
    return (name[:i] if i > 0 else name)[:70]


for t in sorted(glob.glob(os.path.join(d, "trace_*"))):
    if not os.path.isdir(t):
        continue
    for f in glob.glob(os.path.join(t, "*", "*kernel_stats.csv")):
        rows = []
        for r in csv.DictReader(open(f)):
            rows.append({"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                         "pct": float(r["Percentage"])})
        out["traces"][os.path.basename(t)] = rows
        print("==", os.path.basename(t))
        for r in rows[:12]:
            print("   %-70s calls %4d  avg %9.3f ms  %5.1f %%" % (r["kernel"], r["calls"], r["avg_ms"], r["pct"]))

for p in sorted(glob.glob(os.path.join(d, "pmc_*"))):
    if not os.path.isdir(p):
        continue
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(p, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    name = os.path.basename(p)
    out["pmc"][name] = {}
    print("==", name)
    for (k, c), v in sorted(acc.items()):
        if k.startswith("void at::") or "elementwise" in k or "fillBuffer" in k:
            continue
        out["pmc"][name].setdefault(k, {})[c] = {"avg": sum(v) / len(v), "n": len(v)}
        print("   %-60s %-24s n %3d avg %.6g" % (k, c, len(v), sum(v) / len(v)))

json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
