#!/usr/bin/env python3
"""Measured errors behind the parity tolerances: run the GPU tests with BEACON_ERRLOG=<file> (tests/test_gpu_parity.py's
maxdiff() appends every evaluated difference there), then

    python scripts/tolerance_report.py <file> [filter]

prints, per (test, line of the assertion, call within the line), the number of evaluations and the largest error -- the figure a tolerance
is set from (<= 10 x measured)."""
import collections
import json
import sys

rows = collections.OrderedDict()
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    k = (d["test"], d["line"], d.get("pos", 0))
    n, m = rows.get(k, (0, 0.0))
    rows[k] = (n + 1, max(m, d["err"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
src = open(__file__.replace("scripts/tolerance_report.py", "tests/test_gpu_parity.py")).read().splitlines()
for (t, line, pos), (n, m) in rows.items():
    if flt in t:
        print("%-90s line %4d  n=%4d  max err %.3e   | %s" % (t, line, n, m, src[line - 1].strip()[:110]))
