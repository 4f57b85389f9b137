#!/usr/bin/env python3
"""Per-launch timeline of ONE fieldsplit PCApply from a rocprofv3
--kernel-trace CSV: prints the launches of the last complete apply with their
durations and the idle gap before each (launch-bound vs execution-bound)."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import _guard                                      # noqa: E402
_guard.start_rss_watchdog()

root = sys.argv[1]
f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pcd::", "")
         for r in rows]
# an apply starts at its entry permutation (k_gather, or - one rank - the
# k_scatter by the inverse index set) and ends at the exit k_scatter
io = [i for i, n in enumerate(names) if n in ("k_scatter", "k_gather")]
e = io[-1]
s = io[-2]
t0 = int(rows[s]["Start_Timestamp"])
busy = 0
prev_end = t0
print("%-28s %8s %9s %9s %8s" % ("kernel", "grid", "start_us", "dur_us", "gap_us"))
for i in range(s, e + 1):
    r = rows[i]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += en - st
    grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    print("%-28s %8s %9.2f %9.2f %8.2f" % (names[i][:28], grid, (st - t0) / 1e3,
                                             (en - st) / 1e3, (st - prev_end) / 1e3))
    prev_end = en
total = int(rows[e]["End_Timestamp"]) - t0
print("launches %d  span %.1f us  busy %.1f us  idle %.1f us"
      % (e - s + 1, total / 1e3, busy / 1e3, (total - busy) / 1e3))
