#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of tools/time_gmres.py: the launches of
one late GMRES iteration that are NOT part of the preconditioner apply
(system SpMV, Gram-Schmidt, normalisation) with durations and gaps."""
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
norm = [i for i, n in enumerate(names) if n == "k_normalize"]
i1 = norm[len(norm) // 2]                 # a mid-run iteration
i0 = max(i for i in norm if i < i1)
t0 = int(rows[i0]["End_Timestamp"])
prev = t0
inside_pc = busy_pc = busy_rest = 0
print("%-30s %9s %9s %8s" % ("kernel", "start_us", "dur_us", "gap_us"))
for i in range(i0 + 1, i1 + 1):
    st, en = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    n = names[i]
    tail = n in ("k_mdot", "k_mdot_reduce", "k_maxpy_norm", "k_normalize") or \
        i > max(j for j in range(i0, i1) if names[j].startswith("k_axpby")
                or names[j].startswith("k_cheb") or names[j].startswith("k_spmv")) - 2
    if n in ("k_mdot", "k_mdot_reduce", "k_maxpy_norm", "k_normalize") or i >= i1 - 5:
        print("%-30s %9.2f %9.2f %8.2f" % (n[:30], (st - t0) / 1e3, (en - st) / 1e3,
                                           (st - prev) / 1e3))
    prev = en
print("iteration span %.1f us (launches %d)" % ((int(rows[i1]["End_Timestamp"]) - t0) / 1e3, i1 - i0))
