#!/usr/bin/env python3
"""Average FETCH_SIZE / WRITE_SIZE per launch and per kernel from the CSVs of
the two rocprofv3 --pmc passes; prints one JSON object."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import _guard                                      # noqa: E402
_guard.start_rss_watchdog()
from collections import defaultdict


def collect(root):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"),
                       recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            short = name.split("(")[0].replace("void ", "").strip()
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


out = {}
for root in sys.argv[1:]:
    for k, d in collect(root).items():
        for cn, v in d.items():
            out.setdefault(k, {})[cn] = {"mean": sum(v) / len(v), "n": len(v)}
print(json.dumps(out, indent=1))
