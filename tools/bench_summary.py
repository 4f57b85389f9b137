#!/usr/bin/env python3
"""One line per bench.py JSON file: rate, ms per PCApply, GMRES history, the
dominant kernel's launch time (back-to-back and in-cycle)."""
import json
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenapack_amd  # noqa: F401,E402  (resident-set watchdog of the repository's scripts)

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as exc:                                   # noqa: BLE001
        print("%s: %s" % (path, exc))
        continue
    r = d.get("roofline", {})
    ic = (r.get("in_cycle") or {}).get("us_per_launch")
    print("%-52s %8.1f /s %7.4f ms  its %s  %s: %.2f us (in cycle %s) model frac %s"
          % (path.split("/")[-1], d["value"], d["ms_per_step"],
             d.get("gmres_its_per_newton_step"), r.get("kernel", "?").split(" on ")[0],
             r.get("us_per_launch", float("nan")),
             "%.2f" % ic if ic else "-",
             "%.3f" % r["frac_kernel_model"] if r.get("frac_kernel_model") else "-"))
