#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/pmc_pcapply.py into the JSON
that bench.py quotes as `roofline.traffic` (and refuses to quote once
csrc/pcd_kernels.hpp has changed: the file carries the hash of the kernels it
was measured on).

    pmc_roofline.py OUT_FETCH OUT_WRITE [n_u] > profiles/rNN_pmc_roofline.json

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  Calibration and correction
as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3) prescribes:
separate passes; FETCH_SIZE reads one half of a streamed read on gfx950 - the
factor is measured here on k_scale_dinv (known 16 n bytes read, 8 n written),
not assumed; WRITE_SIZE is exact."""
import csv
import glob
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import _guard                                      # noqa: E402
_guard.start_rss_watchdog()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dispatches(root, counter):
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"),
                       recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "") \
                .replace("pcd::", "").strip()
            rows.append((int(r["Dispatch_Id"]), name, int(r["Grid_Size"]),
                         float(r["Counter_Value"]) * 1024.0,
                         int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    return rows


def kernels_sha16():
    h = hashlib.sha256()
    for f in ("pcd_kernels.hpp", "pcd_apply.hip", "pcd_setup.hip"):
        h.update(open(os.path.join(ROOT, "fenapack_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    F = dispatches(sys.argv[1], "FETCH_SIZE")
    W = dispatches(sys.argv[2], "WRITE_SIZE")
    assert [r[1] for r in F] == [r[1] for r in W], "passes ran different kernels"
    # (1) calibration on the 20 standalone k_scale_dinv of the finest level
    sd = [i for i, r in enumerate(F) if r[1] == "k_scale_dinv"]
    if not sd:
        names = {}
        for r in F:
            names[r[1]] = names.get(r[1], 0) + 1
        sys.exit("no k_scale_dinv dispatch in %s (%d dispatches: %s)"
                 % (sys.argv[1], len(F), sorted(names.items())[:40]))
    big = max(F[i][2] for i in sd)
    sd = [i for i in sd if F[i][2] == big][-20:]
    n_u = int(sys.argv[3]) if len(sys.argv) > 3 else None
    fetch_sd = sum(F[i][3] for i in sd) / len(sd)
    write_sd = sum(W[i][3] for i in sd) / len(sd)
    if n_u is None:
        # WRITE_SIZE is exact up to the 32-byte granule of the last store
        n_u = int(write_sd // 8.0)
    corr = 16.0 * n_u / fetch_sd
    # (2) the roofline kernel: k_cheb_step* with the largest grid
    ks = [i for i, r in enumerate(F) if r[1].startswith("k_cheb_step")]
    big = max(F[i][2] for i in ks)
    ks = [i for i in ks if F[i][2] == big]
    kname = F[ks[0]][1]
    kread = corr * sum(F[i][3] for i in ks) / len(ks)
    kwrite = sum(W[i][3] for i in ks) / len(ks)
    kus = sum(F[i][4] for i in ks) / len(ks) / 1e3
    # (3) whole PCApplies: k_gather ... k_scatter windows, the last ten
    ends = [i for i, r in enumerate(F) if r[1] == "k_scatter"][-10:]
    PEAK = 8.0e12
    starts = [max(j for j, r in enumerate(F[:e]) if r[1] == "k_gather")
              for e in ends]
    per = []
    for s, e in zip(starts, ends):
        rd = corr * sum(F[i][3] for i in range(s, e + 1))
        wr = sum(W[i][3] for i in range(s, e + 1))
        per.append((rd, wr, e - s + 1))
    # (4) every launch of the LAST apply (bytes, time under the counter pass,
    # fraction of 8 TB/s), and the same summed by kernel over the applies
    s, e = starts[-1], ends[-1]
    launches = []
    for i in range(s, e + 1):
        rd, wr, ns = corr * F[i][3], W[i][3], 0.5 * (F[i][4] + W[i][4])
        launches.append({"kernel": F[i][1], "grid": F[i][2],
                         "read_bytes": rd, "write_bytes": wr,
                         "us": ns / 1e3,
                         "gbs": (rd + wr) / max(ns, 1.0),
                         "frac": (rd + wr) / max(ns, 1.0) * 1e9 / PEAK})
    by = {}
    for s_, e_ in zip(starts, ends):
        for i in range(s_, e_ + 1):
            k = by.setdefault(F[i][1], [0, 0.0, 0.0])
            k[0] += 1
            k[1] += corr * F[i][3] + W[i][3]
            k[2] += 0.5 * (F[i][4] + W[i][4])
    by_kernel = [{"kernel": k, "launches_per_apply": v[0] / len(ends),
                  "traffic_bytes_per_apply": v[1] / len(ends),
                  "us_per_apply": v[2] / len(ends) / 1e3,
                  "frac": v[1] / max(v[2], 1.0) * 1e9 / PEAK}
                 for k, v in sorted(by.items(), key=lambda kv: -kv[1][2])]
    # (5) the Jacobi-PCG iteration (pmc_pcapply.py --inner jacobi, stage 4):
    # k_cg_spmv_s + k_cg_update of the fixed-count solves on Ap
    cg = None
    upd = [i for i, r in enumerate(F) if r[1] == "k_cg_update"]
    if upd:
        big_u = max(F[i][2] for i in upd)
        # the standalone solves ran 40 iterations each: take their launches
        # (largest run of consecutive spmv / update pairs)
        sp_ = [i for i, r in enumerate(F) if r[1].startswith("k_cg_spmv")]
        pair = [i for i in sp_ if i + 1 < len(F) and F[i + 1][1] == "k_cg_update"]
        n_it = len(pair)
        tr = sum(corr * (F[i][3] + F[i + 1][3]) + W[i][3] + W[i + 1][3]
                 for i in pair)
        ns = sum(0.5 * (F[i][4] + W[i][4] + F[i + 1][4] + W[i + 1][4])
                 for i in pair)
        cg = {"kernels": sorted({F[i][1] for i in pair} | {"k_cg_update"}),
              "iterations_counted": n_it, "grid_update": big_u,
              "launches_per_iteration": 2,
              "traffic_bytes_per_iteration": tr / n_it,
              "us_per_iteration_under_the_counter_pass": ns / n_it / 1e3}
    out = {
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace "
                   "--output-format csv -- python3 tools/pmc_pcapply.py "
                   "(two separate passes), summarised by tools/pmc_roofline.py",
        "kernels_sha16": kernels_sha16(),
        "n_u": n_u,
        "calibration": {"kernel": "k_scale_dinv", "launches": len(sd),
                        "known_read_bytes": 16 * n_u,
                        "known_write_bytes": 8 * n_u,
                        "read_correction": corr,
                        "write_check": write_sd / (8.0 * n_u)},
        "roofline_kernel": {
            "kernel": kname, "grid": big, "launches": len(ks),
            "read_bytes_per_launch": kread, "write_bytes_per_launch": kwrite,
            "traffic_bytes_per_launch": kread + kwrite,
            "us_per_launch_under_the_counter_pass": kus},
        "pcapply": {
            "applies": len(per),
            "launches_per_apply": per[-1][2],
            "read_bytes_per_apply": sum(p[0] for p in per) / len(per),
            "write_bytes_per_apply": sum(p[1] for p in per) / len(per),
            "traffic_bytes_per_apply": sum(p[0] + p[1] for p in per) / len(per)},
        "cg_iteration": cg,
        "pcapply_by_kernel": by_kernel,
        "pcapply_launches": launches,
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
