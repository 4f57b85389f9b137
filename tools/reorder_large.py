#!/usr/bin/env python3
"""Renumbering beyond the caches: one PCApply on the lexicographic input, on
the same problem with every index space randomly permuted (the engine
renumbers it: csrc/pcd_reorder.hpp), and on the permuted input taken as it
comes (PCD_REORDER=none).  The suite's version of this
(tests/test_reorder_gpu.py) runs at cache-resident sizes only.

    python tools/reorder_large.py cavity 7        # 148k dofs, the headline mesh
    python tools/reorder_large.py cube 4 3        # N = 48: A00 ~ 0.9 GB

One JSON line per run."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np                                            # noqa: E402

import fenapack_amd                                           # noqa: E402,F401
from fenapack_amd import _cabi as c                           # noqa: E402
from fenapack_amd._guard import peak_rss_bytes                # noqa: E402
import test_reorder_gpu as T                                  # noqa: E402
from helpers import relerr                                    # noqa: E402


def main():
    kind, level = sys.argv[1], int(sys.argv[2])
    kw = {"n0": int(sys.argv[3])} if len(sys.argv) > 3 else {}
    t_start = time.time()
    st = T._state(kind, level, **kw)
    pb, V, L = st["pb"], st["V"], st["L"]
    d = V.dim
    nlev = level if kind == "cube" else level - 1             # coarsest: a few hundred nodes
    chain = pb.interpolations().chain("u", max(2, nlev))
    base = {"A": st["A"], "is_u": V.is_u, "is_p": V.is_p, "Ap": pb.Ap,
            "Mp": pb.Mp, "Kp": st["Kp"], "bc_idx": pb.bc_p_idx}
    perm = T._permuted(st, 7)
    rng = np.random.default_rng(8)
    sizes = [chain[1].shape[1]] + [P.shape[0] for P in chain[1:]]
    lp = [perm["tu"] if l == len(sizes) - 1 else
          (d * rng.permutation(n // d)[:, None] + np.arange(d)).ravel()
          for l, n in enumerate(sizes)]
    mg0 = T._hierarchy(L["A00"], chain)
    mg1 = T._renumbered(mg0, lp)
    lib = c.hip_library()
    os.environ.pop("PCD_REORDER", None)
    x = rng.standard_normal(V.ndof)
    out = {"tool": "reorder_large", "geometry": kind, "level": level,
           "n0": kw.get("n0"), "ndof": int(V.ndof), "levels": len(sizes),
           "a00_nnz": int(L["A00"].nnz)}
    e0 = T._engine(lib, st, base, mg0)
    y0 = e0.fieldsplit_apply_np(x)
    out["ms_lexicographic"] = 1e3 * T._time_applies(e0, V.ndof)
    out["reordered_lexicographic"] = int(e0.info(c.INFO_REORDERED))
    del e0
    t0 = time.time()
    e1 = T._engine(lib, st, perm, mg1)
    out["setup_s_renumbering"] = time.time() - t0
    out["reordered_permuted"] = int(e1.info(c.INFO_REORDERED))
    out["relerr_renumbered"] = relerr(e1.fieldsplit_apply_np(x[perm["sig"]]),
                                      y0[perm["sig"]])
    out["ms_permuted_renumbered"] = 1e3 * T._time_applies(e1, V.ndof)
    del e1
    os.environ["PCD_REORDER"] = "none"
    t0 = time.time()
    e2 = T._engine(lib, st, perm, mg1)
    out["setup_s_as_it_comes"] = time.time() - t0
    out["relerr_as_it_comes"] = relerr(e2.fieldsplit_apply_np(x[perm["sig"]]),
                                       y0[perm["sig"]])
    out["ms_permuted_as_it_comes"] = 1e3 * T._time_applies(e2, V.ndof, reps=10)
    out["peak_rss_gb"] = peak_rss_bytes() / 1e9
    out["wall_s"] = time.time() - t_start
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
