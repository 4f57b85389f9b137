"""Where the bench set-up goes (cavity level 6 by default): wall time per
C-ABI entry point and per producer function, on the GPU box.

    python tools/setup_breakdown.py [--level 6] [--geometry cavity] [--profile]
"""
import argparse
import collections
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--geometry", default="cavity")
    ap.add_argument("--n0", type=int, default=4)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--algebraic", action="store_true")
    ap.add_argument("--bind", action="store_true",
                    help="OMP_PROC_BIND=spread OMP_PLACES=cores, as bench.py")
    args = ap.parse_args()
    if args.bind:
        os.environ.setdefault("OMP_PROC_BIND", "spread")
        os.environ.setdefault("OMP_PLACES", "cores")
    t_imp = time.time()
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    from fenapack_amd import PETScOptions
    from fenapack_amd import _cabi as c
    from fenapack_amd.driver import make_solver, multigrid_inner_options
    from fenapack_amd.fem import Cavity, Cavity3D
    t_imp = time.time() - t_imp

    per = collections.OrderedDict()
    orig = c.Engine._call

    def timed_call(self, name, *a):
        t = time.perf_counter()
        try:
            return orig(self, name, *a)
        finally:
            r = per.setdefault(name, [0, 0.0])
            r[0] += 1
            r[1] += time.perf_counter() - t
    c.Engine._call = timed_call
    marks = []

    def run():
        t0 = time.time()
        if args.geometry == "cavity":
            pb = Cavity(args.level, nu=0.01, variant="BRM1")
        else:
            pb = Cavity3D(args.level, nu=0.01, n0=args.n0, variant="BRM1")
        marks.append(("problem", time.time() - t0))
        t0 = time.time()
        PETScOptions.clear()
        multigrid_inner_options(dim=pb.space.dim, algebraic=args.algebraic)
        w, nls, nlp = make_solver(pb, gmres_rtol=1e-6, restart=150,
                                  newton_rtol=0.0, max_newton=args.steps,
                                  device=0)
        nls.parameters["absolute_tolerance"] = 0.0
        nls.parameters["error_on_nonconvergence"] = False
        marks.append(("make_solver", time.time() - t0))
        t0 = time.time()
        nls.solve(nlp, w.vector(), on_update=w.touch)
        torch.cuda.synchronize()
        marks.append(("nonlinear solve (%d steps, its %s)"
                      % (args.steps, list(nls.krylov_history)),
                      time.time() - t0))

    pr = cProfile.Profile() if args.profile else None
    if pr:
        pr.enable()
    run()
    if pr:
        pr.disable()
    print("imports %.2f s" % t_imp)
    for k, v in marks:
        print("%-50s %7.3f s" % (k, v))
    print("set-up total %.3f s" % sum(v for _, v in marks))
    print("C-ABI entry points (calls, seconds):")
    for k, (n, s) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  %-28s %5d %8.3f" % (k, n, s))
    print("  total %.3f" % sum(s for _, s in per.values()))
    if pr:
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(r"fenapack_amd|numpy|scipy", 70)
        st.sort_stats("tottime").print_stats(45)


if __name__ == "__main__":
    main()
