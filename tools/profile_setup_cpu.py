"""cProfile of the HOST side of the bench set-up, without a GPU.

A development tool (not product, not test): the engine behind the solver
stack is swapped for the CPU checker library of ``oracle/`` so that the Python
/ native producer - the part of set-up that does not need the device - can be
profiled in a container without one.  The numbers say nothing about the
engine's own hand-over.

    python tools/profile_setup_cpu.py [--level 6] [--geometry cavity] [--steps 2]
"""
import argparse
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
    ap.add_argument("--n0", type=int, default=None)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--algebraic", action="store_true",
                    help="-pc_type gamg (meshes without a nested hierarchy)")
    args = ap.parse_args()

    import oracle
    from fenapack_amd import _cabi as c
    c._hip_library = oracle.library()          # tool-only switch
    from fenapack_amd import PETScOptions
    from fenapack_amd.driver import make_solver, multigrid_inner_options
    from fenapack_amd.fem import Cavity, Cavity3D

    def run():
        t0 = time.time()
        if args.geometry == "cavity":
            pb = Cavity(args.level, nu=0.01, variant="BRM1")
        else:
            pb = Cavity3D(args.level, nu=0.01, n0=args.n0, variant="BRM1")
        t1 = time.time()
        PETScOptions.clear()
        multigrid_inner_options(dim=pb.space.dim, algebraic=args.algebraic)
        w, nls, nlp = make_solver(pb, gmres_rtol=1e-6, restart=150,
                                  newton_rtol=0.0, max_newton=args.steps,
                                  device=0)
        nls.parameters["absolute_tolerance"] = 0.0
        nls.parameters["error_on_nonconvergence"] = False
        nls.solve(nlp, w.vector(), on_update=w.touch)
        t2 = time.time()
        print("problem %.2f s, solver set-up + %d steps %.2f s, its %s"
              % (t1 - t0, args.steps, t2 - t1, list(nls.krylov_history)))

    if args.no_profile:
        run()
        return
    pr = cProfile.Profile()
    pr.enable()
    run()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(args.top)
    st.sort_stats("tottime").print_stats(args.top)


if __name__ == "__main__":
    main()
