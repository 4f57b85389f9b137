"""CPU suite: the partitioned operator producer (fenapack_amd/fem/partition.py)
against the global build - every rank assembles its rows only, and those rows
are BITWISE the rows of the global build (same cells in the same order, same
element matrices).  What the reference gets from DOLFIN's partitioned assembly
(fenapack/_field_split_utils.py:39-50, fenapack/SubfieldBC.h:136-155)."""
import threading

import numpy as np
import pytest
import scipy.sparse as sp

from fenapack_amd.fem import BackwardStep, Cavity, Cavity3D, Channel3D
from fenapack_amd.fem import partition as pt

CASES = {
    "cube16": (Cavity3D, dict(level=2, nu=0.01, n0=4)),
    "cavity4": (Cavity, dict(level=4, nu=0.01)),
    "lshape3_brm2": (BackwardStep, dict(level=3, nu=0.02, variant="BRM2")),
    # 3-D inflow / outflow: the Robin term of Kp over the inflow FACES
    "duct16_brm2": (Channel3D, dict(level=2, nu=0.02, n0=4, variant="BRM2")),
}
_GLOBAL = {}


def global_build(name):
    if name not in _GLOBAL:
        cls, kw = CASES[name]
        kw = dict(kw)
        pb = cls(kw.pop("level"), **kw)
        rng = np.random.default_rng(3)
        V = pb.space
        xu = 0.1 * rng.standard_normal(V.n_u)
        xp = 0.1 * rng.standard_normal(V.n_p)
        lin = pb.linearise(xu, xp)
        _GLOBAL[name] = dict(pb=pb, xu=xu, xp=xp, lin=lin, Kp=pb.Kp(xu),
                             interp=pb.interpolations())
    return _GLOBAL[name]


def rows_equal(M_part, M_glob, own):
    """Owned rows bitwise equal; every other row of the partitioned matrix
    empty."""
    A, B = sp.csr_matrix(M_part), sp.csr_matrix(M_glob)
    assert A.shape == B.shape
    r0, r1 = own
    a, b = A[r0:r1], B[r0:r1]
    assert np.array_equal(a.indptr, b.indptr)
    assert np.array_equal(a.indices, b.indices)
    assert np.array_equal(a.data, b.data)          # bitwise
    assert A.nnz == a.nnz, "rows outside the owned range are populated"


def test_cut_is_the_engine_s_rule():
    # pcd_dist.hpp Space::cut
    assert pt.cut(10, 3) == [0, 3, 6, 10]
    assert pt.cut(14, 4, 2) == [0, 2, 6, 10, 14]
    assert pt.cut(9529569, 8, 3)[1] % 3 == 0
    for n, R, b in ((1000, 7, 3), (12, 8, 2), (5, 8, 1)):
        c = pt.cut(n - n % b, R, b)
        assert c[0] == 0 and c[-1] == n - n % b and sorted(c) == c


@pytest.mark.parametrize("name,R", [("cube16", 2), ("cube16", 3),
                                    ("cube16", 8), ("cavity4", 2),
                                    ("cavity4", 3), ("cavity4", 8),
                                    ("lshape3_brm2", 3),
                                    ("duct16_brm2", 2), ("duct16_brm2", 3)])
def test_owned_rows_are_bitwise_the_global_build(name, R):
    g = global_build(name)
    pb, lin = g["pb"], g["lin"]
    V = pb.space
    cls, kw = CASES[name]
    seen_u = np.zeros(V.n_u, dtype=int)
    for r in range(R):
        pp = pt.partitioned(cls, r, R, **kw)
        f = pp.fine
        assert pp.space.ndof == V.ndof
        # the local space is a fraction of the mesh (slab + halo layer)
        if R >= 3 and name != "lshape3_brm2":
            assert f.sub.cells.size < 0.75 * V.mesh.num_cells
        seen_u[f.own_u[0]:f.own_u[1]] += 1
        L = pp.linearise(g["xu"], g["xp"])
        rows_equal(L["A00"], lin["A00"], f.own_u)
        rows_equal(L["A01"], lin["A01"], f.own_u)
        rows_equal(L["A10"], lin["A10"], f.own_p)
        rows_equal(pp.Ap, pb.Ap, f.own_p)
        rows_equal(pp.Mp, pb.Mp, f.own_p)
        rows_equal(pp.Kp(g["xu"]), g["Kp"], f.own_p)
        for k, own in (("bu", f.own_u), ("bp", f.own_p)):
            assert np.array_equal(L[k][own[0]:own[1]], lin[k][own[0]:own[1]])
            rest = np.ones(L[k].size, bool)
            rest[own[0]:own[1]] = False
            assert not L[k][rest].any()
        # boundary sets are the global ones
        assert np.array_equal(pp.bc_u_idx, pb.bc_u_idx)
        assert np.array_equal(pp.bc_p_idx, pb.bc_p_idx)
        assert np.array_equal(pp.bc_u_values(0.0), pb.bc_u_values(0.0))
        # the finest prolongation: owned rows bitwise, and the restriction
        # rows of the owned coarse dofs complete
        Pg_u, Pg_p = g["interp"].velocity[-1], g["interp"].pressure[-1]
        I = pp.interpolations()
        Pu, Pp = sp.csr_matrix(I.velocity[-1]), sp.csr_matrix(I.pressure[-1])
        for P, Pg, own in ((Pu, Pg_u, f.own_u), (Pp, Pg_p, f.own_p)):
            a, b = P[own[0]:own[1]], sp.csr_matrix(Pg)[own[0]:own[1]]
            assert np.array_equal(a.indptr, b.indptr)
            assert np.array_equal(a.indices, b.indices)
            assert np.array_equal(a.data, b.data)
        if Pg_u.shape[1] > pt.replicate_below():
            c = pt.cut(Pg_u.shape[1], R, V.dim)
            Rt = sp.csr_matrix(Pu.T)[c[r]:c[r + 1]]
            Rg = sp.csr_matrix(Pg_u.T)[c[r]:c[r + 1]]
            Rt.sort_indices(), Rg.sort_indices()
            assert np.array_equal(Rt.indices, Rg.indices)
            assert np.array_equal(Rt.data, Rg.data)
        # the coarser prolongations are built whole
        for l in range(1, len(I.velocity) - 1):
            assert (sp.csr_matrix(I.velocity[l])
                    != sp.csr_matrix(g["interp"].velocity[l])).nnz == 0
    assert np.all(seen_u == 1)                 # the cuts partition the rows


def test_restriction_rows_when_the_coarse_level_is_partitioned(monkeypatch):
    monkeypatch.setenv("PCD_REPLICATE_BELOW", "500")
    g = global_build("cavity4")
    cls, kw = CASES["cavity4"]
    V, R = g["pb"].space, 3
    Pg_u = sp.csr_matrix(g["interp"].velocity[-1])
    for r in range(R):
        pp = pt.partitioned(cls, r, R, **kw)
        Pu = sp.csr_matrix(pp.interpolations().velocity[-1])
        c = pt.cut(Pg_u.shape[1], R, V.dim)
        Rt, Rg = sp.csr_matrix(Pu.T)[c[r]:c[r + 1]], \
            sp.csr_matrix(Pg_u.T)[c[r]:c[r + 1]]
        Rt.sort_indices(), Rg.sort_indices()
        assert np.array_equal(Rt.indptr, Rg.indptr)
        assert np.array_equal(Rt.indices, Rg.indices)
        assert np.array_equal(Rt.data, Rg.data)


def test_rediscretised_coarse_operators_by_rows(monkeypatch):
    """-pc_mg_galerkin none: every level assembled from the injected iterate,
    partitioned levels as owned rows, small ones whole - no communication."""
    monkeypatch.setenv("PCD_REPLICATE_BELOW", "2000")
    g = global_build("cavity4")
    pb, xu = g["pb"], g["xu"]
    cls, kw = CASES["cavity4"]
    ops_g = pb.coarse_velocity_operators(xu, 4)
    R = 2
    for r in range(R):
        pp = pt.partitioned(cls, r, R, **kw)
        ops = pp.coarse_velocity_operators(xu, 4)
        assert len(ops) == len(ops_g) == 3
        for A, Ag in zip(ops, ops_g):
            n = Ag.shape[0]
            if n > 2000:
                c = pt.cut(n, R, 2)
                a = sp.csr_matrix(A)[c[r]:c[r + 1]]
                b = sp.csr_matrix(Ag)[c[r]:c[r + 1]]
                assert np.array_equal(a.indptr, b.indptr)
                assert np.array_equal(a.indices, b.indices)
                # (last-bit: BLAS batches, see the multigrid push test)
                assert abs(a - b).max() <= 4e-16 * abs(b).max()
                assert A.nnz == a.nnz
            else:
                assert (sp.csr_matrix(A) != sp.csr_matrix(Ag)).nnz == 0


def test_thread_host_comm_reductions():
    R = 4
    comms = pt.ThreadHostComm.group(R)
    out = [None] * R

    def body(r):
        v = np.zeros(8)
        v[2 * r:2 * r + 2] = r + 1.0
        out[r] = (comms[r].allgather(r * r), comms[r].sum(v),
                  comms[r].sum(float(r)), comms[r].max(r))

    th = [threading.Thread(target=body, args=(r,)) for r in range(R)]
    [t.start() for t in th]
    [t.join(60) for t in th]
    for r in range(R):
        assert out[r][0] == [0, 1, 4, 9]
        assert np.array_equal(out[r][1], np.repeat([1.0, 2, 3, 4], 2))
        assert out[r][2] == 6.0 and out[r][3] == 3
    assert pt.HostComm().sum(2.5) == 2.5


def _threads(R, body):
    errs = []

    def run(r):
        try:
            body(r)
        except Exception as ex:                       # pragma: no cover
            import traceback
            errs.append((r, repr(ex), traceback.format_exc()))
    th = [threading.Thread(target=run, args=(r,)) for r in range(R)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert not errs, errs


@pytest.mark.parametrize("R", [1, 2, 5])
def test_sum_rows_gives_each_owner_its_rows_of_the_sum(R):
    """``HostComm.sum_rows``: rank r contributes a global-shaped matrix with
    entries anywhere; every rank gets the rows of the SUM that it owns."""
    n = 23
    rng = np.random.default_rng(5)
    Ms = [sp.random(n, 9, density=0.3, random_state=rng.integers(1 << 30),
                    format="csr") for _ in range(R)]
    Ms[-1] = sp.csr_matrix((n, 9)) if R > 1 else Ms[-1]    # an empty term
    total = sum(Ms[1:], Ms[0]).toarray()
    cuts = pt.cut(n, R)
    comms = pt.ThreadHostComm.group(R) if R > 1 else [pt.HostComm()]
    got = [None] * R

    def body(r):
        got[r] = comms[r].sum_rows(Ms[r], cuts)

    _threads(R, body)
    for r in range(R):
        a, b = cuts[r], cuts[r + 1]
        G = got[r]
        assert G.shape == (n, 9) and G.has_sorted_indices
        assert np.abs(G[a:b].toarray() - total[a:b]).max() < 1e-15
        assert G.nnz == G[a:b].nnz                 # own rows only
    with pytest.raises(ValueError):
        comms[0].sum_rows(Ms[0], [0, n + 1] if R == 1 else cuts[:-1])


@pytest.mark.parametrize("R", [2, 3])
def test_reaction_laplacian_of_pcdr_by_rows(R):
    """PCDR's ``R_p = B diag(M_u)^-1 B^T`` from a partitioned producer: the
    product sums over VELOCITY rows, so every rank holds terms of pressure
    rows beside its cut - they travel to the owner (``sum_rows``)."""
    kw = dict(level=3, nu=0.02, dt=0.2, pcdr=True)
    pb = BackwardStep(3, nu=0.02, dt=0.2, pcdr=True)
    Rg = sp.csr_matrix(pb.Rp())
    comms = pt.ThreadHostComm.group(R)
    got = [None] * R

    def body(r):
        pp = pt.partitioned(BackwardStep, r, R, host=comms[r], **kw)
        rows_equal(pp.Mu(), pb.Mu(), pp.fine.own_u)
        got[r] = (pp.Rp(), pp.fine.own_p)

    _threads(R, body)
    scale = np.abs(Rg.data).max()
    for Rr, (a, b) in got:
        assert Rr.shape == Rg.shape and Rr.nnz == Rr[a:b].nnz
        D = (Rr[a:b] - Rg[a:b]).tocsr()
        # (terms summed in another order: round-off, not bits)
        assert (np.abs(D.data).max() if D.nnz else 0.0) < 1e-14 * scale
        # same pattern up to explicit zeros
        assert np.array_equal((Rr[a:b] != 0).indices, (Rg[a:b] != 0).indices)


def test_partitioned_norm_is_the_global_norm():
    g = global_build("cavity4")
    cls, kw = CASES["cavity4"]
    R = 3
    comms = pt.ThreadHostComm.group(R)
    b = g["lin"]["bu"]
    got = [None] * R

    def body(r):
        pp = pt.partitioned(cls, r, R, host=comms[r], **kw)
        L = pp.linearise(g["xu"], g["xp"])
        got[r] = pp.norm(L["bu"])

    th = [threading.Thread(target=body, args=(r,)) for r in range(R)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert len(set(got)) == 1
    assert abs(got[0] - np.linalg.norm(b)) < 1e-12 * np.linalg.norm(b)


def _recording_engine(rank, R, dim):
    class Lib(object):
        hip = False

    class FakeEngine(object):
        local_handover = True
        velocity_block = dim
        L = Lib()

        def __init__(self):
            self.calls = []

        def row_range(self, n_global, velocity=False):
            c = pt.cut(n_global, R, dim if velocity else 1)
            return (c[rank], c[rank + 1])

        def __getattr__(self, name):
            if name == "producer":
                raise AttributeError(name)

            def record(*a, **k):
                self.calls.append((name,) + a)
            return record
    return FakeEngine()


@pytest.mark.parametrize("R", [2, 3])
def test_multigrid_push_from_a_partitioned_producer(monkeypatch, R):
    """petsc._push_multigrid fed by a partitioned producer: the rows handed to
    pcd_mg_set_level_local are the global build's, the smoother bounds are the
    global estimates (power iteration reduced over the ranks), replicated
    levels go over whole - on every rank, ranks as threads."""
    from fenapack_amd import PETScOptions, _cabi as c
    from fenapack_amd.petsc import KSP, Mat
    monkeypatch.setenv("PCD_REPLICATE_BELOW", "700")
    cls, kw = CASES["cavity4"]
    g = global_build("cavity4")
    pb, xu = g["pb"], g["xu"]
    V = pb.space

    def push(problem, eng, A00):
        PETScOptions.clear()
        PETScOptions.set("pc_mg_galerkin", "none")
        k = KSP()
        k.setType("richardson")
        k.pc.setType("mg")
        k.setFromOptions()
        k.setOperators(Mat(A00))
        k.pc.setMGInterpolations(problem.interpolations().chain("u"))
        k.pc.setMGOperators(lambda nlev: problem.coarse_velocity_operators(
            xu, nlev))
        k.bind(eng, c.KSP_A00)
        k.setUp()
        return k

    # the global hand-over, one rank
    class One(object):
        pass
    e1 = _recording_engine(0, 1, 2)
    e1.local_handover = False
    k1 = push(pb, e1, g["lin"]["A00"])
    ref = {a[2]: a for a in e1.calls if a[0] == "mg_set_level"}
    ref_bounds = k1.pc.mg_data["bounds"]

    comms = pt.ThreadHostComm.group(R)
    out, errs = [None] * R, []
    lock = threading.Lock()

    def body(r):
        try:
            pp = pt.partitioned(cls, r, R, host=comms[r], **kw)
            eng = _recording_engine(r, R, 2)
            eng.producer = pp
            L = pp.linearise(g["xu"], g["xp"])
            with lock:                      # (PETScOptions is process-global)
                pass
            k = push(pp, eng, L["A00"])
            out[r] = (eng.calls, k.pc.mg_data["bounds"],
                      [o.shape[0] for o in k.pc.mg_data["ops"]])
        except Exception:                               # pragma: no cover
            import traceback
            errs.append(traceback.format_exc())
            comms[r]._sh.barrier.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(R)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert not errs, errs[0]
    for r in range(R):
        calls, bounds, sizes = out[r]
        Lv = len(sizes)
        for l in range(1, Lv):
            assert abs(bounds[l][1] - ref_bounds[l][1]) <= 1e-12 * ref_bounds[l][1]
        local = {a[2]: a for a in calls if a[0] == "mg_set_level_local"}
        whole = {a[2]: a for a in calls if a[0] == "mg_set_level"}
        for l in range(1, Lv):
            part = l == Lv - 1 or sizes[l] > 700
            assert (l in local) == part and (l in whole) == (not part), l
            c0 = pt.cut(sizes[l], R, 2)
            if part:
                _, slot, lev, ng, A_rows, P_rows, R_rows, emin, emax = local[l]
                Ag, Pg = ref[l][3], sp.csr_matrix(ref[l][4])
                if A_rows is not None:
                    # (convection element matrices are BLAS GEMMs over the
                    # cells of a level: the last bit of a row can depend on
                    # where its cell sits in the batch)
                    Agr = sp.csr_matrix(Ag)[c0[r]:c0[r + 1]]
                    assert np.array_equal(A_rows.indices, Agr.indices)
                    assert abs(A_rows - Agr).max() <= 4e-16 * abs(Agr).max()
                assert (P_rows != Pg[c0[r]:c0[r + 1]]).nnz == 0
                if sizes[l - 1] > 700:
                    cc = pt.cut(sizes[l - 1], R, 2)
                    assert (R_rows != sp.csr_matrix(Pg.T)[cc[r]:cc[r + 1]]).nnz == 0
            else:
                assert (sp.csr_matrix(whole[l][3]) != sp.csr_matrix(ref[l][3])).nnz == 0
        # the explicit coarse inverse, whole and equal
        assert np.allclose(whole[0][3].toarray(), ref[0][3].toarray(),
                           rtol=1e-10, atol=1e-14)
    PETScOptions.clear()


@pytest.mark.parametrize("R", [1, 3])
def test_partitioned_smoothed_aggregation(R):
    """amg.PartitionedSA on thread ranks: every rank aggregates its own rows,
    owns the coarse dofs of its aggregates (uneven cuts), and the operators it
    holds are rows of P^T A P of the hierarchy the ranks built together - the
    Galerkin identity checked on the assembled whole.  One rank: the first
    coarsening is the global builder's."""
    from fenapack_amd import amg, _host
    pb = Cavity3D(2, nu=0.01, n0=4)                       # cube N = 16
    L = pb.linearise(*pb.initial_guess())
    F = _host.kron_factor(sp.csr_matrix(L["A00"]), 3)
    n = F.shape[0]
    comms = pt.ThreadHostComm.group(R) if R > 1 else [pt.HostComm()]
    cuts = pt.cut(n, R, 1)
    res, errs = [None] * R, []

    def body(r):
        try:
            r0, r1 = cuts[r], cuts[r + 1]
            ip = np.zeros(n + 1, dtype=np.int64)
            ip[r0 + 1:r1 + 1] = np.diff(F.indptr)[r0:r1]
            np.cumsum(ip, out=ip)
            Fr = sp.csr_matrix((F.data[F.indptr[r0]:F.indptr[r1]],
                                F.indices[F.indptr[r0]:F.indptr[r1]], ip),
                               shape=F.shape)
            psa = amg.PartitionedSA(Fr, (r0, r1), comms[r], block=3,
                                    replicate_rows=3000, coarse_rows=300)
            res[r] = (psa, psa.operators(Fr))
        except Exception:                               # pragma: no cover
            import traceback
            errs.append(traceback.format_exc())
            if R > 1:
                comms[r]._sh.barrier.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(R)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert not errs, errs[0]
    psa0 = res[0][0]
    Lv = psa0.nlevels
    flags = psa0.partitioned_levels()
    assert flags[-1] and not flags[0] and Lv >= 3
    cl = psa0.level_cuts()
    if R > 1:
        # (35 937 nodes -> ~1000 aggregates x 3 components = the one coarse
        # level above the replication limit: partitioned, cut by aggregates)
        assert sum(flags) == 2
        cut1 = [c for c in cl if c is not None][0]
        assert cut1[0] == 0 and len(cut1) == R + 1 and np.all(np.diff(cut1) > 0)
    else:
        ref = amg.smoothed_aggregation_chain(F, block=1, coarse_rows=300)
        assert psa0.chain()[-1].shape == ref[-1].shape
    chains = [res[r][0].chain() for r in range(R)]
    opss = [res[r][1] for r in range(R)]

    def own_rows(r, l):
        # the prolongator handed over holds my rows AND my halo rows (the
        # restriction rows of my coarse dofs need the neighbours' entries)
        lev = res[r][0].part[Lv - 1 - l]
        a, b = lev["own"]
        P = sp.csr_matrix(chains[r][l])
        ip = np.zeros(P.shape[0] + 1, dtype=np.int64)
        ip[a + 1:b + 1] = np.diff(P.indptr)[a:b]
        np.cumsum(ip, out=ip)
        lo, hi = P.indptr[a], P.indptr[b]
        return sp.csr_matrix((P.data[lo:hi], P.indices[lo:hi], ip),
                             shape=P.shape)

    Pg = [None] + [sum(own_rows(r, l) for r in range(R)) if flags[l]
                   else chains[0][l] for l in range(1, Lv)]
    for l in range(1, Lv):
        if not flags[l]:
            continue
        for r in range(R):
            lev = res[r][0].part[Lv - 1 - l]
            a, b = lev["own"]
            c0, c1 = lev["own_c"]
            mine = sp.csr_matrix(chains[r][l])
            # my rows, and every entry of the whole P in my coarse columns
            assert abs(mine[a:b] - Pg[l][a:b]).max() == 0.0
            assert abs(mine.T.tocsr()[c0:c1]
                       - Pg[l].T.tocsr()[c0:c1]).max() == 0.0
    if R > 1:
        # smoothed across the cuts: P couples the ranks
        a, b = res[0][0].part[0]["own"]
        c0, c1 = res[0][0].part[0]["own_c"]
        P0 = sp.csr_matrix(Pg[Lv - 1])[a:b].tocsc()
        assert P0[:, c1:].nnz > 0
    Og = [sum(opss[r][l] for r in range(R)) if flags[l] else opss[0][l]
          for l in range(Lv)]
    assert abs(Og[-1] - F).max() == 0.0               # the rows partition F
    for l in range(Lv - 1, 0, -1):
        C = (Pg[l].T @ Og[l] @ Pg[l]).tocsr()
        assert abs(C - Og[l - 1]).max() <= 1e-12 * abs(Og[l - 1]).max(), l
        # replicated levels are the same on every rank
        if not flags[l - 1]:
            for r in range(1, R):
                assert (sp.csr_matrix(opss[r][l - 1])
                        != sp.csr_matrix(opss[0][l - 1])).nnz == 0


def test_partitioned_sa_refuses_a_structurally_unsymmetric_pattern():
    """Smoothing across the cuts takes the restriction rows of a rank's coarse
    dofs from its halo rows of P - complete only for a structurally symmetric
    operator.  A one-sided coupling across the cut must be refused (by every
    rank), not restricted wrongly."""
    from fenapack_amd import amg
    n, R = 400, 2
    T = sp.diags([-1.0, 2.5, -1.0], [-1, 0, 1], shape=(n, n)).tolil()
    T[n // 2 - 3, n // 2 + 40] = -0.7           # rank 0 -> rank 1 only
    F = sp.csr_matrix(T)
    comms = pt.ThreadHostComm.group(R)
    cuts = pt.cut(n, R)
    seen = [None] * R

    def body(r):
        r0, r1 = cuts[r], cuts[r + 1]
        ip = np.zeros(n + 1, dtype=np.int64)
        ip[r0 + 1:r1 + 1] = np.diff(F.indptr)[r0:r1]
        np.cumsum(ip, out=ip)
        Fr = sp.csr_matrix((F.data[F.indptr[r0]:F.indptr[r1]],
                            F.indices[F.indptr[r0]:F.indptr[r1]], ip),
                           shape=F.shape)
        try:
            amg.PartitionedSA(Fr, (r0, r1), comms[r], replicate_rows=50,
                              coarse_rows=20)
        except ValueError as ex:
            seen[r] = str(ex)
        # block-diagonal smoothing does not need the symmetry
        amg.PartitionedSA(Fr, (r0, r1), comms[r], replicate_rows=50,
                          coarse_rows=20, smooth="block")

    _threads(R, body)
    assert all(s and "structurally symmetric" in s for s in seen), seen


@pytest.mark.parametrize("R", [2, 3])
def test_galerkin_products_of_partitioned_levels(R):
    """``galerkin_chain(reduce_level=...)``: every rank forms ITS rows' terms
    of each coarse operator; a partitioned coarse level gets them through
    ``HostComm.sum_rows`` (rows to their owners), a replicated one through a
    plain sum - against the global products, level by level."""
    from fenapack_amd.fem.multigrid import galerkin_chain
    g = global_build("cavity4")
    pb = g["pb"]
    A = sp.csr_matrix(g["lin"]["A00"])
    chain = pb.interpolations().chain("u", 4)
    ref = galerkin_chain(A, chain)
    d, limit = 2, 3000
    comms = pt.ThreadHostComm.group(R)
    got = [None] * R
    n = A.shape[0]

    def body(r):
        cu = pt.cut(n, R, d)
        ip = np.zeros(n + 1, dtype=np.int64)
        ip[cu[r] + 1:cu[r + 1] + 1] = np.diff(A.indptr)[cu[r]:cu[r + 1]]
        np.cumsum(ip, out=ip)
        lo, hi = A.indptr[cu[r]], A.indptr[cu[r + 1]]
        Ar = sp.csr_matrix((A.data[lo:hi], A.indices[lo:hi], ip), shape=A.shape)

        def red(C, l):
            m = C.shape[0]
            if l >= 1 and m > limit:
                return comms[r].sum_rows(C, pt.cut(m, R, d)), True
            return comms[r].sum(C), False

        got[r] = galerkin_chain(Ar, chain, reduce_level=red)

    _threads(R, body)
    kinds = []
    for l in range(len(chain) - 1):
        G = sp.csr_matrix(ref[l])
        m = G.shape[0]
        scale = abs(G).max()
        part = l >= 1 and m > limit
        kinds.append(part)
        for r in range(R):
            M = sp.csr_matrix(got[r][l])
            if part:
                a, b = pt.cut(m, R, d)[r], pt.cut(m, R, d)[r + 1]
                assert M.nnz == M[a:b].nnz
                assert abs(M[a:b] - G[a:b]).max() <= 1e-13 * scale
            else:
                assert abs(M - G).max() <= 1e-13 * scale
    assert any(kinds) and not kinds[0]      # both kinds of level were there
