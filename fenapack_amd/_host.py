"""ctypes binding of ``libpcd_host.so`` (``include/pcd_host.h``): the native
(OpenMP) integer work of problem set-up - sparsity patterns, element -> entry
contribution lists, Galerkin gather plans, symbolic products.

The library has no HIP dependency, so the producer uses it on every box.  It
is part of the product build (``__graft_entry__.build()``); when it is missing
the producer fails loudly - there is no silent numpy fallback.  The numpy
implementations it replaced are kept as the *checker* of this library
(``FENAPACK_AMD_NUMPY_PRODUCER=1`` selects them; ``tests/test_host_native.py``
compares both on the same inputs)."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# (FENAPACK_AMD_HOST_LIBRARY: another build of the SAME sources - the ASan +
# UBSan one of csrc/Makefile, which the CPU suite runs its native tests on)
HOST_LIBRARY_PATH = os.environ.get(
    "FENAPACK_AMD_HOST_LIBRARY", os.path.join(HERE, "lib", "libpcd_host.so"))

_I64P = ctypes.POINTER(ctypes.c_int64)
_I32P = ctypes.POINTER(ctypes.c_int32)
_F64P = ctypes.POINTER(ctypes.c_double)


class HostError(RuntimeError):
    pass


def use_numpy():
    """The numpy restatement of the helpers (the checker), on request."""
    return os.environ.get("FENAPACK_AMD_NUMPY_PRODUCER", "0") == "1"


_lib = None


def library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HOST_LIBRARY_PATH):
        raise HostError("%s is missing: run `python -c 'import "
                        "__graft_entry__ as g; g.build()'` (g++ -fopenmp)"
                        % HOST_LIBRARY_PATH)
    L = ctypes.CDLL(HOST_LIBRARY_PATH)
    L.pcdh_last_error.restype = ctypes.c_char_p
    L.pcdh_group_nnz.restype = ctypes.c_int64
    L.pcdh_group_kept.restype = ctypes.c_int64
    L.pcdh_group_nnz.argtypes = [ctypes.c_void_p]
    L.pcdh_group_kept.argtypes = [ctypes.c_void_p]
    L.pcdh_group_free.argtypes = [ctypes.c_void_p]
    L.pcdh_group_free.restype = None
    L.pcdh_group_pairs.argtypes = [ctypes.c_int64, _I64P, _I64P,
                                   ctypes.c_int64, ctypes.c_int64,
                                   ctypes.c_int64,
                                   ctypes.POINTER(ctypes.c_void_p)]
    L.pcdh_pattern_cells.argtypes = [ctypes.c_int64, ctypes.c_int, _I64P,
                                     ctypes.c_int, _I64P, ctypes.c_int64,
                                     ctypes.c_int64, ctypes.c_int64,
                                     ctypes.POINTER(ctypes.c_void_p)]
    L.pcdh_group_export.argtypes = [ctypes.c_void_p, _I64P, _I64P, _I64P,
                                    _I64P, _I64P]
    L.pcdh_extract_count.argtypes = [ctypes.c_int64, _I32P, _I32P, _I32P,
                                     _I32P, _I32P]
    L.pcdh_extract_fill.argtypes = [ctypes.c_int64, _I32P, _I32P, _I32P,
                                    _I32P, _I32P, _I32P, _I64P]
    L.pcdh_transpose.argtypes = [ctypes.c_int64, ctypes.c_int64, _I32P, _I32P,
                                 _F64P, _I32P, _I32P, _F64P]
    L.pcdh_locate_blocks.argtypes = [ctypes.c_int64, _I32P, _I32P, ctypes.c_int,
                                     ctypes.c_int, _I32P, _I32P, ctypes.c_int64,
                                     _I64P, ctypes.c_int64, _I64P, _I32P, _I64P]
    L.pcdh_contribution_src.argtypes = [ctypes.c_int64, _I64P, ctypes.c_int64,
                                        ctypes.c_int64, _I32P]
    L.pcdh_take_segments.argtypes = [ctypes.c_int64, _I64P, ctypes.c_int,
                                     ctypes.c_void_p, _I64P, _F64P]
    L.pcdh_locate.argtypes = [ctypes.c_int64, _I64P, _I64P, ctypes.c_int64,
                              _I64P, _I32P, _I64P]
    L.pcdh_spgemm_count.argtypes = [ctypes.c_int64, ctypes.c_int64,
                                    ctypes.c_int64, _I32P, _I32P, _I32P,
                                    _I32P, _I64P]
    L.pcdh_product_plan_count.argtypes = [ctypes.c_int64, ctypes.c_int64,
                                          _I32P, _I32P, _I32P, _I32P, _I64P,
                                          _I64P]
    L.pcdh_product_plan_fill.argtypes = [ctypes.c_int64, ctypes.c_int64,
                                         _I32P, _I32P, _F64P, _I32P, _I32P,
                                         _F64P, ctypes.c_int, _I64P, _I64P,
                                         _I32P, _I64P, _I32P, _F64P]
    L.pcdh_spgemm_fill.argtypes = [ctypes.c_int64, ctypes.c_int64,
                                   ctypes.c_int64, _I32P, _I32P, _F64P, _I32P,
                                   _I32P, _F64P, _I64P, _I32P, _F64P]
    L.pcdh_set_threads.argtypes = [ctypes.c_int]
    L.pcdh_wind_gradlam.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, _I64P, _F64P, _F64P, _F64P,
                                    _F64P]
    L.pcdh_spmv.argtypes = [ctypes.c_int64, _I32P, _I32P, _F64P, _F64P, _F64P,
                            _F64P]
    L.pcdh_spmm.argtypes = [ctypes.c_int64, _I32P, _I32P, _F64P, ctypes.c_int,
                            _F64P, _F64P, _F64P]
    L.pcdh_kron_factor.argtypes = [ctypes.c_int64, _I32P, _I32P, _F64P,
                                   ctypes.c_int, _I32P, _I32P, _F64P]
    L.pcdh_kron_expand.argtypes = [ctypes.c_int64, _I32P, _I32P, _F64P,
                                   ctypes.c_int, _I32P, _I32P, _F64P]
    L.pcdh_gather_sum.argtypes = [ctypes.c_int64, _I64P, _I64P, _F64P, _F64P]
    PP32 = ctypes.POINTER(_I32P)
    L.pcdh_union_count.argtypes = [ctypes.c_int64, ctypes.c_int, _I64P, PP32,
                                   PP32, _I64P]
    L.pcdh_union_fill.argtypes = [ctypes.c_int64, ctypes.c_int, _I64P, PP32,
                                  PP32, PP32, PP32, _I64P, _I64P, _I32P, _I64P]
    L.pcdh_mis2_degrees.argtypes = [ctypes.c_int64, _I32P, _I32P, _I64P]
    L.pcdh_mis2.argtypes = [ctypes.c_int64, _I32P, _I32P, _F64P,
                            ctypes.POINTER(ctypes.c_int8),
                            ctypes.POINTER(ctypes.c_int64)]
    _lib = L
    nt = os.environ.get("FENAPACK_AMD_HOST_THREADS")
    if nt:
        L.pcdh_set_threads(int(nt))
    return L


def _chk(rc):
    if rc:
        raise HostError(library().pcdh_last_error().decode())


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Group(object):
    """Result of a grouping: ``indptr`` (rows kept + 1), ``ucols`` (column of
    every group), and on demand ``inv`` (group of every input pair),
    ``ptr`` / ``order`` (members of every group in ascending input order)."""

    def __init__(self, handle, n, nrows_kept):
        self._h = handle
        self.n, self.nrows = n, nrows_kept
        L = library()
        self.nnz = int(L.pcdh_group_nnz(handle))
        self.kept = int(L.pcdh_group_kept(handle))
        self.indptr = np.empty(nrows_kept + 1, dtype=np.int64)
        self.ucols = np.empty(self.nnz, dtype=np.int64)
        _chk(L.pcdh_group_export(handle, _p(self.indptr, _I64P),
                                 _p(self.ucols, _I64P), None, None, None))
        self._inv = self._ptr = self._order = None

    @property
    def inv(self):
        if self._inv is None:
            self._inv = np.empty(self.n, dtype=np.int64)
            _chk(library().pcdh_group_export(self._h, None, None,
                                             _p(self._inv, _I64P), None, None))
        return self._inv

    def members(self):
        if self._ptr is None:
            self._ptr = np.empty(self.nnz + 1, dtype=np.int64)
            self._order = np.empty(self.kept, dtype=np.int64)
            _chk(library().pcdh_group_export(self._h, None, None, None,
                                             _p(self._ptr, _I64P),
                                             _p(self._order, _I64P)))
        return self._ptr, self._order

    def release(self):
        """Drop the native copy (the exported numpy arrays stay)."""
        if self._h:
            library().pcdh_group_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def group_pairs(rows, cols, nrows, row0=0, row1=None):
    """Group ``(rows[i], cols[i])``: see ``pcdh_group_pairs``."""
    rows = _i64(np.asarray(rows).ravel())
    cols = None if cols is None else _i64(np.asarray(cols).ravel())
    row1 = nrows if row1 is None else row1
    h = ctypes.c_void_p()
    _chk(library().pcdh_group_pairs(rows.size, _p(rows, _I64P),
                                    _p(cols, _I64P), nrows, row0, row1,
                                    ctypes.byref(h)))
    return Group(h, rows.size, row1 - row0)


def pattern_cells(rdofs, cdofs, nrows, row0=0, row1=None):
    """Pattern of a form from cell dof tables ``rdofs`` (ncell, nr) x
    ``cdofs`` (ncell, nc); element entries are (cell, a, b), b fastest."""
    rdofs, cdofs = _i64(rdofs), _i64(cdofs)
    ncell, nr = rdofs.shape
    nc = cdofs.shape[1]
    assert cdofs.shape[0] == ncell
    row1 = nrows if row1 is None else row1
    h = ctypes.c_void_p()
    _chk(library().pcdh_pattern_cells(ncell, nr, _p(rdofs, _I64P), nc,
                                      _p(cdofs, _I64P), nrows, row0, row1,
                                      ctypes.byref(h)))
    return Group(h, ncell * nr * nc, row1 - row0)


def extract_block(rows, rowptr, col, colmap):
    """Sub-matrix (rows, columns renumbered by ``colmap``; -1 drops) with
    value provenance: returns (rowptr, col, src)."""
    rows, rowptr, col, colmap = _i32(rows), _i32(rowptr), _i32(col), \
        _i32(colmap)
    L = library()
    orp = np.empty(rows.size + 1, dtype=np.int32)
    _chk(L.pcdh_extract_count(rows.size, _p(rows, _I32P), _p(rowptr, _I32P),
                              _p(col, _I32P), _p(colmap, _I32P),
                              _p(orp, _I32P)))
    oc = np.empty(int(orp[-1]), dtype=np.int32)
    osrc = np.empty(int(orp[-1]), dtype=np.int64)
    _chk(L.pcdh_extract_fill(rows.size, _p(rows, _I32P), _p(rowptr, _I32P),
                             _p(col, _I32P), _p(colmap, _I32P),
                             _p(orp, _I32P), _p(oc, _I32P), _p(osrc, _I64P)))
    return orp, oc, osrc


def spgemm(A, B, row0=0, row1=None):
    """Rows ``[row0, row1)`` of ``A @ B`` (scipy CSR in, scipy CSR out) with a
    STRUCTURAL pattern: entries that cancel to zero are kept, columns
    sorted."""
    import scipy.sparse as sp
    A, B = sp.csr_matrix(A), sp.csr_matrix(B)
    assert A.shape[1] == B.shape[0]
    row1 = A.shape[0] if row1 is None else row1
    arp, ac, av = _i32(A.indptr), _i32(A.indices), \
        np.ascontiguousarray(A.data, dtype=np.float64)
    brp, bc, bv = _i32(B.indptr), _i32(B.indices), \
        np.ascontiguousarray(B.data, dtype=np.float64)
    L = library()
    crp = np.empty(row1 - row0 + 1, dtype=np.int64)
    _chk(L.pcdh_spgemm_count(row0, row1, B.shape[1], _p(arp, _I32P),
                             _p(ac, _I32P), _p(brp, _I32P), _p(bc, _I32P),
                             _p(crp, _I64P)))
    nnz = int(crp[-1])
    cc = np.empty(nnz, dtype=np.int32)
    cv = np.empty(nnz, dtype=np.float64)
    _chk(L.pcdh_spgemm_fill(row0, row1, B.shape[1], _p(arp, _I32P),
                            _p(ac, _I32P), _p(av, _F64P), _p(brp, _I32P),
                            _p(bc, _I32P), _p(bv, _F64P), _p(crp, _I64P),
                            _p(cc, _I32P), _p(cv, _F64P)))
    idx = crp.astype(np.int32) if nnz < 2 ** 31 else crp
    C = sp.csr_matrix((cv, cc, idx), shape=(row1 - row0, B.shape[1]))
    C.has_sorted_indices = True
    return C


def take_segments(idx, arrays):
    """``numpy.concatenate(arrays)[idx]`` without the concatenation, on
    threads (``pcdh_take_segments``)."""
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in arrays]
    if use_numpy() or idx.size < 200000:
        return np.concatenate(arrs)[idx]
    off = np.zeros(len(arrs) + 1, dtype=np.int64)
    np.cumsum([a.size for a in arrs], out=off[1:])
    ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    out = np.empty(idx.size, dtype=np.float64)
    L = library()
    _chk(L.pcdh_take_segments(idx.size, _p(idx, _I64P), len(arrs), ptrs,
                              _p(off, _I64P), _p(out, _F64P)))
    return out


def locate(M, rows, cols):
    """Positions (in ``M.data`` order) of the entries ``(rows[q], cols[q])``
    of the scipy CSR matrix ``M`` (columns sorted in every row)."""
    import scipy.sparse as sp
    M = sp.csr_matrix(M)
    if not M.has_sorted_indices:
        raise ValueError("locate: the matrix's indices must be sorted")
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    if use_numpy():
        ip = M.indptr.astype(np.int64)
        key = np.repeat(np.arange(M.shape[0], dtype=np.int64),
                        np.diff(ip)) * M.shape[1] + M.indices
        q = rows * M.shape[1] + cols
        pos = np.searchsorted(key, q)
        if pos.max(initial=0) >= key.size or np.any(key[pos] != q):
            raise HostError("locate: an entry is not in the pattern")
        return pos
    rp = np.ascontiguousarray(M.indptr, dtype=np.int64)
    ci = _i32(M.indices)
    pos = np.empty(rows.size, dtype=np.int64)
    L = library()
    _chk(L.pcdh_locate(rows.size, _p(rows, _I64P), _p(cols, _I64P),
                       M.shape[0], _p(rp, _I64P), _p(ci, _I32P),
                       _p(pos, _I64P)))
    return pos


def locate_blocks(M, rows, cols, d, pairs, is_u):
    """``pos[p, q]`` = position in ``M.data`` of the entry ``(is_u[d rows[q] +
    i], is_u[d cols[q] + j])`` for every component pair ``(i, j) = pairs[p]``
    (``pcdh_locate_blocks``; the numpy route: :func:`locate` per pair)."""
    import scipy.sparse as sp
    M = sp.csr_matrix(M)
    is_u = np.ascontiguousarray(is_u, dtype=np.int64)
    pairs = [(int(i), int(j)) for i, j in pairs]
    if use_numpy():
        r64 = np.asarray(rows, dtype=np.int64)
        c64 = np.asarray(cols, dtype=np.int64)
        return np.stack([locate(M, is_u[d * r64 + i], is_u[d * c64 + j])
                         for i, j in pairs])
    if not M.has_sorted_indices:
        raise ValueError("locate_blocks: the matrix's indices must be sorted")
    rows, cols = _i32(rows), _i32(cols)
    ci = np.array([p[0] for p in pairs], dtype=np.int32)
    cj = np.array([p[1] for p in pairs], dtype=np.int32)
    rp = np.ascontiguousarray(M.indptr, dtype=np.int64)
    col = _i32(M.indices)
    pos = np.empty((len(pairs), rows.size), dtype=np.int64)
    _chk(library().pcdh_locate_blocks(
        rows.size, _p(rows, _I32P), _p(cols, _I32P), int(d), len(pairs),
        _p(ci, _I32P), _p(cj, _I32P), is_u.size, _p(is_u, _I64P), M.shape[0],
        _p(rp, _I64P), _p(col, _I32P), _p(pos, _I64P)))
    return pos


def contribution_src(order, nloc2, ncells):
    """``(order % nloc2) * ncells + order // nloc2`` as int32, on threads."""
    order = np.ascontiguousarray(order, dtype=np.int64)
    if use_numpy():
        cell, ab = np.divmod(order, nloc2)
        src = ab * ncells + cell
        assert src.size == 0 or src.max() < 2 ** 31
        return src.astype(np.int32)
    src = np.empty(order.size, dtype=np.int32)
    _chk(library().pcdh_contribution_src(order.size, _p(order, _I64P),
                                         int(nloc2), int(ncells),
                                         _p(src, _I32P)))
    return src


def product_pattern(a_indptr, a_indices, b_indptr, b_indices, b_cols):
    """STRUCTURAL pattern of ``A @ B`` from the two patterns alone (the
    symbolic phase of the device's numeric product, ``pcd_fe_set_level_product``):
    ``(indptr, indices)`` as int32, columns sorted in every row."""
    arp, ac = _i32(a_indptr), _i32(a_indices)
    brp, bc = _i32(b_indptr), _i32(b_indices)
    n = arp.size - 1
    L = library()
    crp = np.empty(n + 1, dtype=np.int64)
    _chk(L.pcdh_spgemm_count(0, n, int(b_cols), _p(arp, _I32P), _p(ac, _I32P),
                             _p(brp, _I32P), _p(bc, _I32P), _p(crp, _I64P)))
    nnz = int(crp[-1])
    if nnz >= 2 ** 31:
        raise ValueError("product_pattern: more than 2^31 entries")
    cc = np.empty(nnz, dtype=np.int32)
    cv = np.empty(nnz, dtype=np.float64)          # (values of ones: dropped)
    _chk(L.pcdh_spgemm_fill(0, n, int(b_cols), _p(arp, _I32P), _p(ac, _I32P),
                            None, _p(brp, _I32P), _p(bc, _I32P), None,
                            _p(crp, _I64P), _p(cc, _I32P), _p(cv, _F64P)))
    return crp.astype(np.int32), cc


def product_plan(A, B, mode):
    """Gather plan of ``C = A @ B`` on its structural pattern
    (``pcdh_product_plan_*``): returns ``(c_indptr, c_indices, ptr, src, w)`` -
    entry ``e`` of ``C`` (CSR order) is ``sum(w[t] * X[src[t]])`` over
    ``t in ptr[e]:ptr[e+1]`` with ``X`` = the values of ``A`` (``mode`` 0, the
    weights are ``B``'s) or of ``B`` (``mode`` 1, the weights are ``A``'s)."""
    import scipy.sparse as sp
    A, B = sp.csr_matrix(A), sp.csr_matrix(B)
    assert A.shape[1] == B.shape[0]
    for M in (A, B):
        if not M.has_sorted_indices:
            M.sort_indices()
    arp, ac, av = _i32(A.indptr), _i32(A.indices), \
        np.ascontiguousarray(A.data, dtype=np.float64)
    brp, bc, bv = _i32(B.indptr), _i32(B.indices), \
        np.ascontiguousarray(B.data, dtype=np.float64)
    L = library()
    n = A.shape[0]
    crp = np.empty(n + 1, dtype=np.int64)
    trp = np.empty(n + 1, dtype=np.int64)
    _chk(L.pcdh_product_plan_count(n, B.shape[1], _p(arp, _I32P), _p(ac, _I32P),
                                   _p(brp, _I32P), _p(bc, _I32P),
                                   _p(crp, _I64P), _p(trp, _I64P)))
    nnz, nt = int(crp[-1]), int(trp[-1])
    if max(nnz, nt, A.nnz, B.nnz) >= 2 ** 31:
        raise ValueError("product_plan: more than 2^31 entries / terms")
    cc = np.empty(nnz, dtype=np.int32)
    ptr = np.empty(nnz + 1, dtype=np.int64)
    src = np.empty(nt, dtype=np.int32)
    w = np.empty(nt, dtype=np.float64)
    _chk(L.pcdh_product_plan_fill(n, B.shape[1], _p(arp, _I32P), _p(ac, _I32P),
                                  _p(av, _F64P), _p(brp, _I32P), _p(bc, _I32P),
                                  _p(bv, _F64P), int(mode), _p(crp, _I64P),
                                  _p(trp, _I64P), _p(cc, _I32P), _p(ptr, _I64P),
                                  _p(src, _I32P), _p(w, _F64P)))
    return crp, cc, ptr, src, w


def transpose(A):
    """CSR transpose with sorted columns (counting sort by column)."""
    import scipy.sparse as sp
    A = sp.csr_matrix(A)
    nr, nc = A.shape
    rp, ci = _i32(A.indptr), _i32(A.indices)
    va = np.ascontiguousarray(A.data, dtype=np.float64)
    trp = np.empty(nc + 1, dtype=np.int32)
    tc = np.empty(A.nnz, dtype=np.int32)
    tv = np.empty(A.nnz, dtype=np.float64)
    _chk(library().pcdh_transpose(nr, nc, _p(rp, _I32P), _p(ci, _I32P),
                                  _p(va, _F64P), _p(trp, _I32P),
                                  _p(tc, _I32P), _p(tv, _F64P)))
    T = sp.csr_matrix((tv, tc, trp), shape=(nc, nr))
    T.has_sorted_indices = True
    return T


def union_blocks(n, blocks):
    """Pattern of the union of index-mapped CSR blocks.  ``blocks``: list of
    ``(rowmap, colmap, indptr, indices)``; returns ``(indptr, indices,
    order)`` with ``order[k]`` = position of entry ``k`` in the concatenation
    of the blocks' value arrays."""
    nb = len(blocks)
    keep = [[_i32(a) for a in blk] for blk in blocks]

    def ptrs(j):
        arr = (_I32P * nb)()
        for b in range(nb):
            arr[b] = _p(keep[b][j], _I32P)
        return arr
    nr = np.array([blk[0].size for blk in keep], dtype=np.int64)
    off = np.zeros(nb, dtype=np.int64)
    off[1:] = np.cumsum([blk[3].size for blk in keep])[:-1]
    L = library()
    indptr = np.empty(n + 1, dtype=np.int64)
    rm, cm, ip, ix = ptrs(0), ptrs(1), ptrs(2), ptrs(3)
    _chk(L.pcdh_union_count(n, nb, _p(nr, _I64P), rm, ip, _p(indptr, _I64P)))
    nnz = int(indptr[-1])
    indices = np.empty(nnz, dtype=np.int32)
    order = np.empty(nnz, dtype=np.int64)
    _chk(L.pcdh_union_fill(n, nb, _p(nr, _I64P), rm, cm, ip, ix,
                           _p(off, _I64P), _p(indptr, _I64P),
                           _p(indices, _I32P), _p(order, _I64P)))
    return indptr, indices, order


class SpMV(object):
    """``y = scale .* (A x)`` with the arrays of ``A`` converted once (threaded;
    bitwise scipy's ``csr_matvec`` row sums).  ``nvec`` > 1: ``A`` is the
    scalar factor of ``A (x) I_nvec`` and ``x`` / ``scale`` / ``y`` are the
    expanded operator's (node-interleaved) vectors."""

    def __init__(self, A, scale=None, nvec=1):
        import scipy.sparse as sp
        A = sp.csr_matrix(A)
        self.n, self.nvec = A.shape[0], int(nvec)
        self._rp, self._ci = _i32(A.indptr), _i32(A.indices)
        self._va = np.ascontiguousarray(A.data, dtype=np.float64)
        self._sc = None if scale is None else \
            np.ascontiguousarray(scale, dtype=np.float64)

    def __call__(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.n * self.nvec)
        if self.nvec == 1:
            _chk(library().pcdh_spmv(self.n, _p(self._rp, _I32P),
                                     _p(self._ci, _I32P), _p(self._va, _F64P),
                                     _p(x, _F64P), _p(self._sc, _F64P),
                                     _p(y, _F64P)))
        else:
            _chk(library().pcdh_spmm(self.n, _p(self._rp, _I32P),
                                     _p(self._ci, _I32P), _p(self._va, _F64P),
                                     self.nvec, _p(x, _F64P),
                                     _p(self._sc, _F64P), _p(y, _F64P)))
        return y


NOT_KRON = 100


def kron_factor(A, nc):
    """``F`` (scipy CSR) when ``A == F (x) I_nc`` exactly on node-interleaved
    dofs - pattern AND values - else ``None``."""
    import scipy.sparse as sp
    A = sp.csr_matrix(A)
    n, m = A.shape
    if nc < 2 or n % nc or m % nc or A.nnz % nc:
        return None
    if use_numpy():
        F = sp.csr_matrix(A[::nc, ::nc])
        E = sp.kron(F, sp.identity(nc), format="csr")
        if E.nnz != A.nnz or abs(E - A).max() != 0:
            return None
        F.sort_indices()
        return F
    if not A.has_sorted_indices:
        A = A.sorted_indices()
    rp, ci = _i32(A.indptr), _i32(A.indices)
    va = np.ascontiguousarray(A.data, dtype=np.float64)
    frp = np.empty(n // nc + 1, dtype=np.int32)
    fc = np.empty(A.nnz // nc, dtype=np.int32)
    fv = np.empty(A.nnz // nc, dtype=np.float64)
    rc = library().pcdh_kron_factor(n, _p(rp, _I32P), _p(ci, _I32P),
                                    _p(va, _F64P), nc, _p(frp, _I32P),
                                    _p(fc, _I32P), _p(fv, _F64P))
    if rc == NOT_KRON:
        return None
    _chk(rc)
    F = sp.csr_matrix((fv, fc, frp), shape=(n // nc, m // nc))
    F.has_sorted_indices = True
    return F


def kron_expand(F, nc):
    """``F (x) I_nc`` (node-interleaved, sorted columns); the result remembers
    its factor (``.kron_scalar``, ``.kron_block``)."""
    import scipy.sparse as sp
    F = sp.csr_matrix(F)
    if nc == 1:
        return F
    if use_numpy():
        A = sp.kron(F, sp.identity(nc, format="csr"), format="csr")
        A.sort_indices()
    else:
        if not F.has_sorted_indices:
            F = F.sorted_indices()
        ns, ms = F.shape
        frp, fc = _i32(F.indptr), _i32(F.indices)
        fv = np.ascontiguousarray(F.data, dtype=np.float64)
        rp = np.empty(nc * ns + 1, dtype=np.int32)
        ci = np.empty(nc * F.nnz, dtype=np.int32)
        va = np.empty(nc * F.nnz, dtype=np.float64)
        _chk(library().pcdh_kron_expand(ns, _p(frp, _I32P), _p(fc, _I32P),
                                        _p(fv, _F64P), nc, _p(rp, _I32P),
                                        _p(ci, _I32P), _p(va, _F64P)))
        A = sp.csr_matrix((va, ci, rp), shape=(nc * ns, nc * ms))
        A.has_sorted_indices = True
    A.kron_scalar, A.kron_block = F, nc
    return A


def wind_gradlam(cell_dofs, U, gradlam, area):
    """``out[c, m, k] = area[c] * sum_d U[cell_dofs[c, m], d] * gradlam[c, k, d]``
    (threaded, one pass; bitwise the elementwise numpy chain)."""
    dofs = _i64(cell_dofs)
    U = np.ascontiguousarray(U, dtype=np.float64)
    g = np.ascontiguousarray(gradlam, dtype=np.float64)
    a = np.ascontiguousarray(area, dtype=np.float64)
    nc, na = dofs.shape
    nvl, dim = g.shape[1], g.shape[2]
    out = np.empty((nc, na, nvl))
    _chk(library().pcdh_wind_gradlam(nc, na, nvl, dim, _p(dofs, _I64P),
                                     _p(U, _F64P), _p(g, _F64P), _p(a, _F64P),
                                     _p(out, _F64P)))
    return out


def gather_sum(ptr, members, vals, out=None):
    """``out[g] = sum(vals[members[ptr[g]:ptr[g+1]]])``, added in ascending
    position (``numpy.bincount(inv, weights=vals)`` for the grouping whose
    member lists these are)."""
    ptr, members = _i64(ptr), _i64(members)
    vals = np.ascontiguousarray(vals, dtype=np.float64).ravel()
    ng = ptr.size - 1
    if out is None:
        out = np.empty(ng)
    _chk(library().pcdh_gather_sum(ng, _p(ptr, _I64P), _p(members, _I64P),
                                   _p(vals, _F64P), _p(out, _F64P)))
    return out


def mis2_degrees(S):
    """Number of vertices within two edges of every vertex of the graph ``S``
    (CSR pattern, symmetric, no diagonal) - the row lengths of the
    off-diagonal pattern of ``(S + I)^2`` without forming it."""
    import scipy.sparse as sp
    S = sp.csr_matrix(S)
    n = S.shape[0]
    deg = np.zeros(n, dtype=np.int64)
    ip, ix = _i32(S.indptr), _i32(S.indices)
    _chk(library().pcdh_mis2_degrees(n, _p(ip, _I32P), _p(ix, _I32P),
                                     _p(deg, _I64P)))
    return deg


def mis2(S, w):
    """Maximal independent set of the DISTANCE-2 graph of ``S`` by Luby's
    rounds with the priorities ``w`` (``pcdh_mis2``): boolean mask."""
    import scipy.sparse as sp
    S = sp.csr_matrix(S)
    n = S.shape[0]
    out = np.zeros(n, dtype=np.int8)
    ip, ix = _i32(S.indptr), _i32(S.indices)
    w = np.ascontiguousarray(w, dtype=np.float64)
    _chk(library().pcdh_mis2(n, _p(ip, _I32P), _p(ix, _I32P), _p(w, _F64P),
                             out.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)),
                             None))
    return out.astype(bool)
