"""Triangular meshes for the PCD benchmark problems.

Two geometries are carried:

* the reference's backward-facing step on an L-shaped domain: the 20-vertex /
  22-cell coarse mesh of ``demo/data/mesh_lshape.xml:5-51`` refined uniformly
  ``level`` times (``demo/navier-stokes-pcd/demo_navier-stokes-pcd.py:50-53``);
* the lid-driven unit-square cavity BASELINE.json names (N = 5 * 2**level
  cells per side, "right" diagonals, like DOLFIN's ``UnitSquareMesh``).

The coarse L-shape coordinates/connectivity below are mesh *data* restated
from the reference's XML fixture, not code.
"""

import numpy as np

# demo/data/mesh_lshape.xml:6-25 (vertices) and :28-49 (cells)
_LSHAPE_VERTICES = np.array([
    [-1.0, 0.0], [-1.0, 1.0], [0.0, 0.0], [0.0, 1.0], [1.0, 0.0],
    [1.0, 1.0], [0.0, -1.0], [1.0, -1.0], [2.0, 0.0], [2.0, 1.0],
    [3.0, 0.0], [3.0, 1.0], [4.0, 0.0], [4.0, 1.0], [5.0, 0.0],
    [5.0, 1.0], [2.0, -1.0], [3.0, -1.0], [4.0, -1.0], [5.0, -1.0]],
    dtype=np.float64)

_LSHAPE_CELLS = np.array([
    [0, 2, 1], [3, 1, 2], [2, 4, 3], [5, 3, 4], [4, 2, 7], [6, 7, 2],
    [4, 8, 5], [9, 5, 8], [8, 10, 9], [11, 9, 10], [10, 12, 11],
    [13, 11, 12], [12, 14, 13], [15, 13, 14], [7, 16, 4], [8, 4, 16],
    [16, 17, 8], [10, 8, 17], [17, 18, 10], [12, 10, 18], [18, 19, 12],
    [14, 12, 19]], dtype=np.int64)


class Mesh(object):
    """Conforming triangle mesh with unique edges.
    ``dim`` / ``local_edges`` give the dimension-independent view the P2/P1
    producer uses (``TetMesh`` is the 3D counterpart).

    Attributes
    ----------
    vertices : (nv, 2) float64
    cells : (nc, 3) int64 vertex ids, counter-clockwise after construction
    edges : (ne, 2) int64 sorted vertex pairs
    cell_edges : (nc, 3) int64; local edge k is *opposite* local vertex k
    boundary_edges : (nbe,) int64 ids of edges with a single adjacent cell
    """

    dim = 2
    local_edges = ((1, 2), (2, 0), (0, 1))      # edge k opposite vertex k

    def __init__(self, vertices, cells):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64)
        cells = np.ascontiguousarray(cells, dtype=np.int64)
        # enforce counter-clockwise orientation
        p = self.vertices[cells]
        det = ((p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1])
               - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0]))
        flip = det < 0
        cells[flip] = cells[flip][:, [0, 2, 1]]
        self.cells = cells
        self._build_edges()

    def _build_edges(self):
        c = self.cells
        # local edge k opposite to local vertex k
        pairs = np.stack([c[:, [1, 2]], c[:, [2, 0]], c[:, [0, 1]]], axis=1)
        pairs = np.sort(pairs.reshape(-1, 2), axis=1)
        nv = self.num_vertices
        from .. import _host
        if _host.use_numpy() or pairs.shape[0] < 50000:
            key = pairs[:, 0] * nv + pairs[:, 1]
            ukey, inv, counts = np.unique(key, return_inverse=True,
                                          return_counts=True)
            self.edges = np.stack([ukey // nv, ukey % nv], axis=1)
        else:
            # the same ordering (by first, then second vertex) from the native
            # counting sort instead of a comparison sort of the keys
            g = _host.group_pairs(pairs[:, 0], pairs[:, 1], nv)
            inv = g.inv
            counts = np.diff(g.members()[0])
            first = np.repeat(np.arange(nv, dtype=np.int64),
                              np.diff(g.indptr))
            self.edges = np.stack([first, g.ucols], axis=1)
            g.release()
        self.cell_edges = inv.reshape(-1, 3)
        self.boundary_edges = np.nonzero(counts == 1)[0]
        self.boundary_vertices = np.unique(self.edges[self.boundary_edges])

    @property
    def num_vertices(self):
        return self.vertices.shape[0]

    @property
    def num_cells(self):
        return self.cells.shape[0]

    @property
    def num_edges(self):
        return self.edges.shape[0]

    def edge_midpoints(self):
        return 0.5 * (self.vertices[self.edges[:, 0]]
                      + self.vertices[self.edges[:, 1]])

    def refine(self):
        """Uniform (red) refinement: every triangle into four."""
        nv = self.num_vertices
        mid = nv + self.cell_edges            # new vertex ids per local edge
        v = self.cells
        m0, m1, m2 = mid[:, 0], mid[:, 1], mid[:, 2]   # opposite v0, v1, v2
        children = np.concatenate([
            np.stack([v[:, 0], m2, m1], axis=1),
            np.stack([v[:, 1], m0, m2], axis=1),
            np.stack([v[:, 2], m1, m0], axis=1),
            np.stack([m0, m1, m2], axis=1)], axis=0)
        verts = np.concatenate([self.vertices, self.edge_midpoints()], axis=0)
        return Mesh(verts, children)

    def hmin(self):
        e = self.vertices[self.edges[:, 0]] - self.vertices[self.edges[:, 1]]
        return float(np.sqrt((e * e).sum(axis=1)).min())


def lshape_mesh(level):
    """Reference geometry: L-shaped step refined ``level`` times."""
    mesh = Mesh(_LSHAPE_VERTICES.copy(), _LSHAPE_CELLS.copy())
    for _ in range(level):
        mesh = mesh.refine()
    return mesh


def unit_square_mesh(n):
    """``n x n`` squares, each cut by its "right" diagonal (2 n^2 cells)."""
    xs = np.linspace(0.0, 1.0, n + 1)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    verts = np.stack([X.ravel(), Y.ravel()], axis=1)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="xy")
    v00 = (j * (n + 1) + i).ravel()
    v10 = v00 + 1
    v01 = v00 + (n + 1)
    v11 = v01 + 1
    cells = np.concatenate([np.stack([v00, v10, v11], axis=1),
                            np.stack([v00, v11, v01], axis=1)], axis=0)
    return Mesh(verts, cells)


def cavity_mesh(level):
    """Unit-square cavity with N = 5 * 2**level cells per side:
    the 5 x 5 base mesh refined ``level`` times (a uniformly refined
    right-diagonal mesh is the right-diagonal mesh of twice the resolution)."""
    mesh = unit_square_mesh(5)
    for _ in range(level):
        mesh = mesh.refine()
    return mesh


# --------------------------------------------------------------------- 3D
_TET_EDGES = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))
_KUHN = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))


class TetMesh(object):
    """Conforming tetrahedral mesh with unique edges.  ``cell_edges[:, k]`` is
    the edge joining the local vertices ``_TET_EDGES[k]``.  Boundary vertices
    and edges come from the faces that belong to a single cell."""

    dim = 3
    local_edges = _TET_EDGES

    def __init__(self, vertices, cells):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64)
        cells = np.ascontiguousarray(cells, dtype=np.int64)
        p = self.vertices[cells]
        T = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0],
                      p[:, 3] - p[:, 0]], axis=2)
        flip = np.linalg.det(T) < 0
        cells[flip] = cells[flip][:, [0, 2, 1, 3]]
        self.cells = cells
        nv = self.num_vertices
        pairs = np.stack([cells[:, list(e)] for e in _TET_EDGES], axis=1)
        pairs = np.sort(pairs.reshape(-1, 2), axis=1)
        from .. import _host
        native = not _host.use_numpy() and pairs.shape[0] >= 50000
        if native:
            # (counting sorts instead of comparison sorts of the keys; the
            # orderings - by first, then second vertex - are numpy.unique's)
            g = _host.group_pairs(pairs[:, 0], pairs[:, 1], nv)
            inv = g.inv
            e0 = np.repeat(np.arange(nv, dtype=np.int64), np.diff(g.indptr))
            ukey = e0 * nv + g.ucols
            self.edges = np.stack([e0, g.ucols], axis=1)
            g.release()
        else:
            key = pairs[:, 0] * nv + pairs[:, 1]
            ukey, inv = np.unique(key, return_inverse=True)
            self.edges = np.stack([ukey // nv, ukey % nv], axis=1)
        self.cell_edges = inv.reshape(-1, 6)
        faces = np.stack([cells[:, [1, 2, 3]], cells[:, [0, 2, 3]],
                          cells[:, [0, 1, 3]], cells[:, [0, 1, 2]]], axis=1)
        faces = np.sort(faces.reshape(-1, 3), axis=1)
        if native:
            g = _host.group_pairs(faces[:, 0], faces[:, 1] * nv + faces[:, 2],
                                  nv)
            ptr, order = g.members()
            first, cnt = order[ptr[:-1]], np.diff(ptr)
            g.release()
        else:
            fkey = (faces[:, 0] * nv + faces[:, 1]) * nv + faces[:, 2]
            uf, first, cnt = np.unique(fkey, return_index=True,
                                       return_counts=True)
        bfirst = np.sort(first[cnt == 1])
        bfaces = faces[bfirst]
        # boundary facets (sorted vertex triples), the cell each belongs to and
        # that cell's vertex opposite the facet (it lies inside: outward
        # normals of the boundary integrals, TaylorHood.robin_plan)
        self.boundary_faces = bfaces
        self.boundary_face_cells = bfirst // 4
        self.boundary_face_opposite = cells[bfirst // 4, bfirst % 4]
        self.boundary_vertices = np.unique(bfaces)
        bpairs = np.sort(np.concatenate([bfaces[:, [0, 1]], bfaces[:, [0, 2]],
                                         bfaces[:, [1, 2]]]), axis=1)
        bkey = np.unique(bpairs[:, 0] * nv + bpairs[:, 1])
        self.boundary_edges = np.searchsorted(ukey, bkey)

    @property
    def num_vertices(self):
        return self.vertices.shape[0]

    @property
    def num_cells(self):
        return self.cells.shape[0]

    @property
    def num_edges(self):
        return self.edges.shape[0]

    def edge_midpoints(self):
        return 0.5 * (self.vertices[self.edges[:, 0]]
                      + self.vertices[self.edges[:, 1]])

    def edge_index(self, a, b):
        """Index in ``edges`` of the edges joining vertices ``a`` and ``b``
        (arrays; ``edges`` is sorted by (first, second) vertex)."""
        nv = self.num_vertices
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        key = self.edges[:, 0] * nv + self.edges[:, 1]
        idx = np.searchsorted(key, lo * nv + hi)
        assert np.array_equal(key[idx], lo * nv + hi), "not edges of this mesh"
        return idx

    def boundary_face_centroids(self):
        return self.vertices[self.boundary_faces].mean(axis=1)


def unit_cube_mesh(n):
    """``n^3`` cubes, each cut into the 6 Kuhn tetrahedra (all share the main
    diagonal); the mesh of ``2n`` is the uniform refinement of the mesh of
    ``n``, which is what the 3D multigrid hierarchy relies on."""
    xs = np.linspace(0.0, 1.0, n + 1)
    Z, Y, X = np.meshgrid(xs, xs, xs, indexing="ij")
    verts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n),
                          indexing="ij")
    base = ((k * (n + 1) + j) * (n + 1) + i).ravel()
    step = np.array([1, n + 1, (n + 1) ** 2])
    cells = []
    for perm in _KUHN:
        v0 = base
        v1 = v0 + step[perm[0]]
        v2 = v1 + step[perm[1]]
        v3 = v2 + step[perm[2]]
        cells.append(np.stack([v0, v1, v2, v3], axis=1))
    return TetMesh(verts, np.concatenate(cells, axis=0))


def kuhn_parents(coarse_n, fine_mesh):
    """Coarse Kuhn tetrahedron (index into ``unit_cube_mesh(coarse_n)``'s
    cell list) containing each cell of the twice finer mesh."""
    c = fine_mesh.vertices[fine_mesh.cells].mean(axis=1) * coarse_n
    ijk = np.minimum(np.floor(c).astype(np.int64), coarse_n - 1)
    loc = c - ijk
    # the Kuhn tet of permutation pi is {x_pi0 >= x_pi1 >= x_pi2}
    order = np.argsort(-loc, axis=1, kind="stable")
    perm_id = {p: q for q, p in enumerate(_KUHN)}
    pid = np.array([perm_id[tuple(o)] for o in order])
    cube = (ijk[:, 2] * coarse_n + ijk[:, 1]) * coarse_n + ijk[:, 0]
    return pid * coarse_n ** 3 + cube
