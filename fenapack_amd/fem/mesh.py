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

    Attributes
    ----------
    vertices : (nv, 2) float64
    cells : (nc, 3) int64 vertex ids, counter-clockwise after construction
    edges : (ne, 2) int64 sorted vertex pairs
    cell_edges : (nc, 3) int64; local edge k is *opposite* local vertex k
    boundary_edges : (nbe,) int64 ids of edges with a single adjacent cell
    """

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
        key = pairs[:, 0] * nv + pairs[:, 1]
        ukey, inv, counts = np.unique(key, return_inverse=True,
                                      return_counts=True)
        self.edges = np.stack([ukey // nv, ukey % nv], axis=1)
        self.cell_edges = inv.reshape(-1, 3)
        self.boundary_edges = np.nonzero(counts == 1)[0]

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
