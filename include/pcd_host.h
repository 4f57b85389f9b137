/* pcd_host.h - C ABI of libpcd_host.so: the host-side (OpenMP) set-up helpers
 * of the MI355X PCD engine.
 *
 * Everything here is INTEGER work of problem set-up: sparsity patterns,
 * element -> entry contribution lists, prolongation patterns, Galerkin
 * gather plans, the fieldsplit sub-matrix extraction.  The floating-point
 * work of set-up runs on the device (pcd_fe_* in pcd_engine.h) or as GEMMs
 * over reference tensors in the Python producer; the hot path never comes
 * here.  No HIP dependency: the library loads and is tested on a box without
 * a GPU.
 *
 * What it replaces in the reference: the reference delegates set-up to
 * DOLFIN's SystemAssembler / PETSc's MatCreateSubMatrix / MatPtAP
 * (fenapack/field_split_backend.py:230-263, 285-291, 311-342;
 * fenapack/assembling.py:151-180); on the partitioned path every rank only
 * ever touches its owned rows (fenapack/SubfieldBC.h:136-155) - the
 * `row0/row1` arguments below are that restriction.
 *
 * All functions return 0 on success, a PCDH_ERR_* code otherwise;
 * pcdh_last_error() gives the message.  Nothing throws across the ABI.
 */
#ifndef PCD_HOST_H
#define PCD_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCDH_ERR_ARG 1
#define PCDH_ERR_NOMEM 2
#define PCDH_ERR_STATE 3
#define PCDH_NOT_KRON 100   /* pcdh_kron_factor: a valid answer, not an error */

const char* pcdh_last_error(void);
/* OpenMP threads the helpers use (0: the runtime's default) */
int pcdh_set_threads(int nthreads);
int pcdh_get_threads(void);

/* ---- grouping of (row, col) pairs -------------------------------------------
 * The one primitive behind every pattern of the producer: n pairs with
 * 0 <= rows[i] < nrows and cols[i] >= 0 are grouped by (row, col), groups
 * ordered by row then col, members of a group in ascending input position
 * (the order numpy.bincount adds element contributions on the host and the
 * device gather kernels add them in HBM: bitwise reproducible assembly).
 * Only rows in [row0, row1) are kept (rank-local set-up: owned rows only);
 * pairs of other rows get inv = -1.
 *
 * Replaces numpy.unique(keys, return_inverse=True) + stable argsort in the
 * Python producer (fem/taylor_hood.py FixedPattern, device_producer._group /
 * _contribution_plan, fem/multigrid._unique_entries, fem/mesh edge tables).
 */
typedef struct pcdh_group_s* pcdh_group;

int pcdh_group_pairs(int64_t n, const int64_t* rows, const int64_t* cols,
                     int64_t nrows, int64_t row0, int64_t row1,
                     pcdh_group* out);
/* number of distinct (row, col) groups / of kept input pairs */
int64_t pcdh_group_nnz(pcdh_group g);
int64_t pcdh_group_kept(pcdh_group g);
/* Any output may be NULL.  indptr[row1 - row0 + 1]: CSR row pointer of the
 * kept rows; ucols[nnz]: column of every group; inv[n]: group of every input
 * pair (-1: row outside [row0, row1)); ptr[nnz + 1], order[kept]: members of
 * group k are order[ptr[k] .. ptr[k+1]) in ascending input position. */
int pcdh_group_export(pcdh_group g, int64_t* indptr, int64_t* ucols,
                      int64_t* inv, int64_t* ptr, int64_t* order);
void pcdh_group_free(pcdh_group g);

/* ---- element -> entry maps from cell dof tables ------------------------------
 * Pattern of a bilinear form on a mesh: cell c couples its `nr` row dofs
 * rdofs[c*nr + a] with its `nc` column dofs cdofs[c*nc + b].  Same outputs as
 * pcdh_group_pairs on the ncell*nr*nc pairs (a-major, b-minor within a cell),
 * without materialising them.
 */
int pcdh_pattern_cells(int64_t ncell, int nr, const int64_t* rdofs, int nc,
                       const int64_t* cdofs, int64_t nrows, int64_t row0,
                       int64_t row1, pcdh_group* out);

/* ---- fieldsplit sub-matrix extraction -----------------------------------------
 * [ext PETSc] MatCreateSubMatrix(A, isrow, iscol) with value provenance, the
 * host half of pcd_set_system (field_split_backend.py:311-342): rows
 * `rows[0..nr)` of the CSR (rowptr, col), columns renumbered through
 * colmap[] (-1: dropped), entries of a row sorted by the new column;
 * src[k] = position of entry k in the caller's value array.
 * Two calls: counts (orp) first, then the fill.
 */
int pcdh_extract_count(int64_t nr, const int32_t* rows, const int32_t* rowptr,
                       const int32_t* col, const int32_t* colmap,
                       int32_t* out_rowptr /* nr + 1 */);
int pcdh_extract_fill(int64_t nr, const int32_t* rows, const int32_t* rowptr,
                      const int32_t* col, const int32_t* colmap,
                      const int32_t* out_rowptr, int32_t* out_col,
                      int64_t* out_src);

/* ---- CSR transpose (restriction = prolongation^T) ---------------------------- */
int pcdh_transpose(int64_t nr, int64_t nc, const int32_t* rowptr,
                   const int32_t* col, const double* val, int32_t* t_rowptr,
                   int32_t* t_col, double* t_val);

/* ---- fixed-pattern sparse product C = A B ------------------------------------
 * Symbolic + numeric row-wise SpGEMM (Gustavson, one dense marker per
 * thread), columns of every output row sorted.  Two calls: counts, then fill.
 * The Galerkin operators P^T (A P) of the multigrid hierarchies
 * ([ext PETSc] MatPtAP; fem/multigrid.galerkin_chain).  Rows [row0, row1) of A
 * only (rank-local set-up).  Entries that cancel to an exact zero are KEPT:
 * the pattern is structural, so later value refreshes fit it.
 */
int pcdh_spgemm_count(int64_t row0, int64_t row1, int64_t b_cols,
                      const int32_t* a_rowptr, const int32_t* a_col,
                      const int32_t* b_rowptr, const int32_t* b_col,
                      int64_t* c_rowptr /* row1 - row0 + 1 */);
int pcdh_spgemm_fill(int64_t row0, int64_t row1, int64_t b_cols,
                     const int32_t* a_rowptr, const int32_t* a_col,
                     const double* a_val, const int32_t* b_rowptr,
                     const int32_t* b_col, const double* b_val,
                     const int64_t* c_rowptr, int32_t* c_col, double* c_val);

/* ---- gather plan of a sparse product ------------------------------------------
 * C = A B on a fixed pattern as a plan: for every entry of C (rows of A in
 * order, columns ascending; c_rowptr / c_col) the terms that sum to it
 * (ptr[e] .. ptr[e + 1] in src / w), in the order of A's row.  mode 0: src =
 * index of the A entry, w = the B value (B = F P: F changes, P is fixed);
 * mode 1: src = index of the B entry, w = the A value (F_c = P^T B).  The
 * device-side Galerkin refresh of the multigrid hierarchies
 * (pcd_fe_set_level_galerkin; what hypre's set-up redoes per outer iteration in
 * demo_navier-stokes-pcd.py:153-160).  Two calls: counts (entries and terms per
 * row, as row pointers), then fill.
 */
int pcdh_product_plan_count(int64_t nrows, int64_t b_cols, const int32_t* a_rowptr,
                            const int32_t* a_col, const int32_t* b_rowptr,
                            const int32_t* b_col, int64_t* c_rowptr /* nrows + 1 */,
                            int64_t* t_rowptr /* nrows + 1 */);
int pcdh_product_plan_fill(int64_t nrows, int64_t b_cols, const int32_t* a_rowptr,
                           const int32_t* a_col, const double* a_val,
                           const int32_t* b_rowptr, const int32_t* b_col,
                           const double* b_val, int mode, const int64_t* c_rowptr,
                           const int64_t* t_rowptr, int32_t* c_col,
                           int64_t* ptr /* nnz(C) + 1 */, int32_t* src, double* w);

/* ---- y = scale .* (A x) ------------------------------------------------------
 * Threaded CSR SpMV (row sums in ascending entry order: bitwise what scipy's
 * csr_matvec gives).  The host-side power iterations behind
 * -ksp_chebyshev_esteig / the smoother bounds ([ext PETSc] KSPChebyshevEstEig)
 * are a dozen of these per multigrid level; `scale` (may be NULL) is the
 * Jacobi diagonal D^-1. */
int pcdh_spmv(int64_t nrows, const int32_t* rowptr, const int32_t* col,
              const double* val, const double* x, const double* scale, double* y);

/* The same for `nvec` (<= 8) interleaved vectors: X, Y are nrows x nvec row-major,
 * scale has nrows * nvec entries.  With the scalar factor F of F (x) I_nvec this
 * IS the expanded operator's SpMV (same sums, same order) at half / a third of
 * the matrix traffic. */
int pcdh_spmm(int64_t nrows, const int32_t* rowptr, const int32_t* col,
              const double* val, int nvec, const double* x, const double* scale,
              double* y);

/* ---- F (x) I_nc <-> F ---------------------------------------------------------
 * The velocity block of the preconditioner matrix is F (x) I_d on node-
 * interleaved dofs (the Picard operator of demo_navier-stokes-pcd.py:101-106
 * couples no components), and so are its prolongations: Galerkin products,
 * smoother bounds and composed levels are computed on the scalar factor and
 * expanded for the hand-over.
 * pcdh_kron_factor: 0 and F (f_rowptr: nrows / nc + 1, f_col / f_val: nnz / nc;
 * both may be NULL = test only) when A == F (x) I_nc EXACTLY (pattern and
 * values), PCDH_NOT_KRON when it is not.
 * pcdh_kron_expand: rowptr nc * ns + 1, col / val nc * nnz(F); sorted columns.
 */
int pcdh_kron_factor(int64_t nrows, const int32_t* rowptr, const int32_t* col,
                     const double* val, int nc, int32_t* f_rowptr,
                     int32_t* f_col, double* f_val);
int pcdh_kron_expand(int64_t ns, const int32_t* f_rowptr, const int32_t* f_col,
                     const double* f_val, int nc, int32_t* rowptr, int32_t* col,
                     double* val);

/* ---- out[g] = sum of vals[members[ptr[g] .. ptr[g+1])] ------------------------
 * Element contributions summed per matrix entry (the scatter-add of assembly,
 * assembling.py:151-155, as a threaded gather): additions in ascending
 * position, the order of numpy.bincount and of the device's k_fe_gather. */
int pcdh_gather_sum(int64_t ngroups, const int64_t* ptr, const int64_t* members,
                    const double* vals, double* out);

/* ---- out[c, m, k] = |T_c| sum_d U[dofs[c, m], d] gradlam[c, k, d] ------------
 * The wind-dependent factor of the P2 convection element matrices (the form
 * ((w.grad) u, v) of demo_navier-stokes-pcd.py:106-109 on nodal winds), in one
 * threaded pass; bitwise the elementwise numpy chain it replaces. */
int pcdh_wind_gradlam(int64_t ncell, int na, int nvl, int dim, const int64_t* dofs,
                      const double* U, const double* gradlam, const double* area,
                      double* out);

/* ---- union of index-mapped blocks -------------------------------------------
 * The monolithic pattern of a block system in the caller's mixed numbering
 * (what DOLFIN's SystemAssembler produces directly, assembling.py:151-155):
 * block b contributes its CSR rows i -> global row rowmap[b][i], columns
 * c -> colmap[b][c].  Blocks must not overlap.  Output: CSR with sorted
 * columns and, per entry, its position in the concatenation of the blocks'
 * value arrays (data_off[b] + k): assembling values is one gather.
 * Two calls: counts, then fill.
 */
int pcdh_union_count(int64_t n, int nb, const int64_t* nr,
                     const int32_t* const* rowmap, const int32_t* const* indptr,
                     int64_t* out_indptr /* n + 1 */);
int pcdh_union_fill(int64_t n, int nb, const int64_t* nr,
                    const int32_t* const* rowmap, const int32_t* const* colmap,
                    const int32_t* const* indptr, const int32_t* const* indices,
                    const int64_t* data_off, const int64_t* out_indptr,
                    int32_t* out_indices, int64_t* out_order);

/* ---- distance-2 maximal independent set without the squared graph ------------
 * The roots of the smoothed-aggregation hierarchy behind -pc_type gamg
 * (fenapack_amd/amg.py; the reference leaves its algebraic hierarchy to hypre:
 * demo_navier-stokes-pcd.py:153-160) are a maximal independent set of the
 * DISTANCE-2 graph G2 = off-diagonal pattern of (S + I)^2 of the strength graph
 * S.  For a 3-D P2 stencil G2 has ~170 entries per row (cube N = 73: 5.4e8
 * entries, tens of GB and half a minute of SpGEMM before the first Luby round);
 * both functions below walk two hops of S instead and never store G2.
 * `rowptr` / `col`: pattern of S - symmetric, no diagonal entries.
 * pcdh_mis2_degrees: deg[i] = number of vertices other than i within two edges
 *   of i (= the row lengths of G2: the degree term of Luby's priorities).
 * pcdh_mis2: Luby's rounds on G2 with the priorities `w` - an undecided vertex
 *   joins the set when its priority is STRICTLY greater than that of every
 *   undecided vertex within two edges (the maximum over the two-hop
 *   neighbourhood excluding the vertex itself comes from the two largest
 *   DISTINCT candidates of every closed one-hop neighbourhood), vertices
 *   within two edges of a new member drop out; a round without a winner
 *   (tied priorities) admits the undecided vertex of largest priority, lowest
 *   index first.  in_set[i] = 1 / 0.  Every round is a pure function of the
 *   state before it: the result does not depend on the thread count, and it
 *   is the set the numpy restatement over the explicit G2 finds (amg._mis).
 */
int pcdh_mis2_degrees(int64_t n, const int32_t* rowptr, const int32_t* col,
                      int64_t* deg);
int pcdh_mis2(int64_t n, const int32_t* rowptr, const int32_t* col,
              const double* w, int8_t* in_set, int64_t* rounds /* may be NULL */);

/* ---- gathers of values ------------------------------------------------------
 * out[i] = concat(seg[0 .. nseg-1])[idx[i]], the concatenation never formed:
 * seg_off[s] = first index of segment s (nseg + 1 entries, at most 16
 * segments).  The values of the monolithic system from its blocks, of a
 * sub-matrix from its parent (what createSubMatrix(..., submat=) refreshes:
 * fenapack/field_split_backend.py:331-334), on threads.                      */
int pcdh_take_segments(int64_t n, const int64_t* idx, int nseg,
                       const double* const* seg, const int64_t* seg_off,
                       double* out);

/* ---- positions of entries ---------------------------------------------------
 * pos[q] = position of entry (qrow[q], qcol[q]) in a CSR pattern with sorted
 * columns (threads over the queries, a bisection per query).  The device
 * producer's "where does entry k of the velocity block sit in the system
 * values" (fenapack/field_split_backend.py:331-334 keeps the same map inside
 * PETSc's createSubMatrix) - round 6: 10.6 s of numpy argsort on config 5's
 * 1.1e9-entry system became a fraction of a second.  PCDH_ERR_ARG if an
 * entry is missing.                                                          */
int pcdh_locate(int64_t nq, const int64_t* qrow, const int64_t* qcol,
                int64_t nrows, const int64_t* rowptr, const int32_t* col,
                int64_t* pos);

/* pcdh_locate for the d (ci = cj = 0 .. d-1: Picard) or d*d (Newton) component
 * entries of every scalar pattern entry at once: pos[p * nq + q] = position of
 * (is_u[d rows[q] + ci[p]], is_u[d cols[q] + cj[p]]) in the CSR - the device
 * producer's map from the velocity block to the monolithic system's values. */
int pcdh_locate_blocks(int64_t nq, const int32_t* rows, const int32_t* cols,
                       int d, int npairs, const int32_t* ci, const int32_t* cj,
                       int64_t n_u, const int64_t* is_u, int64_t nrows,
                       const int64_t* rowptr, const int32_t* col, int64_t* pos);
/* element storage position (component-major) of every member of a contribution
 * list whose members are element entries laid out (cell, ab):
 * src[t] = (order[t] % nloc2) * ncells + order[t] / nloc2                     */
int pcdh_contribution_src(int64_t n, const int64_t* order, int64_t nloc2,
                          int64_t ncells, int32_t* src);

#ifdef __cplusplus
}
#endif
#endif
