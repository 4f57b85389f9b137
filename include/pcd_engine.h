/* pcd_engine.h - C ABI of the MI355X-native PCD preconditioner-apply engine.
 *
 * This is the drop-in boundary for fenapack's PCFieldSplit Schur path
 * (SURVEY.md section 8b).  Every entry point names the reference interface it
 * replaces (paths relative to the fenapack tree).  Calls that fenapack hands to
 * PETSc through petsc4py are marked [ext PETSc]: their arithmetic lives in
 * PETSc (version unpinned by the reference), restated in DESIGN.md.
 *
 * Conventions
 *  - every function returns 0 on success, a nonzero pcd_status otherwise; it
 *    never throws and never aborts; the message of the last failure on the
 *    calling thread is available from pcd_last_error().
 *  - matrices are CSR, fp64 values, int32 indices, passed as HOST arrays and
 *    copied at the call (the caller may free them afterwards).
 *  - vectors are fp64; `mem` says where the pointers live (PCD_MEM_HOST: the
 *    call copies in/out and synchronises; PCD_MEM_DEVICE: device pointers, work
 *    is enqueued on the engine's stream and the call does not synchronise
 *    unless stated).
 *  - a handle is not re-entrant; with several ranks, collective calls must be
 *    issued in the same order on every rank.
 */
#ifndef PCD_ENGINE_H
#define PCD_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcd_engine_s* pcd_handle;

enum pcd_status {
  PCD_OK = 0,
  PCD_ERR_ARG = 1,       /* bad argument / wrong call order                  */
  PCD_ERR_HIP = 2,       /* HIP runtime failure (message has the HIP string) */
  PCD_ERR_NOMEM = 3,
  PCD_ERR_STATE = 4,     /* operator or solver missing at setup/apply        */
  PCD_ERR_COMM = 5,      /* RCCL failure                                     */
  PCD_ERR_BREAKDOWN = 6, /* numerical breakdown (zero pivot / NaN)           */
  PCD_ERR_INTERNAL = 7   /* an exception of the host side was caught at the ABI */
};

/* fenapack/preconditioners.py:89,139,211,256 - the four python contexts */
enum pcd_variant {
  PCD_BRM1 = 1,   /* y = -Mp^-1 (x + Kp Ap^-1 bc(x))            :124-135 */
  PCD_BRM2 = 2,   /* y = -(I + Ap^-1 bc Kp) Mp^-1 x             :158-169 */
  PCDR_BRM1 = 3,  /* BRM1 - Rp^-1 x                             :239-252 */
  PCDR_BRM2 = 4   /* BRM2 - Rp^-1 x                             :285-298 */
};

/* operators owned by the context: preconditioners.py:29-30,79-82,197-207 and
 * the blocks PETSc's PCFIELDSPLIT extracts (field_split.py:82,89-93) */
enum pcd_mat {
  PCD_MAT_AP = 0, PCD_MAT_MP = 1, PCD_MAT_KP = 2, PCD_MAT_RP = 3,
  PCD_MAT_A00 = 4,  /* velocity block of the preconditioner matrix P       */
  PCD_MAT_A01 = 5,  /* discrete gradient block                             */
  PCD_MAT_A = 6,    /* whole operator in split ordering [u; p] (read only) */
  PCD_MAT_COUNT = 7
};

/* inner KSPs: preconditioners.py:29-34 (Ap, Mp), :180-183 (Rp);
 * field_split.py:93-98 (ksp0 = A00 solve) */
enum pcd_slot { PCD_KSP_AP = 0, PCD_KSP_MP = 1, PCD_KSP_RP = 2,
                PCD_KSP_A00 = 3, PCD_KSP_COUNT = 4 };

/* [ext PETSc] KSP types honoured (-<prefix>ksp_type), SURVEY 5 "Config" */
enum pcd_ksp_type { PCD_KSP_PREONLY = 0, PCD_KSP_RICHARDSON = 1,
                    PCD_KSP_CHEBYSHEV = 2, PCD_KSP_CG = 3,
                    /* KSPCG with -ksp_cg_single_reduction [ext PETSc]: both
                     * inner products of an iteration in one reduction (one
                     * 16-byte all-reduce per iteration on several ranks) */
                    PCD_KSP_CG_SR = 4 };
/* [ext PETSc] PC types honoured (-<prefix>pc_type) */
enum pcd_pc_type { PCD_PC_NONE = 0, PCD_PC_JACOBI = 1, PCD_PC_MG = 2,
                   /* the solve is a product of sparse factors the caller
                    * composed: pcd_set_inner_factor (under PCD_KSP_PREONLY) */
                   PCD_PC_EXPLICIT = 3 };

enum pcd_mem { PCD_MEM_HOST = 0, PCD_MEM_DEVICE = 1 };

/* keys of pcd_get_info */
enum pcd_info {
  PCD_INFO_N_U = 0, PCD_INFO_N_P = 1,
  PCD_INFO_ITS_AP = 2, PCD_INFO_ITS_MP = 3, PCD_INFO_ITS_RP = 4,
  PCD_INFO_ITS_A00 = 5,      /* iterations executed by the last inner solve */
  PCD_INFO_NUM_PCD_APPLY = 6, PCD_INFO_NUM_FS_APPLY = 7,
  PCD_INFO_GMRES_ITS = 8, PCD_INFO_GMRES_RNORM = 9,
  PCD_INFO_N_U_LOCAL = 10, PCD_INFO_N_P_LOCAL = 11,  /* rows of this rank */
  /* kernel path of the velocity block: components carried together by the
   * multi-component kernels (2 / 3; 0 = scalar path) and rows (nodes) per
   * workgroup of its stream kernels (0 = CSR-vector fallback; negative: rows
   * per block of the LDS-staged vector-tile kernels k_*_tc) */
  PCD_INFO_A00_COMPONENTS = 12, PCD_INFO_A00_ROWS_PER_WG = 13,
  PCD_INFO_RANKS = 14,       /* ranks of the attached communicator (0: none) */
  PCD_INFO_REORDERED = 15,   /* engine renumbering active: +1 velocity, +2 pressure */
  PCD_INFO_NNZ_BASE = 16,    /* + pcd_mat: stored nonzeros of that operator */
  PCD_INFO_LAUNCHES = 64,    /* kernel launches this host thread has issued so far (a
                              * replayed hipGraph counts as none: it is one hipGraphLaunch) */
  PCD_INFO_PEER_CALLS = 65,  /* halo exchanges / all-reduces issued as kernels of the
                              * stream (peer protocol), so far */
  PCD_INFO_BOOT_CALLS = 66,  /* ... and those that went through RCCL / the host transport */
  PCD_INFO_A00_KERNEL = 69,  /* kernel family the fused Chebyshev step on the velocity block
                              * takes: 0 row-wise CSR, 1 CSR stream (k_cheb_step_s), 2 multi-
                              * component stream (_sc), 3 vector tiles, direct form (_tc),
                              * 4 vector tiles, lane-major form (_lm: operators beyond the
                              * Infinity Cache) */
  PCD_INFO_PEER_DECLINED = 68, /* halo channels that did not fit the peer arena
                              * (PCD_PEER_ARENA_MB): their exchanges take the bootstrap
                              * path and a PCApply that holds one is not graph-captured */
  PCD_INFO_A00_MODEL_BYTES = 67  /* bytes one launch of the fused Chebyshev step on the velocity
                                  * block moves BY CONSTRUCTION with the kernel in force (matrix
                                  * stream + tile sources / row pointers + five vector streams);
                                  * bench.py's `kernel_model_bytes_per_launch` */
};

/* ---- lifetime ----------------------------------------------------------- */

/* BasePCDPC.create (preconditioners.py:28-34).  `device` is the HIP device
 * ordinal this rank drives (one process per GPU). */
int pcd_create(pcd_handle* out, int variant, int device);
/* idempotent; frees every device allocation of the handle */
int pcd_destroy(pcd_handle h);
const char* pcd_last_error(void);
/* share the caller's HIP stream (e.g. torch's current stream); NULL = own */
int pcd_set_stream(pcd_handle h, void* hip_stream);

/* ---- operators ----------------------------------------------------------- */

/* PCDInterface.setup_ksp_Ap/Mp/Rp, setup_mat_Kp (field_split_backend.py:
 * 67-83,115-139): hand over an already extracted p-p (or u-u / u-p)
 * submatrix.  First call for `which` fixes the sparsity pattern. */
int pcd_set_csr(pcd_handle h, int which, int64_t nrows, int64_t ncols,
                const int32_t* rowptr, const int32_t* colidx,
                const double* vals);
/* Rank-local hand-over (several ranks): the caller passes ONLY the rows this
 * rank owns - rows [r0, r1) of pcd_row_range - with GLOBAL column ids, the way
 * a partitioned assembly produces them (owned-rows-only construction,
 * SubfieldBC.h:136-155; PETSc MPIAIJ row blocks).  Nobody walks the global
 * matrix: ghost columns and the halo plan follow from the owned rows plus a
 * set-up handshake over the communicator (collective: every rank calls it for
 * the same operator in the same order).  pcd_update_values then carries this
 * rank's values only, in the order of the arrays handed over here.
 * `which`: Ap / Mp / Kp / Rp / A00 / A01. */
int pcd_set_csr_local(pcd_handle h, int which, int64_t nrows_global,
                      int64_t ncols_global, int64_t nrows_local,
                      const int32_t* rowptr_local, const int32_t* colidx_global,
                      const double* vals_local);
/* rows [*r0, *r1) that this rank owns of a space of `n_global` rows - a field
 * or a level of its multigrid hierarchy (`velocity` != 0: cuts fall on node
 * boundaries; call pcd_set_velocity_block first in 3-D); one GPU: [0, n). */
int pcd_row_range(pcd_handle h, int velocity, int64_t n_global, int64_t* r0,
                  int64_t* r1);
/* non-constant forms are re-assembled into the existing submatrix every outer
 * iteration (field_split_backend.py:82-83,285-291; assembling.py:103-104):
 * same pattern, new values, Jacobi diagonals re-derived. */
int pcd_update_values(pcd_handle h, int which, const double* vals, int mem);

/* PCDKSP.init_pcd, field_split.py:71-82 + [ext PETSc] PCSetUp_FieldSplit:
 * the monolithic operator A (caller's mixed numbering) and the two index sets
 * (dofmap_dofs_is, _field_split_utils.py:39-50).  The engine extracts A00,
 * A01 and the split-ordered operator itself.  `pvals` (may be NULL) are the
 * values of the preconditioning matrix P on the SAME pattern (a_pc,
 * assembling.py:143-148); NULL means P = A (nonlinear_solvers.py:75). */
int pcd_set_system(pcd_handle h, int64_t n, const int32_t* rowptr,
                   const int32_t* colidx, const double* vals,
                   const double* pvals,
                   int64_t n_u, const int32_t* is_u,
                   int64_t n_p, const int32_t* is_p);
/* Rank-local form (several ranks): this rank's rows of the monolithic matrix
 * only - what a partitioned assembly holds (PETSc MPIAIJ row blocks; the
 * reference never touches a row it does not own: SubfieldBC.h:136-155).
 * `rows[i]` is the caller's global index of local row i: the rank's velocity
 * rows is_u[u0 .. u1) followed by its pressure rows is_p[p0 .. p1), the ranges
 * being pcd_row_range's; `colidx` holds the caller's global indices; the index
 * sets are passed whole (O(n) integers - the matrix is what is O(nnz / R)).
 * Collective.  Keeps the caller's dof order (PCD_REORDER needs the whole
 * graph).  Afterwards pcd_update_system takes the values of THESE rows, in the
 * order handed over here.  One GPU: the same call with every row. */
int pcd_set_system_local(pcd_handle h, int64_t n, int64_t n_u,
                         const int32_t* is_u, int64_t n_p, const int32_t* is_p,
                         int64_t nrows_local, const int32_t* rows,
                         const int32_t* rowptr_local,
                         const int32_t* colidx_global, const double* vals,
                         const double* pvals);
/* Engine renumbering (csrc/pcd_reorder.hpp).  The index sets arrive as the
 * caller's dofmap gives them (_field_split_utils.py:39-50: dofmap.dofs() as
 * is); when that numbering is not local - mean |row - col| / n of the velocity
 * block above 0.1; a geometric numbering gives a few per cent, a random one
 * 0.33 - pcd_set_system renumbers the velocity nodes (reverse Cuthill-McKee
 * on the node graph of A00, components of a node kept together) and the
 * pressure dofs (by the velocity nodes they couple to), multigrid levels
 * inherit their order through the prolongations when they are handed over
 * finest first.  Invisible at this boundary: operators, BC indices and field
 * vectors keep crossing it in the caller's numbering (at the price of one
 * gather / scatter on the by-parts entry points; the fused fieldsplit / GMRES
 * calls fold it into the split gather they perform anyway).
 * mode 0: never, 1: auto (default), 2: always; the environment variable
 * PCD_REORDER = none | auto | always overrides.  Call before pcd_set_system.
 * Not available together with the device producer (pcd_fe_*) or
 * pcd_set_csr_local / pcd_set_system_local / pcd_mg_set_level_local, which
 * address entries in the caller's numbering. */
int pcd_set_reorder(pcd_handle h, int mode);
/* in-place re-assembly of A (and P) between Newton steps, SURVEY 3.1 */
int pcd_update_system(pcd_handle h, const double* vals, const double* pvals,
                      int mem);

/* SubfieldBC (SubfieldBC.h:48-53,92-160): subfield indices + values of the
 * PCD Dirichlet condition in the contiguous pressure numbering. */
int pcd_set_bc(pcd_handle h, int64_t n_bc, const int32_t* idx,
               const double* vals);

/* ---- inner solvers ([ext PETSc] options, demo_navier-stokes-pcd.py:151-165)
 * rtol <= 0 means "run exactly max_it iterations" (-ksp_norm_type none). */
int pcd_set_inner(pcd_handle h, int slot, int ksp_type, int pc_type,
                  int max_it, double rtol, double emin, double emax);

/* ---- geometric multigrid as inner preconditioner (pc_type = PCD_PC_MG) ----
 * The reference's "iterative" configuration runs hypre BoomerAMG V-cycles
 * under Richardson for A00, Ap (and Rp): demo_navier-stokes-pcd.py:153-160,
 * demo_unsteady-navier-stokes-pcdr.py:167-170.  hypre does not exist here;
 * this is the device-native counterpart with the same role and call shape
 * ([ext PETSc] PCMG: PCMGSetLevels / PCMGSetInterpolation / per-level
 * operators): a multiplicative V(nu_pre, nu_post) cycle whose smoother is the
 * fused Chebyshev-Jacobi kernel, restriction = transpose of the prolongation,
 * level 0 solved by an explicit inverse handed over as a (dense) CSR matrix.
 * Level nlevels-1 is the slot's own operator.  Supported under
 * PCD_KSP_PREONLY (one cycle) and PCD_KSP_RICHARDSON (max_it cycles). */
int pcd_mg_begin(pcd_handle h, int slot, int nlevels, int nu_pre, int nu_post);
/* level 0: (n, rowptr, colidx, vals) = A_0^-1, no prolongation, no bounds.
 * 0 < level < nlevels-1: A_level, prolongation level-1 -> level
 *   (p_rows x p_cols = n_level x n_{level-1}) and the bounds [emin, emax] of
 *   diag(A)^-1 A used by the smoother.
 * level nlevels-1: rowptr == NULL (operator of the slot), prolongation and
 *   bounds as above. */
int pcd_mg_set_level(pcd_handle h, int slot, int level, int64_t n,
                     const int32_t* rowptr, const int32_t* colidx,
                     const double* vals, int64_t p_rows, int64_t p_cols,
                     const int32_t* prowptr, const int32_t* pcolidx,
                     const double* pvals, double emin, double emax);
/* Rank-local form for a PARTITIONED level (several ranks; a level of more than
 * PCD_REPLICATE_BELOW rows - smaller ones are replicated on every rank and keep
 * the global form): this rank's rows of the level operator (n_global x
 * n_global, GLOBAL column ids; rowptr == NULL on the finest level), its rows
 * of the prolongation (fine rows owned x p_cols global coarse columns) and,
 * when the level below is partitioned too, its rows of the restriction P^T
 * (coarse rows owned x n_global fine columns - [ext PETSc] MatTranspose of the
 * distributed P; NULL when the level below is replicated: the transpose of
 * the local rows then sums this rank's contribution and an all-reduce in the
 * cycle completes it).  Row ranges: pcd_row_range of a field of that many
 * rows.  Collective.  pcd_mg_update_values then carries this rank's values. */
int pcd_mg_set_level_local(pcd_handle h, int slot, int level, int64_t n_global,
                           int64_t nrows_local, const int32_t* rowptr,
                           const int32_t* colidx, const double* vals,
                           int64_t p_cols, const int32_t* prowptr,
                           const int32_t* pcolidx, const double* pvals,
                           int64_t r_rows_local, const int32_t* rrowptr,
                           const int32_t* rcolidx, const double* rvals,
                           double emin, double emax);
/* Row cuts of a partitioned level other than the even ones of pcd_row_range:
 * an ALGEBRAIC hierarchy built rank by rank gives every rank the aggregates of
 * its own rows ([ext PETSc] PCGAMG's coarse ownership), so coarse levels are
 * cut where the aggregates fall.  bounds[nranks + 1], from 0 to n, ascending,
 * on node boundaries for PCD_KSP_A00; the same on every rank.  After
 * pcd_mg_begin and before this level or the level above it is set
 * (pcd_mg_set_level_local then checks the row counts against these cuts).
 * The finest level always has the field's cuts. */
int pcd_mg_set_level_cuts(pcd_handle h, int slot, int level, int64_t n,
                          const int64_t* bounds);
/* re-assembled operator of one level (same pattern) and refreshed bounds;
 * vals == NULL only refreshes the bounds (finest level) */
int pcd_mg_update_values(pcd_handle h, int slot, int level, const double* vals,
                         double emin, double emax, int mem);

/* ---- pre-composed inner solves -------------------------------------------
 * A fixed number of Chebyshev steps, or one level of a V(nu1, nu2) cycle, is
 * a fixed linear operator.  Where a launch is latency-bound (operators of
 * <= ~10^5 rows: DESIGN.md 4) the caller may multiply adjacent steps out on
 * the host (fenapack_amd/compose.py) and hand over the products; the engine
 * then applies ONE sparse product where it ran several dependent kernels.
 * Same role as PETSc composing a KSP from fewer, fatter operators would have
 * ([ext PETSc]: none of this is visible in the reference, whose KSPSolve calls
 * at preconditioners.py:130,133,162,167 are opaque); identical results up to
 * re-association (tests: 1e-11 against the step-by-step path and the oracle).
 *
 * pcd_mg_set_fused: level `level` (>= 1) of slot's hierarchy as
 *   r_c = Wd b  (Wd = R (I - A H1): residual + restriction of the pre-smoothed
 *   x1 = H1 b, which the ordinary kernels still compute) and
 *   x = Wu [x1 | r_c | e_c | b]  (prolongation, correction, post-smoothing)
 *   (Wd: n_c x n;  Wu: n x (2n + 2n_c)); wd_rowptr == NULL drops it.
 *   Any later update of that level's values or smoother bounds drops it too.
 * pcd_set_inner_factor: factor k of nfactors of x = W_{nfactors-1} ... W_0 b
 *   for pc_type PCD_PC_EXPLICIT (e.g. Chebyshev(5)+Jacobi on the constant
 *   M_p as two factors).  After an update of the slot's operator the factors
 *   are stale: the solve then runs the recurrence they stand for (max_it
 *   Chebyshev-Jacobi steps with the emin / emax given to pcd_set_inner) until
 *   new factors arrive; without valid bounds it fails (PCD_ERR_STATE). */
int pcd_mg_set_fused(pcd_handle h, int slot, int level,
                     int64_t wd_rows, int64_t wd_cols, const int32_t* wd_rowptr,
                     const int32_t* wd_col, const double* wd_val,
                     int64_t wu_rows, int64_t wu_cols, const int32_t* wu_rowptr,
                     const int32_t* wu_col, const double* wu_val);
int pcd_set_inner_factor(pcd_handle h, int slot, int k, int nfactors, int64_t n,
                         const int32_t* rowptr, const int32_t* colidx,
                         const double* vals);

/* BasePCDPC.setUp / BasePCDRPC.setUp (preconditioners.py:71-85,191-207) and
 * ksp.setUp() of field_split_backend.py:254-263: checks every operator the
 * variant needs is present, derives Jacobi diagonals, allocates work vectors
 * (get_work_vecs, preconditioners.py:52-61). */
int pcd_setup(pcd_handle h);

/* ---- the hot path -------------------------------------------------------- */

/* PCDPC_BRM1/BRM2/PCDRPC_BRM1/BRM2.apply(pc, x, y): x borrowed (n_p),
 * y overwritten (n_p). */
int pcd_apply(pcd_handle h, const double* x, double* y, int mem);

/* [ext PETSc] PCApply_FieldSplit_Schur, factorisation UPPER (field_split.py:
 * 54-57): y_p = pcd(x_p); y_u = A00^-1 (x_u - A01 y_p).  x, y have length
 * n_u + n_p in the caller's mixed numbering. */
int pcd_fieldsplit_apply(pcd_handle h, const double* x, double* y, int mem);

/* [ext PETSc] KSPSolve of PCDKSP (field_split.py:46-57): restarted GMRES,
 * right preconditioning by pcd_fieldsplit_apply, classical Gram-Schmidt,
 * zero initial guess, stop when ||b - A x|| <= rtol*||b|| (or atol).
 * Synchronises.  *its = iterations, *rnorm = final residual estimate. */
int pcd_gmres_solve(pcd_handle h, const double* b, double* x, int mem,
                    double rtol, double atol, int restart, int max_it,
                    int* its, double* rnorm);

/* ---- building blocks, exported for parity tests and profiling ------------ */

/* Mat.mult (preconditioners.py:131,164) on one stored operator */
int pcd_spmv(pcd_handle h, int which, const double* x, double* y, int mem);
/* KSP.solve (preconditioners.py:130,133,162,166,249,295) of one slot */
int pcd_inner_solve(pcd_handle h, int slot, const double* b, double* x,
                    int mem);
/* SubfieldBC::apply (SubfieldBC.h:74-77,162-182) on a pressure vector */
int pcd_apply_bc(pcd_handle h, double* x, int mem);

int pcd_get_info(pcd_handle h, int key, double* out);
int pcd_synchronize(pcd_handle h);
/* capture the fixed-iteration fieldsplit apply into a hipGraph and replay it
 * (launch-bound at the 2D sizes: SURVEY 7, hard part 3); 0 = eager launches */
int pcd_graph_enable(pcd_handle h, int on);
/* Measurement aid of bench.py: the dominant kernel - the fused
 * Chebyshev-Jacobi step on the finest velocity operator, [ext PETSc]
 * KSPCHEBYSHEV + PCJACOBI inside the A00 solve of field_split.py:96-100 - timed
 * WHERE IT RUNS: `reps` eager fieldsplit applies on device vectors x, y with an
 * event pair around every such launch; *us = mean microseconds per launch,
 * *launches = launches timed.  (A back-to-back loop on one operator keeps the
 * caches warm; inside the multigrid cycle the coarser levels have used them.) */
int pcd_probe_a00_step(pcd_handle h, const double* x, double* y, int reps,
                       double* us, int* launches);
/* Streaming bandwidth of this GPU measured by a kernel of this library
 * (16 B per lane, unit stride): kind 0 copy, 1 triad, 2 read-only sweep,
 * 3 read-mostly (6 % writes: the mix of the dominant kernel), 4 read-only
 * with non-temporal loads; arrays of `bytes` each.
 * The practical roof the roofline numbers are quoted next to. */
int pcd_bandwidth_probe(pcd_handle h, int kind, int64_t bytes, int reps,
                        double* gbs);

/* ---- multi-GPU: contiguous row blocks per rank, RCCL over xGMI ----------- */

/* SURVEY 8e.  `nccl_unique_id` is the 128-byte ncclUniqueId produced on rank 0
 * (pcd_comm_unique_id) and broadcast by the host (torch.distributed store).
 * After this call set_csr/set_system take GLOBAL matrices on every rank and
 * keep only the owned row block [row_begin, row_end) of each field.
 *
 * Errors and the communicator: every compute entry point of a partitioned
 * engine is COLLECTIVE (same calls, same order on every rank - SURVEY 8b
 * "Threading").  A nonzero status on one rank (a HIP error, a GMRES
 * breakdown) is returned on that rank only; the others are then waiting in a
 * collective the failed rank will never join.  Treat any engine error in a
 * multi-rank run as fatal to the job: exit the process (the launcher of
 * bench.py ends the other ranks), do not retry on the same communicator.
 * Data-dependent decisions inside the engine (GMRES stopping test, CG
 * convergence) are taken from all-reduced numbers, so they agree on all
 * ranks by construction. */
/* velocity components per node (2 or 3): row cuts of velocity operators fall
 * on node boundaries.  Call before handing operators over; default 2. */
int pcd_set_velocity_block(pcd_handle h, int ncomp);
int pcd_comm_unique_id(void* out128);
int pcd_comm_init(pcd_handle h, int rank, int nranks,
                  const void* nccl_unique_id);
/* Communicator over a HOST transport the caller owns (the reference's ranks
 * talk through PETSc's MPI communicator, PCDKSP(comm):
 * fenapack/field_split.py:46-57): two callbacks on host buffers, collective
 * like their MPI counterparts.
 *   allreduce(ctx, buf, count): in-place sum of `count` doubles over all ranks
 *   exchange(ctx, nsend, send_peers, send_bufs, send_counts,
 *            nrecv, recv_peers, recv_bufs, recv_counts): one neighbour
 *     exchange; several messages between one pair match in posting order
 * (non-zero return = failure).  Device data is staged through the host for
 * these calls, so they carry set-up traffic; the hot path between the
 * processes of one node then runs on the one-shot peer-write protocol
 * (csrc/pcd_peer.hpp: arenas shared through HIP IPC, halo exchange and dot
 * products as single kernels of the engine's stream, capturable into a
 * hipGraph) - also between two processes that share ONE GPU, where RCCL
 * cannot build a communicator.  PCD_COMM_PEER=0 keeps everything on the
 * transport given here (or on RCCL for pcd_comm_init). */
typedef int (*pcd_host_allreduce_fn)(void* ctx, double* buf, int64_t count);
typedef int (*pcd_host_exchange_fn)(void* ctx, int nsend, const int* send_peers,
                                    double* const* send_bufs,
                                    const int64_t* send_counts, int nrecv,
                                    const int* recv_peers,
                                    double* const* recv_bufs,
                                    const int64_t* recv_counts);
int pcd_comm_init_host(pcd_handle h, int rank, int nranks,
                       pcd_host_allreduce_fn allreduce,
                       pcd_host_exchange_fn exchange, void* ctx);
/* TEST backend with the same two primitives (halo exchange, all-reduce):
 * `nranks` handles driven by `nranks` threads of ONE process on one GPU.
 * *group must be NULL for the first caller and is shared by the others.  It
 * exists because RCCL refuses two ranks on one device, so that partitioning,
 * column localisation and halo plans can be exercised on a single-GPU box. */
int pcd_comm_init_threads(pcd_handle h, int rank, int nranks, void** group);
/* Host-only probe of the same partitioning code (no device call, no handle):
 * row block, localised columns and halo plan of `rank`.  counts[7] = local
 * rows, owned cols, ghosts, #send peers, #recv peers, first row, first col.
 * Used by the world_size-2 gloo tests on CPU. */
int pcd_dist_probe(int64_t nrows, int64_t ncols, const int32_t* rowptr,
                   const int32_t* colidx, const double* vals, int rank,
                   int nranks, int even_rows, int even_cols, int64_t* counts,
                   int32_t* out_rowptr, int32_t* out_col, double* out_val,
                   int32_t* send_peers, int32_t* send_off, int32_t* send_idx,
                   int32_t* recv_peers, int32_t* recv_off);

/* ---- device operator producer ------------------------------ *
 * Replaces, for the fixed P2/P1 Picard forms, what the reference re-assembles
 * on the host every nonlinear iteration through DOLFIN [ext]:
 *   fenapack/assembling.py:151-155  system_matrix (the velocity block changes),
 *   fenapack/assembling.py:165-171  kp,
 *   fenapack/nonlinear_solvers.py:85-112  F (the matrix-dependent part),
 * plus the re-discretised coarse velocity operators of `-pc_mg_galerkin none`
 * and their smoother bounds.  Everything is assembled in HBM from the iterate
 * and written into the engine's operators in place.  One GPU.  Arrays are
 * host pointers, stored COMPONENT-MAJOR ([entry][cell]) so that device lanes
 * read unit-stride; `nv = dim + 1`, `na = 6 (dim 2) | 10 (dim 3)`.            */

/* quadrature / basis tables: qw[nq] (weights, sum 1), phi[nq][na] (P2 basis),
 * dphi[nq][na][nv] (d phi_a / d lambda_k), psi[nq][nv] (P1 basis).  `nlevels`
 * FE levels, coarsest first - one per multigrid level of the velocity solve. */
int pcd_fe_begin(pcd_handle h, int dim, int nlevels, int nq, const double* qw,
                 const double* phi, const double* dphi, const double* psi);
/* Mesh + plan of one level: dofs2[na][ncells] scalar P2 nodes of each cell,
 * gradlam[nv*dim][ncells] barycentric gradients, measure[ncells].  The scalar
 * operator F (velocity block = F x I_dim) has nnz_f entries in CSR order;
 * entry k = f_const[k] + sum_{t in f_ptr[k]..f_ptr[k+1]} cell_matrix[f_src[t]]
 * with f_src indexing the component-major element storage
 * ((a*na + b)*ncells + cell); f_keep[k] = 0 removes Dirichlet rows/columns,
 * diag_pos/diag_val then set the Dirichlet diagonal.  inject[nn2] = the same
 * node on the next finer level (NULL on the finest level).                   */
int pcd_fe_set_level(pcd_handle h, int level, int64_t ncells, int64_t nn2,
                     const int32_t* dofs2, const double* gradlam,
                     const double* measure, int64_t nnz_f, const int32_t* f_ptr,
                     const int32_t* f_src, const double* f_const,
                     const unsigned char* f_keep, int64_t n_diag,
                     const int32_t* diag_pos, const double* diag_val,
                     const int32_t* inject);
/* Several ranks: the plan handed to pcd_fe_set_level for `level` covers the node
 * rows [node_row0, node_row0 + n_node_rows) of the level's scalar operator only -
 * this rank's rows (pcd_row_range): the cells that touch them, the contribution
 * lists / constants / masks of their entries; dofs2 and inject keep GLOBAL node
 * numbers, nn2 the global node count.  A rank then assembles and stores 1 / R
 * of a partitioned level (the reference's ranks assemble their own cells:
 * assembling.py:151-155 on DOLFIN's partitioned mesh); vectors stay replicated.
 * Picard block, re-discretised levels.  pcd_fe_bind_pattern then takes the
 * GLOBAL entry offsets of these rows (rowptr[0] = entries before them) and
 * their column indices; pcd_fe_bind_system / the mass values of
 * pcd_fe_bind_residual these rows' entries.                                    */
int pcd_fe_set_rows(pcd_handle h, int level, int64_t node_row0, int64_t n_node_rows);
/* A coarse level as the Galerkin product of the next finer one
 * (-pc_mg_galerkin both, [ext PETSc] MatPtAP) instead of a re-discretisation:
 * B = F_finer P has nnz_b entries, entry e = sum_t b_w[t] * F_finer[b_src[t]],
 * t in b_ptr[e]..b_ptr[e+1]; this level's F = P^T B likewise from c_*.  The
 * weights are the prolongation's entries: numeric product = two gathers.    */
int pcd_fe_set_level_galerkin(pcd_handle h, int level, int64_t nnz_f,
                              int64_t nnz_b, const int64_t* b_ptr,
                              const int32_t* b_src, const double* b_w,
                              const int64_t* c_ptr, const int32_t* c_src,
                              const double* c_w);
/* The same coarse level by a NUMERIC SPARSE PRODUCT on fixed patterns - the
 * reference's transposeMatMult(..., result=) (fenapack/field_split_backend.py:
 * 160-166: pattern reused, values recomputed) and the per-iteration AMG set-up
 * of demo_navier-stokes-pcd.py:153-160 on a kept aggregation.  P (n_fine x
 * n_coarse, scalar) and P^T with their values; the pattern of the finer
 * level's scalar F; the STRUCTURAL patterns of B = F P (n_fine rows) and of
 * this level's F = P^T B (n_coarse rows) from a symbolic product run once on
 * the host (pcd_host.h pcdh_spgemm_*).  Columns sorted in every row.  Every
 * entry is summed in the order of the first factor's row: reproducible.     */
int pcd_fe_set_level_product(pcd_handle h, int level, int64_t n_fine,
                             int64_t n_coarse, const int32_t* p_rowptr,
                             const int32_t* p_col, const double* p_val,
                             const int32_t* pt_rowptr, const int32_t* pt_col,
                             const double* pt_val, const int32_t* f_rowptr,
                             const int32_t* f_col, const int32_t* b_rowptr,
                             const int32_t* b_col, const int32_t* c_rowptr,
                             const int32_t* c_col);
/* The product BY ROWS: the coarse levels of a hierarchy whose operators are
 * partitioned (fenapack/assembling.py:98-106 and field_split_backend.py:79-83,
 * 285-291: every rank re-assembles fp / kp for ITS rows each outer iteration;
 * SubfieldBC.h:136-155: owned rows only; the AMG set-up re-run on them,
 * demo_navier-stokes-pcd.py:153-160).  This rank holds n_own node rows of the
 * finer level's scalar F (f_*: global columns), P_ext = the prolongation rows
 * of its own and its halo nodes (n_fine rows, global shape), the transpose of
 * its own rows (pt_*: n_coarse x n_own), the structural patterns of
 * B = F_rows P_ext (n_own rows) and of its TERMS T = P_own^T B (n_coarse rows).
 * Terms of other ranks' coarse rows are consecutive in T: sends[3q..] = (first
 * term, count, wire offset); wire_len doubles, alike on every rank, are
 * delivered by ONE all-reduce.  adds[4q..] = (source: 0 T / 1 wire, source
 * offset, count, offset into pos): applied in order (rank order), they land in
 * this rank's n_out entries.  gather_total > 0: the level is replicated - all
 * ranks' rows, this rank's at gather_off, form its F.  Otherwise the level
 * keeps node rows [node_row0, node_row0 + n_node_rows) (like pcd_fe_set_rows). */
int pcd_fe_set_level_product_rows(
    pcd_handle h, int level, int64_t n_own, int64_t n_fine, int64_t n_coarse,
    const int32_t* pe_rowptr, const int32_t* pe_col, const double* pe_val,
    const int32_t* pt_rowptr, const int32_t* pt_col, const double* pt_val,
    const int32_t* f_rowptr, const int32_t* f_col, const int32_t* b_rowptr,
    const int32_t* b_col, const int32_t* t_rowptr, const int32_t* t_col,
    int64_t n_out, int64_t wire_len, int n_sends, const int64_t* sends,
    int n_adds, const int64_t* adds, const int32_t* pos, int64_t gather_off,
    int64_t gather_total, int64_t node_row0, int64_t n_node_rows);
/* The blocks of pcd_fe_bind_residual hold this rank's rows only (global shape;
 * SubfieldBC.h:136-155): their products are completed by all-reduces.       */
int pcd_fe_set_residual_rows(pcd_handle h, int on);
/* SUPG-stabilised preconditioner matrix (fenapack/stabilization.py:39-68, used
 * at demo_navier-stokes-pcd.py:122-127): cell sizes h of one re-discretised
 * level (ncells values), viscosity, P2 basis at the cell midpoint (na values),
 * and the rule the streamline-diffusion term is integrated with (its integrand
 * has degree 6 with a P2 wind): qw_s[nq_s], phi_s[nq_s][na], dphi_s[nq_s][na][nv].
 * Afterwards A00/A01 and the multigrid come from the stabilised operator, the
 * system matrix from the unstabilised one (P != A: pvals of pcd_set_system).  */
int pcd_fe_set_supg(pcd_handle h, int level, const double* cell_h, double nu,
                    const double* phi_mid, int nq_s, const double* qw_s,
                    const double* phi_s, const double* dphi_s);
/* Several ranks (pcd_comm_init before pcd_fe_begin; the reference's `mpirun
 * -np N` over the same script, test/regression/test.py:186-195): the producer
 * is REPLICATED - every rank assembles every level from the replicated iterate
 * (a few ms per nonlinear step) and evaluates the whole residual, the linear
 * solve is partitioned.  This call gives the engine the scalar CSR pattern of
 * a level (nn2 + 1 row pointers, nnz_f sorted columns - the order of
 * pcd_fe_get_level_values) = the layout of the global F x I_d it cuts its rows
 * from; on the finest level it also creates the replicated operator the
 * residual applies.  Host-pointer vectors of the pcd_fe_* calls are GLOBAL and
 * identical on every rank.  Not needed on one GPU.  With the Newton block
 * call it AFTER pcd_fe_set_newton (the finest level then also gets the
 * replicated coupled operator of the residual's boundary-defect term).        */
int pcd_fe_bind_pattern(pcd_handle h, int level, int64_t nn2,
                        const int32_t* rowptr, const int32_t* colidx);
/* Newton linearisation (`--nls newton`: demo_navier-stokes-pcd.py:42,113-116,
 * J = derivative(F, w)): the velocity block becomes F x I_d + N(w) with
 * N_ij = (phi_b d_j w_i, phi_a), d*d scalar matrices on the pattern of F; the
 * block lives on (pattern of F) x ones(d, d).  Call once per level after
 * pcd_fe_set_level[_galerkin] and before pcd_fe_bind_system /
 * pcd_fe_bind_coarse_inverse.  pos[(i*d+j)*nnz_f + k] = where block (i, j) of
 * scalar entry k sits in the values of that level's operator: the multigrid
 * level's CSR (0 < level < finest), PCD_MAT_A00 (finest), the CSR handed to
 * pcd_fe_bind_coarse_inverse (level 0: then of d*n0 rows, inverted whole).
 * The residual of pcd_fe_residual / pcd_fe_picard_solve stays that of the
 * nonlinear problem; only the Jacobian changes (nonlinear_solvers.py:85-112). */
int pcd_fe_set_newton(pcd_handle h, int level, const int32_t* pos);
/* sys_pos[c*nnz_f + k] = position, in the value array given to
 * pcd_set_system, of entry k of the finest F for velocity component c; after
 * pcd_fe_set_newton: sys_pos[(i*d+j)*nnz_f + k] for every block (i, j)       */
int pcd_fe_bind_system(pcd_handle h, const int64_t* sys_pos);
/* Kp = scale * (w . grad p, q) + kp_const on the pattern of PCD_MAT_KP
 * (same plan layout with nv x nv element matrices; kp_const may be NULL)     */
int pcd_fe_bind_kp(pcd_handle h, int64_t nnz_kp, const int32_t* kp_ptr,
                   const int32_t* kp_src, const double* kp_const, double scale);
/* Several ranks, plans cut by rows (pcd_fe_set_rows): the entries bound by
 * pcd_fe_bind_kp are this rank's pressure rows, `entry_offset` entries into the
 * operator's `nnz_global` values; the cells of the finest level's plan then
 * cover the cells that touch these rows as well.                               */
int pcd_fe_set_kp_rows(pcd_handle h, int64_t entry_offset, int64_t nnz_global);
/* replace kp_const (terms the host keeps assembling every iteration, e.g. the
 * BRM2 boundary integral of demo_navier-stokes-pcd.py:131-135); NULL = none  */
int pcd_fe_set_kp_const(pcd_handle h, const double* kp_const);
/* BRM2 boundary term of Kp, -(1/nu) int_inflow (w.n) p q ds
 * (demo_navier-stokes-pcd.py:131-135), on the device (2-D): per inflow edge
 * its P2 nodes nodes[3][nb] (start, end, midpoint), outward unit normals
 * normals[2][nb], lengths[nb]; aff_pos[n_aff] = the distinct entries of Kp
 * it touches, entry i += sum_t aff_w[t] * loc[aff_src[t]], t in
 * aff_ptr[i]..aff_ptr[i+1], loc = the local 2 x 2 edge matrices stored
 * [(i*2+j)][edge]; aff_w carries -1/nu.  In space (three velocity components)
 * the facets are the inflow FACES: nodes[6][nb] (three vertices, then the
 * midpoints of the edges 01, 02, 12), normals[3][nb], lengths = areas, loc =
 * local 3 x 3 matrices [(i*3+j)][face] - the reference's form is
 * dimension-free.                                                            */
int pcd_fe_bind_robin(pcd_handle h, int64_t nb, const int32_t* nodes,
                      const double* normals, const double* lengths,
                      int64_t n_aff, const int32_t* aff_pos,
                      const int64_t* aff_ptr, const int32_t* aff_src,
                      const double* aff_w);
/* the multigrid hierarchy of inner solve `slot` follows the FE levels; after
 * every update the smoother bounds of level l >= 1 become
 * [emin_factor, emax_factor] * lambda_max(D^-1 A_l) (power iteration, `iters`
 * steps cold, a quarter of that warm) - [ext PETSc] -mg_levels_esteig        */
int pcd_fe_bind_mg(pcd_handle h, int slot, double emin_factor,
                   double emax_factor, int iters);
/* Invert the coarsest level on the device after every update (Gauss-Jordan
 * with partial pivoting; level 0 of the hierarchy must be the dense
 * (dim n0)^2 inverse): rowptr/colidx = CSR pattern of the coarsest scalar
 * operator, n0 <= 8192.  Without it the caller fetches level 0, inverts and
 * calls pcd_mg_update_values itself.                                         */
int pcd_fe_bind_coarse_inverse(pcd_handle h, int64_t n0, const int32_t* rowptr,
                               const int32_t* colidx);
/* Assemble at the iterate xu (velocity dofs, fieldsplit-local numbering) and
 * refresh system, A00, A01, Kp, multigrid levels in place.  If v/ru are given:
 * ru = (unconstrained velocity operator) v.  The coarsest level is left to
 * the caller: fetch it with pcd_fe_get_level_values(h, 0, .), invert, hand the
 * inverse to pcd_mg_update_values.                                           */
int pcd_fe_update(pcd_handle h, const double* xu, const double* v, double* ru,
                  int mem);
/* ---- the nonlinear iteration itself -------------------------- *
 * Constant pieces of the residual F (fenapack/nonlinear_solvers.py:85-112 ->
 * assembling.py:143-148 with DOLFIN's symmetric BC application [ext]): the
 * unconstrained blocks B^T (n_u x n_p) and B (n_p x n_u) as CSR, the velocity
 * Dirichlet dofs (fieldsplit-local indices) with the diagonal values of their
 * rows, and for time stepping the scalar mass values on the pattern of F with
 * idt = 1/dt (mass_vals may be NULL when idt == 0).                           */
int pcd_fe_bind_residual(pcd_handle h, const int32_t* bt_rowptr,
                         const int32_t* bt_col, const double* bt_val,
                         const int32_t* b_rowptr, const int32_t* b_col,
                         const double* b_val, int64_t n_bc,
                         const int32_t* bc_idx, const double* bc_mult,
                         const double* mass_vals, double idt);
/* boundary values of those dofs (time dependent inflow: demo_unsteady-...:88) */
int pcd_fe_set_bc_values(pcd_handle h, const double* g);
/* previous time level (velocity dofs); NULL drops the idt M u0 term          */
int pcd_fe_set_previous(pcd_handle h, const double* u0, int mem);
/* operators refreshed at x (caller's mixed numbering), b = F(x), *norm = |b| */
int pcd_fe_residual(pcd_handle h, const double* x, double* b, int mem,
                    double* norm);
/* The whole Picard loop on the device (nonlinear_solvers.py:28-82 around
 * dolfin::NewtonSolver [ext]): b = F(x); stop on |b| < atol or |b|/r0 < rtol;
 * right-preconditioned GMRES for J dx = b; x -= relax dx.  r0 <= 0: the first
 * residual of this call is the reference norm.  lin_its[max_it] and
 * res_hist[max_it + 1] receive the history.  x is updated in place.          */
int pcd_fe_picard_solve(pcd_handle h, double* x, int mem, double r0,
                        double rtol, double atol, int max_it, double relax,
                        double lin_rtol, double lin_atol, int restart,
                        int lin_max_it, int* n_it, int* lin_its,
                        double* res_hist, int* converged);
int pcd_fe_get_level_values(pcd_handle h, int level, double* out);
/* Newton term of one level as last assembled: d*d*nnz_f values [(i*d+j)][k]  */
int pcd_fe_get_newton_values(pcd_handle h, int level, double* out);
int pcd_fe_get_kp_values(pcd_handle h, double* out);
int pcd_fe_get_bounds(pcd_handle h, int level, double* emin, double* emax);

#ifdef __cplusplus
}
#endif
#endif /* PCD_ENGINE_H */
