/* pcd_oracle.c - CPU restatement of fenapack's PCD / fieldsplit apply path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (fenapack_amd + libpcd_hip.so) never calls into it.
 *
 * What it restates (paths relative to the fenapack tree):
 *   - the four PCPYTHON apply bodies, op for op:
 *       fenapack/preconditioners.py:124-135 (PCDPC_BRM1), :158-169 (BRM2),
 *       :239-252 (PCDRPC_BRM1), :285-298 (PCDRPC_BRM2);
 *   - SubfieldBC::apply = VecSetValues(INSERT): fenapack/SubfieldBC.h:162-182;
 *   - the algorithm PCDKSP selects and PETSc then runs (field_split.py:46-57):
 *     right-preconditioned restarted GMRES over PCFIELDSPLIT/SCHUR/UPPER.
 *
 * Pinning: the apply bodies are pinned against goldens produced by running
 * the reference's own preconditioners.py (tests/golden/make_goldens.py).  The
 * Krylov/Chebyshev/fieldsplit arithmetic belongs to PETSc, which is NOT in
 * /root/reference and whose version the reference does not pin:
 * those parts restate PETSc's published algorithms (KSPCG with Jacobi and the
 * natural norm, KSPCHEBYSHEV's three-term recurrence with user eigenvalue
 * bounds, KSPRICHARDSON scale 1, KSPGMRES with classical Gram-Schmidt and
 * right preconditioning, PCApply_FieldSplit_Schur UPPER) and are
 * "parity unpinned" at the PETSc boundary; they are checked by algebraic
 * properties instead (tests/test_oracle.py).
 *
 * Plain scalar C on purpose: one thread, sequential summation order.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Optional OpenMP build (make omp): used ONLY as the multi-core timing
 * baseline of bench.py; parity tests use the serial build (sequential
 * summation order). */
#ifdef _OPENMP
#include <omp.h>
#define OMP_FOR _Pragma("omp parallel for schedule(static)")
#define OMP_SUM _Pragma("omp parallel for schedule(static) reduction(+ : s)")
#else
#define OMP_FOR
#define OMP_SUM
#endif

enum { MAT_AP = 0, MAT_MP, MAT_KP, MAT_RP, MAT_A00, MAT_A01, MAT_A, MAT_COUNT };
enum { SLOT_AP = 0, SLOT_MP, SLOT_RP, SLOT_A00, SLOT_COUNT };
enum { KSP_PREONLY = 0, KSP_RICHARDSON, KSP_CHEBYSHEV, KSP_CG, KSP_CG_SR };
enum { PC_NONE = 0, PC_JACOBI };
enum { BRM1 = 1, BRM2, RBRM1, RBRM2 };

typedef struct {
  int64_t nrows, ncols, nnz;
  int32_t *rowptr, *col;
  double *val;
  double *dinv;        /* reciprocal diagonal (square operators)   */
  int64_t *src;        /* position in the caller's monolithic vals */
  int set;
} csr_t;

enum { PC_MG = 2 };
#define MG_MAX_LEVELS 16

typedef struct {
  int ksp, pc, max_it;
  double rtol, emin, emax;
  int last_its;
  /* [ext PETSc] PCMG: geometric V-cycle data, level 0 = coarsest */
  int mg_levels, nu_pre, nu_post;
  csr_t mgA[MG_MAX_LEVELS];   /* operator (level 0: explicit inverse);
                                 finest level: unused, the slot's operator */
  csr_t mgP[MG_MAX_LEVELS];   /* prolongation level-1 -> level */
  csr_t mgR[MG_MAX_LEVELS];   /* its transpose (restriction), explicit */
  double mg_emin[MG_MAX_LEVELS], mg_emax[MG_MAX_LEVELS];
} inner_t;

typedef struct pcdo_s {
  int variant;
  csr_t mat[MAT_COUNT];
  inner_t inner[SLOT_COUNT];
  int64_t n_bc;
  int32_t *bc_idx;
  double *bc_val;
  int64_t n_u, n_p;
  int32_t *perm;       /* split position -> caller's index */
  int64_t sys_nnz;
  int ready;
  double *w[8];        /* pressure-sized work vectors */
  double *wu[4];       /* velocity-sized work vectors */
  double *ws[2];       /* system-sized work vectors   */
  long num_pcd, num_fs;
  int gmres_its;
  double gmres_rnorm;
} pcdo_t;

static char g_err[512] = "";
static int fail(int code, const char *msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}
const char *pcdo_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ CSR */
static void csr_free(csr_t *m) {
  free(m->rowptr); free(m->col); free(m->val); free(m->dinv); free(m->src);
  memset(m, 0, sizeof *m);
}

static void csr_diag(csr_t *m) {
  if (m->nrows != m->ncols) return;
  if (!m->dinv) m->dinv = (double *)malloc(sizeof(double) * m->nrows);
  for (int64_t i = 0; i < m->nrows; ++i) {
    double d = 0.0;
    for (int32_t k = m->rowptr[i]; k < m->rowptr[i + 1]; ++k)
      if (m->col[k] == i) d += m->val[k];
    m->dinv[i] = (d != 0.0) ? 1.0 / d : 1.0;
  }
}

/* Mat.mult: y = A x (preconditioners.py:131,164) */
static void spmv(const csr_t *A, const double *x, double *y) {
  OMP_FOR
  for (int64_t i = 0; i < A->nrows; ++i) {
    double s = 0.0;
    for (int32_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k)
      s += A->val[k] * x[A->col[k]];
    y[i] = s;
  }
}

static double dot(int64_t n, const double *a, const double *b) {
  double s = 0.0;
  OMP_SUM
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* ----------------------------------------------------------- inner KSPs */
static void pc_apply(const csr_t *A, int pc, const double *r, double *z) {
  if (pc == PC_JACOBI)
    OMP_FOR
    for (int64_t i = 0; i < A->nrows; ++i) z[i] = A->dinv[i] * r[i];
  else
    memcpy(z, r, sizeof(double) * A->nrows);
}

/* [ext PETSc] KSPCG, PCJACOBI, zero initial guess, natural norm sqrt(r.z):
 * stop at max_it, or when rtol > 0 and sqrt(r.z) <= rtol*sqrt(r0.z0). */
static int solve_cg(const csr_t *A, inner_t *s, const double *b, double *x,
                    double *r, double *z, double *p, double *q) {
  int64_t n = A->nrows;
  memset(x, 0, sizeof(double) * n);
  memcpy(r, b, sizeof(double) * n);
  pc_apply(A, s->pc, r, z);
  double rz = dot(n, r, z), rz0 = rz, rz_old = 0.0;
  int it = 0;
  while (it < s->max_it) {
    if (rz == 0.0) break;
    if (s->rtol > 0.0 && sqrt(fabs(rz)) <= s->rtol * sqrt(fabs(rz0))) break;
    if (it == 0)
      memcpy(p, z, sizeof(double) * n);
    else {
      double beta = rz / rz_old;
      OMP_FOR
      for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    }
    spmv(A, p, q);
    double pq = dot(n, p, q);
    double alpha = rz / pq;
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) x[i] += alpha * p[i];
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) r[i] -= alpha * q[i];
    pc_apply(A, s->pc, r, z);
    rz_old = rz;
    rz = dot(n, r, z);
    ++it;
  }
  s->last_its = it;
  return 0;
}

/* [ext PETSc] KSPCG with -ksp_cg_single_reduction (KSPCGUseSingleReduction;
 * the Chronopoulos-Gear recurrence of KSPSolve_CG): the two inner products of
 * an iteration, beta = (z, r) and delta = (z, A z), are formed together - ONE
 * all-reduce of two numbers per iteration on several ranks instead of two of
 * one - and p.Ap follows from the recurrence
 *     dpi = delta - beta^2 dpi_old / beta_old^2.
 * Same stopping rule as solve_cg (natural norm sqrt|beta|, zero guess). */
static int solve_cg_sr(const csr_t *A, inner_t *s, const double *b, double *x,
                       double *r, double *z, double *p, double *sv) {
  int64_t n = A->nrows;
  double *w = (double *)malloc(sizeof(double) * n);
  memset(x, 0, sizeof(double) * n);
  memcpy(r, b, sizeof(double) * n);
  pc_apply(A, s->pc, r, z);
  spmv(A, z, sv);                                  /* s = A z */
  double beta = dot(n, z, r), delta = dot(n, z, sv);
  double beta0 = beta, betaold = 0.0, dpi = 0.0, dpiold = 0.0;
  int it = 0;
  while (it < s->max_it) {
    if (beta == 0.0) break;
    if (s->rtol > 0.0 && sqrt(fabs(beta)) <= s->rtol * sqrt(fabs(beta0))) break;
    if (it == 0) {
      memcpy(p, z, sizeof(double) * n);
      memcpy(w, sv, sizeof(double) * n);
      dpi = delta;
    } else {
      double bb = beta / betaold;
      for (int64_t i = 0; i < n; ++i) p[i] = z[i] + bb * p[i];
      for (int64_t i = 0; i < n; ++i) w[i] = sv[i] + bb * w[i];
      dpi = delta - beta * beta * dpiold / (betaold * betaold);
    }
    dpiold = dpi; betaold = beta;
    double a = beta / dpi;
    for (int64_t i = 0; i < n; ++i) x[i] += a * p[i];
    for (int64_t i = 0; i < n; ++i) r[i] -= a * w[i];
    pc_apply(A, s->pc, r, z);
    spmv(A, z, sv);
    beta = dot(n, z, r); delta = dot(n, z, sv);
    ++it;
  }
  s->last_its = it;
  free(w);
  return 0;
}

/* [ext PETSc] KSPCHEBYSHEV (cheby.c), eigenvalue bounds [emin, emax] of the
 * preconditioned operator given by the user
 * (-ksp_chebyshev_eigenvalues, demo_navier-stokes-pcd.py:163), zero initial
 * guess, no norms: one initial preconditioner application, then max_it
 * three-term updates
 *   p_{k+1} = (1-omega) p_{k-1} + omega p_k + omega*scale * B (b - A p_k). */
static int solve_cheb(const csr_t *A, inner_t *s, const double *b, double *x,
                      double *pa, double *pb, double *pc_, double *r) {
  int64_t n = A->nrows;
  double scale = 2.0 / (s->emax + s->emin);
  double alpha = 1.0 - scale * s->emin;
  double mu = 1.0 / alpha, omegaprod = 2.0 / alpha;
  double c_km1 = 1.0, c_k = mu;
  double *pkm1 = pa, *pk = pb, *pkp1 = pc_;
  memset(pkm1, 0, sizeof(double) * n);
  pc_apply(A, s->pc, b, pk);
  OMP_FOR
  for (int64_t i = 0; i < n; ++i) pk[i] = scale * pk[i] + pkm1[i];
  for (int it = 0; it < s->max_it; ++it) {
    double c_kp1 = 2.0 * mu * c_k - c_km1;
    double omega = omegaprod * c_k / c_kp1;
    spmv(A, pk, r);
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) r[i] = b[i] - r[i];
    pc_apply(A, s->pc, r, pkp1);
    OMP_FOR
    for (int64_t i = 0; i < n; ++i)
      pkp1[i] = (1.0 - omega) * pkm1[i] + omega * pk[i]
                + omega * scale * pkp1[i];
    double *t = pkm1; pkm1 = pk; pk = pkp1; pkp1 = t;
    c_km1 = c_k; c_k = c_kp1;
  }
  memcpy(x, pk, sizeof(double) * n);
  s->last_its = s->max_it;
  return 0;
}

/* [ext PETSc] KSPRICHARDSON, scale 1, zero guess: x += B (b - A x) */
static int solve_rich(const csr_t *A, inner_t *s, const double *b, double *x,
                      double *r, double *z) {
  int64_t n = A->nrows;
  memset(x, 0, sizeof(double) * n);
  for (int it = 0; it < s->max_it; ++it) {
    if (it == 0)
      memcpy(r, b, sizeof(double) * n);
    else {
      spmv(A, x, r);
      OMP_FOR
      for (int64_t i = 0; i < n; ++i) r[i] = b[i] - r[i];
    }
    pc_apply(A, s->pc, r, z);
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) x[i] += z[i];
  }
  s->last_its = s->max_it;
  return 0;
}

/* ---- [ext PETSc] PCMG: multiplicative V-cycle, Chebyshev/Jacobi smoothing,
 * explicit coarse inverse.  Same recurrences as solve_cheb; a smoother call
 * with a nonzero guess starts with p_1 = x + scale*B(b - A x). */
static void csr_transpose(csr_t *T, const csr_t *P) {
  csr_free(T);
  T->nrows = P->ncols; T->ncols = P->nrows; T->nnz = P->nnz;
  T->rowptr = (int32_t *)calloc(T->nrows + 1, sizeof(int32_t));
  T->col = (int32_t *)malloc(sizeof(int32_t) * (T->nnz ? T->nnz : 1));
  T->val = (double *)malloc(sizeof(double) * (T->nnz ? T->nnz : 1));
  for (int64_t k = 0; k < P->nnz; ++k) ++T->rowptr[P->col[k] + 1];
  for (int64_t c = 0; c < T->nrows; ++c) T->rowptr[c + 1] += T->rowptr[c];
  int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (T->nrows ? T->nrows : 1));
  memcpy(fill, T->rowptr, sizeof(int32_t) * T->nrows);
  for (int64_t i = 0; i < P->nrows; ++i)      /* rows ascending: each row of */
    for (int32_t k = P->rowptr[i]; k < P->rowptr[i + 1]; ++k) { /* T sorted */
      int32_t q = fill[P->col[k]]++;
      T->col[q] = (int32_t)i; T->val[q] = P->val[k];
    }
  free(fill);
  T->set = 1;
}

static void mg_smooth(const csr_t *A, double emin, double emax, int nu,
                      const double *b, double *x, int zero_guess) {
  int64_t n = A->nrows;
  if (nu == 0) { if (zero_guess) memset(x, 0, sizeof(double) * n); return; }
  double *base = (double *)malloc(sizeof(double) * n * 4);
  double *pkm1 = base, *pk = pkm1 + n, *pkp1 = pk + n, *r = pkp1 + n;
  double scale = 2.0 / (emax + emin), alpha = 1.0 - scale * emin;
  double mu = 1.0 / alpha, omegaprod = 2.0 / alpha, c_km1 = 1.0, c_k = mu;
  if (zero_guess) {
    memset(pkm1, 0, sizeof(double) * n);
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) pk[i] = scale * (A->dinv[i] * b[i]);
  } else {
    memcpy(pkm1, x, sizeof(double) * n);
    spmv(A, x, r);
    OMP_FOR
    for (int64_t i = 0; i < n; ++i)
      pk[i] = x[i] + scale * (A->dinv[i] * (b[i] - r[i]));
  }
  for (int it = 0; it < nu - 1; ++it) {
    double c_kp1 = 2.0 * mu * c_k - c_km1, omega = omegaprod * c_k / c_kp1;
    spmv(A, pk, r);
    OMP_FOR
    for (int64_t i = 0; i < n; ++i)
      pkp1[i] = (1.0 - omega) * pkm1[i] + omega * pk[i]
                + omega * scale * (A->dinv[i] * (b[i] - r[i]));
    double *t = pkm1; pkm1 = pk; pk = pkp1; pkp1 = t;
    c_km1 = c_k; c_k = c_kp1;
  }
  memcpy(x, pk, sizeof(double) * n);
  free(base);
}

static void mg_vcycle(const inner_t *s, const csr_t *Afine, int l,
                      const double *b, double *x) {
  if (l == 0) { spmv(&s->mgA[0], b, x); return; }
  const csr_t *A = (l == s->mg_levels - 1) ? Afine : &s->mgA[l];
  const csr_t *P = &s->mgP[l];
  int64_t n = A->nrows, nc = P->ncols;
  double *r = (double *)malloc(sizeof(double) * (n + 2 * nc));
  double *bc = r + n, *ec = bc + nc;
  mg_smooth(A, s->mg_emin[l], s->mg_emax[l], s->nu_pre, b, x, 1);
  if (s->nu_pre) {
    spmv(A, x, r);
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) r[i] = b[i] - r[i];
  } else memcpy(r, b, sizeof(double) * n);
  spmv(&s->mgR[l], r, bc);                 /* restrict: bc = P^T r */
  mg_vcycle(s, Afine, l - 1, bc, ec);
  spmv(P, ec, r);
  OMP_FOR
  for (int64_t i = 0; i < n; ++i) x[i] += r[i];
  mg_smooth(A, s->mg_emin[l], s->mg_emax[l], s->nu_post, b, x, 0);
  free(r);
}

/* KSPRICHARDSON / KSPPREONLY around the V-cycle */
static int solve_mg(const csr_t *A, inner_t *s, const double *b, double *x,
                    double *r, double *z) {
  int64_t n = A->nrows;
  if (s->mg_levels < 1) return fail(4, "pc mg: no hierarchy set");
  int its = (s->ksp == KSP_PREONLY) ? 1 : s->max_it;
  memset(x, 0, sizeof(double) * n);
  for (int it = 0; it < its; ++it) {
    if (it == 0) memcpy(r, b, sizeof(double) * n);
    else { spmv(A, x, r); for (int64_t i = 0; i < n; ++i) r[i] = b[i] - r[i]; }
    mg_vcycle(s, A, s->mg_levels - 1, r, z);
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) x[i] += z[i];
  }
  s->last_its = its;
  return 0;
}

static const int slot_mat[SLOT_COUNT] = {MAT_AP, MAT_MP, MAT_RP, MAT_A00};

/* KSP.solve(b, x); b and x must not alias */
static int inner_solve(pcdo_t *h, int slot, const double *b, double *x) {
  csr_t *A = &h->mat[slot_mat[slot]];
  inner_t *s = &h->inner[slot];
  if (!A->set) return fail(4, "inner_solve: operator not set");
  int64_t n = A->nrows;
  double *t0 = (double *)malloc(sizeof(double) * n * 4);
  double *t1 = t0 + n, *t2 = t1 + n, *t3 = t2 + n;
  int rc = 0;
  if (s->pc == PC_MG) {
    if (s->ksp != KSP_PREONLY && s->ksp != KSP_RICHARDSON)
      rc = fail(1, "pc mg is supported under preonly / richardson only");
    else rc = solve_mg(A, s, b, x, t0, t1);
    free(t0);
    return rc;
  }
  switch (s->ksp) {
    case KSP_PREONLY: pc_apply(A, s->pc, b, x); s->last_its = 1; break;
    case KSP_RICHARDSON: rc = solve_rich(A, s, b, x, t0, t1); break;
    case KSP_CHEBYSHEV: rc = solve_cheb(A, s, b, x, t0, t1, t2, t3); break;
    case KSP_CG: rc = solve_cg(A, s, b, x, t0, t1, t2, t3); break;
    case KSP_CG_SR: rc = solve_cg_sr(A, s, b, x, t0, t1, t2, t3); break;
    default: rc = fail(1, "unknown ksp type");
  }
  free(t0);
  return rc;
}

/* --------------------------------------------------------------- public */
int pcdo_create(pcdo_t **out, int variant, int device) {
  (void)device;
  if (variant < BRM1 || variant > RBRM2) return fail(1, "bad variant");
  pcdo_t *h = (pcdo_t *)calloc(1, sizeof *h);
  h->variant = variant;
  for (int s = 0; s < SLOT_COUNT; ++s) {
    /* reference default is an exact factorisation (preconditioners.py:42-49),
     * which has no counterpart here: default to tightly converged PCG */
    h->inner[s].ksp = KSP_CG; h->inner[s].pc = PC_JACOBI;
    h->inner[s].max_it = 10000; h->inner[s].rtol = 1e-12;
    h->inner[s].emin = 0.5; h->inner[s].emax = 2.0;
  }
  *out = h;
  return 0;
}

int pcdo_destroy(pcdo_t *h) {
  if (!h) return 0;
  for (int m = 0; m < MAT_COUNT; ++m) csr_free(&h->mat[m]);
  for (int s = 0; s < SLOT_COUNT; ++s)
    for (int l = 0; l < MG_MAX_LEVELS; ++l) {
      csr_free(&h->inner[s].mgA[l]); csr_free(&h->inner[s].mgP[l]);
      csr_free(&h->inner[s].mgR[l]);
    }
  free(h->bc_idx); free(h->bc_val); free(h->perm);
  for (int i = 0; i < 8; ++i) free(h->w[i]);
  for (int i = 0; i < 4; ++i) free(h->wu[i]);
  for (int i = 0; i < 2; ++i) free(h->ws[i]);
  free(h);
  return 0;
}

static int csr_store(csr_t *m, int64_t nrows, int64_t ncols,
                     const int32_t *rowptr, const int32_t *col,
                     const double *val) {
  csr_free(m);
  int64_t nnz = rowptr[nrows];
  m->nrows = nrows; m->ncols = ncols; m->nnz = nnz;
  m->rowptr = (int32_t *)malloc(sizeof(int32_t) * (nrows + 1));
  m->col = (int32_t *)malloc(sizeof(int32_t) * (nnz ? nnz : 1));
  m->val = (double *)malloc(sizeof(double) * (nnz ? nnz : 1));
  memcpy(m->rowptr, rowptr, sizeof(int32_t) * (nrows + 1));
  memcpy(m->col, col, sizeof(int32_t) * nnz);
  if (val) memcpy(m->val, val, sizeof(double) * nnz);
  m->set = 1;
  return 0;
}

int pcdo_set_csr(pcdo_t *h, int which, int64_t nrows, int64_t ncols,
                 const int32_t *rowptr, const int32_t *col,
                 const double *val) {
  if (which < 0 || which >= MAT_A) return fail(1, "set_csr: bad operator");
  csr_store(&h->mat[which], nrows, ncols, rowptr, col, val);
  csr_diag(&h->mat[which]);
  h->ready = 0;
  return 0;
}

int pcdo_update_values(pcdo_t *h, int which, const double *val, int mem) {
  (void)mem;
  if (which < 0 || which >= MAT_A || !h->mat[which].set)
    return fail(4, "update_values: operator not set");
  memcpy(h->mat[which].val, val, sizeof(double) * h->mat[which].nnz);
  csr_diag(&h->mat[which]);
  return 0;
}

/* [ext PETSc] MatCreateSubMatrix on (is_r, is_c) plus value provenance */
static void extract_block(csr_t *out, int64_t n, const int32_t *rowptr,
                          const int32_t *col, const int32_t *rows,
                          int64_t nr, const int32_t *colmap /* caller->local
                          or -1 */, int64_t nc) {
  csr_free(out);
  out->nrows = nr; out->ncols = nc;
  out->rowptr = (int32_t *)malloc(sizeof(int32_t) * (nr + 1));
  int64_t nnz = 0;
  for (int64_t i = 0; i < nr; ++i)
    for (int32_t k = rowptr[rows[i]]; k < rowptr[rows[i] + 1]; ++k)
      if (colmap[col[k]] >= 0) ++nnz;
  out->nnz = nnz;
  out->col = (int32_t *)malloc(sizeof(int32_t) * (nnz ? nnz : 1));
  out->val = (double *)malloc(sizeof(double) * (nnz ? nnz : 1));
  out->src = (int64_t *)malloc(sizeof(int64_t) * (nnz ? nnz : 1));
  int64_t p = 0;
  for (int64_t i = 0; i < nr; ++i) {
    out->rowptr[i] = (int32_t)p;
    int64_t start = p;
    for (int32_t k = rowptr[rows[i]]; k < rowptr[rows[i] + 1]; ++k) {
      int32_t c = colmap[col[k]];
      if (c < 0) continue;
      /* insertion sort by local column keeps rows sorted */
      int64_t q = p;
      while (q > start && out->col[q - 1] > c) {
        out->col[q] = out->col[q - 1]; out->src[q] = out->src[q - 1]; --q;
      }
      out->col[q] = c; out->src[q] = k; ++p;
    }
  }
  out->rowptr[nr] = (int32_t)p;
  out->set = 1;
  (void)n;
}

static void block_values(csr_t *m, const double *vals) {
  for (int64_t k = 0; k < m->nnz; ++k) m->val[k] = vals[m->src[k]];
}

int pcdo_update_system(pcdo_t *h, const double *vals, const double *pvals,
                       int mem) {
  (void)mem;
  if (!h->mat[MAT_A].set) return fail(4, "update_system: no system set");
  block_values(&h->mat[MAT_A], vals);
  block_values(&h->mat[MAT_A00], pvals ? pvals : vals);
  block_values(&h->mat[MAT_A01], pvals ? pvals : vals);
  csr_diag(&h->mat[MAT_A00]);
  return 0;
}

int pcdo_set_system(pcdo_t *h, int64_t n, const int32_t *rowptr,
                    const int32_t *col, const double *vals,
                    const double *pvals, int64_t n_u, const int32_t *is_u,
                    int64_t n_p, const int32_t *is_p) {
  if (n_u + n_p != n) return fail(1, "set_system: n_u + n_p != n");
  h->n_u = n_u; h->n_p = n_p; h->sys_nnz = rowptr[n];
  free(h->perm);
  h->perm = (int32_t *)malloc(sizeof(int32_t) * n);
  memcpy(h->perm, is_u, sizeof(int32_t) * n_u);
  memcpy(h->perm + n_u, is_p, sizeof(int32_t) * n_p);
  int32_t *mu = (int32_t *)malloc(sizeof(int32_t) * n);
  int32_t *mp = (int32_t *)malloc(sizeof(int32_t) * n);
  int32_t *ma = (int32_t *)malloc(sizeof(int32_t) * n);
  for (int64_t i = 0; i < n; ++i) mu[i] = mp[i] = ma[i] = -1;
  for (int64_t i = 0; i < n_u; ++i) { mu[is_u[i]] = (int32_t)i; }
  for (int64_t i = 0; i < n_p; ++i) { mp[is_p[i]] = (int32_t)i; }
  for (int64_t i = 0; i < n; ++i) ma[h->perm[i]] = (int32_t)i;
  for (int64_t i = 0; i < n; ++i)
    if (ma[i] < 0) { free(mu); free(mp); free(ma);
      return fail(1, "set_system: index sets do not cover 0..n-1"); }
  extract_block(&h->mat[MAT_A00], n, rowptr, col, is_u, n_u, mu, n_u);
  extract_block(&h->mat[MAT_A01], n, rowptr, col, is_u, n_u, mp, n_p);
  extract_block(&h->mat[MAT_A], n, rowptr, col, h->perm, n, ma, n);
  free(mu); free(mp); free(ma);
  h->ready = 0;
  return pcdo_update_system(h, vals, pvals, 0);
}

int pcdo_set_bc(pcdo_t *h, int64_t n_bc, const int32_t *idx,
                const double *vals) {
  free(h->bc_idx); free(h->bc_val);
  h->n_bc = n_bc;
  h->bc_idx = (int32_t *)malloc(sizeof(int32_t) * (n_bc ? n_bc : 1));
  h->bc_val = (double *)malloc(sizeof(double) * (n_bc ? n_bc : 1));
  memcpy(h->bc_idx, idx, sizeof(int32_t) * n_bc);
  memcpy(h->bc_val, vals, sizeof(double) * n_bc);
  return 0;
}

int pcdo_set_inner(pcdo_t *h, int slot, int ksp, int pc, int max_it,
                   double rtol, double emin, double emax) {
  if (slot < 0 || slot >= SLOT_COUNT) return fail(1, "set_inner: bad slot");
  if (ksp < KSP_PREONLY || ksp > KSP_CG_SR) return fail(1, "set_inner: bad ksp");
  if (pc != PC_NONE && pc != PC_JACOBI && pc != PC_MG)
    return fail(1, "set_inner: bad pc");
  if (ksp == KSP_CHEBYSHEV && !(emax > emin && emin > 0.0))
    return fail(1, "set_inner: chebyshev needs 0 < emin < emax");
  inner_t *s = &h->inner[slot];
  s->ksp = ksp; s->pc = pc; s->max_it = max_it; s->rtol = rtol;
  s->emin = emin; s->emax = emax;
  return 0;
}

/* [ext PETSc] PCMGSetLevels / PCMGSetInterpolation / per-level operators */
int pcdo_mg_begin(pcdo_t *h, int slot, int nlevels, int nu_pre, int nu_post) {
  if (slot < 0 || slot >= SLOT_COUNT) return fail(1, "mg_begin: bad slot");
  if (nlevels < 1 || nlevels > MG_MAX_LEVELS || nu_pre < 0 || nu_post < 0)
    return fail(1, "mg_begin: bad level / smoothing counts");
  inner_t *s = &h->inner[slot];
  for (int l = 0; l < MG_MAX_LEVELS; ++l) {
    csr_free(&s->mgA[l]); csr_free(&s->mgP[l]); csr_free(&s->mgR[l]);
  }
  s->mg_levels = nlevels; s->nu_pre = nu_pre; s->nu_post = nu_post;
  return 0;
}

int pcdo_mg_set_level(pcdo_t *h, int slot, int level, int64_t n,
                      const int32_t *rowptr, const int32_t *col,
                      const double *val, int64_t p_rows, int64_t p_cols,
                      const int32_t *prowptr, const int32_t *pcol,
                      const double *pval, double emin, double emax) {
  if (slot < 0 || slot >= SLOT_COUNT) return fail(1, "mg_set_level: bad slot");
  inner_t *s = &h->inner[slot];
  if (level < 0 || level >= s->mg_levels)
    return fail(1, "mg_set_level: level out of range");
  if (level > 0 && !(emax > emin && emin > 0.0))
    return fail(1, "mg_set_level: smoother needs 0 < emin < emax");
  if (rowptr) {
    csr_store(&s->mgA[level], n, n, rowptr, col, val);
    csr_diag(&s->mgA[level]);
  } else if (level != s->mg_levels - 1)
    return fail(1, "mg_set_level: coarse levels need an operator");
  if (level > 0) {
    if (!prowptr) return fail(1, "mg_set_level: prolongation missing");
    csr_store(&s->mgP[level], p_rows, p_cols, prowptr, pcol, pval);
    csr_transpose(&s->mgR[level], &s->mgP[level]);
  }
  s->mg_emin[level] = emin; s->mg_emax[level] = emax;
  return 0;
}

int pcdo_mg_update_values(pcdo_t *h, int slot, int level, const double *val,
                          double emin, double emax, int mem) {
  (void)mem;
  if (slot < 0 || slot >= SLOT_COUNT) return fail(1, "mg_update: bad slot");
  inner_t *s = &h->inner[slot];
  if (level < 0 || level >= s->mg_levels)
    return fail(4, "mg_update_values: level not set");
  if (val) {
    if (!s->mgA[level].set) return fail(4, "mg_update_values: level not set");
    memcpy(s->mgA[level].val, val, sizeof(double) * s->mgA[level].nnz);
    csr_diag(&s->mgA[level]);
  }
  if (level > 0) { s->mg_emin[level] = emin; s->mg_emax[level] = emax; }
  return 0;
}

int pcdo_setup(pcdo_t *h) {
  if (!h->mat[MAT_AP].set || !h->mat[MAT_MP].set || !h->mat[MAT_KP].set)
    return fail(4, "setup: Ap, Mp and Kp are required");
  if (h->variant >= RBRM1 && !h->mat[MAT_RP].set)
    return fail(4, "setup: PCDR variants require Rp");
  int64_t np = h->mat[MAT_AP].nrows;
  if (h->n_p && h->n_p != np) return fail(1, "setup: n_p mismatch");
  h->n_p = np;
  for (int i = 0; i < 8; ++i) {
    free(h->w[i]); h->w[i] = (double *)calloc(np ? np : 1, sizeof(double));
  }
  if (h->mat[MAT_A00].set) {
    int64_t nu = h->mat[MAT_A00].nrows;
    h->n_u = nu;
    for (int i = 0; i < 4; ++i) {
      free(h->wu[i]); h->wu[i] = (double *)calloc(nu ? nu : 1, sizeof(double));
    }
    for (int i = 0; i < 2; ++i) {
      free(h->ws[i]);
      h->ws[i] = (double *)calloc(nu + np ? nu + np : 1, sizeof(double));
    }
  }
  h->ready = 1;
  return 0;
}

/* SubfieldBC::apply_subfield_bc: VecSetValues(..., INSERT_VALUES) */
static void apply_bc(const pcdo_t *h, double *x) {
  for (int64_t i = 0; i < h->n_bc; ++i) x[h->bc_idx[i]] = h->bc_val[i];
}

int pcdo_apply_bc(pcdo_t *h, double *x, int mem) {
  (void)mem; apply_bc(h, x); return 0;
}

/* The four apply bodies.  Comments quote the reference line being restated. */
static int pcd_apply_core(pcdo_t *h, const double *x, double *y) {
  int64_t n = h->n_p;
  int rc = 0;
  if (h->variant == BRM1 || h->variant == RBRM1) {
    double *z = h->w[0];                            /* get_work_vecs(x, 1) */
    memcpy(z, x, sizeof(double) * n);               /* x.copy(result=z)    */
    apply_bc(h, z);                                 /* bcs_applier(z)      */
    if ((rc = inner_solve(h, SLOT_AP, z, y))) return rc; /* y = Ap^-1 z   */
    spmv(&h->mat[MAT_KP], y, z);                    /* z = Kp y            */
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) z[i] += 1.0 * x[i]; /* z.axpy(1, x)   */
    if ((rc = inner_solve(h, SLOT_MP, z, y))) return rc; /* y = Mp^-1 z   */
    if (h->variant == RBRM1) {
      if ((rc = inner_solve(h, SLOT_RP, x, z))) return rc; /* z = Rp^-1 x */
      OMP_FOR
      for (int64_t i = 0; i < n; ++i) y[i] += 1.0 * z[i];  /* y.axpy(1,z) */
    }
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) y[i] *= -1.0;   /* y.scale(-1)         */
  } else {
    double *z0 = h->w[0], *z1 = h->w[1];            /* get_work_vecs(x, 2) */
    if ((rc = inner_solve(h, SLOT_MP, x, y))) return rc; /* y = Mp^-1 x   */
    memcpy(z0, y, sizeof(double) * n);              /* y.copy(result=z0)   */
    spmv(&h->mat[MAT_KP], z0, z1);                  /* z1 = Kp z0          */
    apply_bc(h, z1);                                /* bcs_applier(z1)     */
    if ((rc = inner_solve(h, SLOT_AP, z1, z0))) return rc; /* z0=Ap^-1 z1 */
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) y[i] += 1.0 * z0[i]; /* y.axpy(1,z0)  */
    if (h->variant == RBRM2) {
      if ((rc = inner_solve(h, SLOT_RP, x, z0))) return rc; /* z0=Rp^-1 x */
      OMP_FOR
      for (int64_t i = 0; i < n; ++i) y[i] += 1.0 * z0[i];
    }
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) y[i] *= -1.0;   /* y.scale(-1)         */
  }
  ++h->num_pcd;
  return 0;
}

int pcdo_apply(pcdo_t *h, const double *x, double *y, int mem) {
  (void)mem;
  if (!h->ready) return fail(4, "apply: call setup first");
  return pcd_apply_core(h, x, y);
}

/* [ext PETSc] PCApply_FieldSplit_Schur, UPPER, on split-ordered vectors */
static int fs_apply_split(pcdo_t *h, const double *x, double *y) {
  int64_t nu = h->n_u;
  const double *xu = x, *xp = x + nu;
  double *yu = y, *yp = y + nu, *t = h->wu[0];
  int rc;
  if ((rc = pcd_apply_core(h, xp, yp))) return rc;      /* y_p = S^-1 x_p */
  spmv(&h->mat[MAT_A01], yp, t);
  OMP_FOR
  for (int64_t i = 0; i < nu; ++i) t[i] = xu[i] - t[i]; /* x_u - A01 y_p  */
  if ((rc = inner_solve(h, SLOT_A00, t, yu))) return rc; /* A00^-1 (...)  */
  ++h->num_fs;
  return 0;
}

int pcdo_fieldsplit_apply(pcdo_t *h, const double *x, double *y, int mem) {
  (void)mem;
  if (!h->ready || !h->mat[MAT_A00].set)
    return fail(4, "fieldsplit_apply: system/setup missing");
  int64_t n = h->n_u + h->n_p;
  double *xs = h->ws[0], *ys = h->ws[1];
  OMP_FOR
  for (int64_t i = 0; i < n; ++i) xs[i] = x[h->perm[i]];
  int rc = fs_apply_split(h, xs, ys);
  if (rc) return rc;
  OMP_FOR
  for (int64_t i = 0; i < n; ++i) y[h->perm[i]] = ys[i];
  return 0;
}

/* [ext PETSc] KSPGMRES: restart m, right preconditioning, classical
 * Gram-Schmidt without refinement, zero initial guess, true-residual test
 * ||r|| <= max(rtol*||b||, atol) on the recurrence estimate. */
int pcdo_gmres_solve(pcdo_t *h, const double *b, double *x, int mem,
                     double rtol, double atol, int m, int max_it, int *its,
                     double *rnorm) {
  (void)mem;
  if (!h->ready || !h->mat[MAT_A].set)
    return fail(4, "gmres_solve: system/setup missing");
  int64_t n = h->n_u + h->n_p;
  double *V = (double *)malloc(sizeof(double) * n * (m + 1));
  double *H = (double *)calloc((size_t)(m + 1) * m, sizeof(double));
  double *cs = (double *)calloc(m, sizeof(double));
  double *sn = (double *)calloc(m, sizeof(double));
  double *g = (double *)calloc(m + 1, sizeof(double));
  double *yk = (double *)calloc(m, sizeof(double));
  double *xs = (double *)calloc(n, sizeof(double));
  double *bs = (double *)malloc(sizeof(double) * n);
  double *z = (double *)malloc(sizeof(double) * n);
  double *w = (double *)malloc(sizeof(double) * n);
  OMP_FOR
  for (int64_t i = 0; i < n; ++i) bs[i] = b[h->perm[i]];
  double bnorm = sqrt(dot(n, bs, bs));
  double tol = rtol * bnorm; if (atol > tol) tol = atol;
  int it = 0, rc = 0; double res = bnorm;
  memcpy(w, bs, sizeof(double) * n);               /* r0 = b (x0 = 0) */
  while (it < max_it && res > tol) {
    double beta = sqrt(dot(n, w, w));
    res = beta;
    if (beta <= tol) break;
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) V[i] = w[i] / beta;
    memset(g, 0, sizeof(double) * (m + 1)); g[0] = beta;
    int k = 0;
    for (; k < m && it < max_it; ++k) {
      double *vk = V + (size_t)k * n, *vn = V + (size_t)(k + 1) * n;
      if ((rc = fs_apply_split(h, vk, z))) goto done;  /* z = M^-1 v_k */
      spmv(&h->mat[MAT_A], z, vn);                     /* w = A z      */
      double *hc = H + (size_t)k * (m + 1);
      for (int j = 0; j <= k; ++j) hc[j] = dot(n, V + (size_t)j * n, vn);
      for (int j = 0; j <= k; ++j) {
        const double *vj = V + (size_t)j * n; double hj = hc[j];
        OMP_FOR
        for (int64_t i = 0; i < n; ++i) vn[i] -= hj * vj[i];
      }
      double hn = sqrt(dot(n, vn, vn));
      hc[k + 1] = hn;
      if (hn != 0.0) for (int64_t i = 0; i < n; ++i) vn[i] /= hn;
      for (int j = 0; j < k; ++j) {                    /* old rotations */
        double t = cs[j] * hc[j] + sn[j] * hc[j + 1];
        hc[j + 1] = -sn[j] * hc[j] + cs[j] * hc[j + 1]; hc[j] = t;
      }
      double d = hypot(hc[k], hc[k + 1]);
      cs[k] = hc[k] / d; sn[k] = hc[k + 1] / d;
      hc[k] = d; hc[k + 1] = 0.0;
      g[k + 1] = -sn[k] * g[k]; g[k] = cs[k] * g[k];
      res = fabs(g[k + 1]);
      ++it;
      if (res <= tol || hn == 0.0) { ++k; break; }
    }
    /* y = H^-1 g; x += M^-1 (V y)  (KSPGMRESBuildSoln + unwind right PC) */
    for (int i = k - 1; i >= 0; --i) {
      double s = g[i];
      for (int j = i + 1; j < k; ++j) s -= H[(size_t)j * (m + 1) + i] * yk[j];
      yk[i] = s / H[(size_t)i * (m + 1) + i];
    }
    memset(w, 0, sizeof(double) * n);
    for (int j = 0; j < k; ++j) {
      const double *vj = V + (size_t)j * n;
      OMP_FOR
      for (int64_t i = 0; i < n; ++i) w[i] += yk[j] * vj[i];
    }
    if ((rc = fs_apply_split(h, w, z))) goto done;
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) xs[i] += z[i];
    if (res <= tol || it >= max_it) break;
    spmv(&h->mat[MAT_A], xs, z);                       /* restart: true r */
    OMP_FOR
    for (int64_t i = 0; i < n; ++i) w[i] = bs[i] - z[i];
  }
done:
  OMP_FOR
  for (int64_t i = 0; i < n; ++i) x[h->perm[i]] = xs[i];
  h->gmres_its = it; h->gmres_rnorm = res;
  if (its) *its = it;
  if (rnorm) *rnorm = res;
  free(V); free(H); free(cs); free(sn); free(g); free(yk); free(xs);
  free(bs); free(z); free(w);
  return rc;
}

int pcdo_spmv(pcdo_t *h, int which, const double *x, double *y, int mem) {
  (void)mem;
  if (which < 0 || which >= MAT_COUNT || !h->mat[which].set)
    return fail(4, "spmv: operator not set");
  spmv(&h->mat[which], x, y);
  return 0;
}

int pcdo_inner_solve(pcdo_t *h, int slot, const double *b, double *x,
                     int mem) {
  (void)mem;
  if (slot < 0 || slot >= SLOT_COUNT) return fail(1, "inner_solve: bad slot");
  return inner_solve(h, slot, b, x);
}

int pcdo_get_info(pcdo_t *h, int key, double *out) {
  switch (key) {
    case 0: *out = (double)h->n_u; break;
    case 1: *out = (double)h->n_p; break;
    case 2: case 3: case 4: case 5:
      *out = (double)h->inner[key - 2].last_its; break;
    case 6: *out = (double)h->num_pcd; break;
    case 7: *out = (double)h->num_fs; break;
    case 8: *out = (double)h->gmres_its; break;
    case 9: *out = h->gmres_rnorm; break;
    case 10: *out = (double)h->n_u; break;      /* one rank owns every row */
    case 11: *out = (double)h->n_p; break;
    case 12: case 13: case 14: *out = 0.0; break; /* no kernel paths, no communicator on the CPU */
    default:
      if (key >= 16 && key < 16 + MAT_COUNT) {
        *out = (double)h->mat[key - 16].nnz; break;
      }
      return fail(1, "get_info: bad key");
  }
  return 0;
}


/* ==========================================================================
 * TEAM build (make omp): the multi-core TIMING port behind bench.py's
 * cpu_baseline - never used for parity.  Same algorithm and the same data as
 * the serial functions above, organised the way a bandwidth-bound host wants
 * it ("credible CPU baseline"):
 *   - ONE parallel region per fieldsplit PCApply; inside it every loop is an
 *     orphaned `omp for schedule(static)` (implicit barrier), no fork/join
 *     per kernel;
 *   - a static row partition that is the same in every loop, and first-touch
 *     placement: pcdo_team_prepare() re-allocates every matrix and work
 *     vector and lets the thread that will stream a row block copy / zero it,
 *     so pages sit on the NUMA node of their reader (the thread count is
 *     frozen by prepare for that reason);
 *   - the same loop fusion the GPU kernels have: a Chebyshev step is one pass
 *     (SpMV + residual + Jacobi + three-term update), residuals and
 *     prolongation-adds are SpMV epilogues.
 * Checked against the serial path by tests/test_oracle.py (<= 1e-12).
 * ========================================================================== */
#ifdef _OPENMP
#define T_FOR _Pragma("omp for schedule(static)")
/* Loops of the team: one static partition per loop LENGTH.  Long loops (>=
 * t_big iterations) run on the whole team and end in the team's barrier; short
 * ones - the coarse multigrid levels, a few hundred to a few thousand rows -
 * run on the first t_sub threads only and end in a barrier of those threads:
 * libgomp's barrier is centralised, its cost grows with the team (measured,
 * round 3: 2.5 ms per PCApply on 16 threads, 12.8 ms on 128 - about 150
 * barriers per apply at 0.5 us per thread each), and a 7-row share per thread
 * is all overhead.  Before the next long loop (or `single`) the threads that
 * sat the short ones out meet the others in one full barrier (t_sync). */
static int t_big = 40000, t_sub = 8;
static __thread int t_dirty = 0;            /* short loops since the last full barrier */
static __thread int t_sense = 0;
static volatile int t_sb_count = 0, t_sb_sense = 0;

/* The whole team's barrier, NUMA-aware (round 6): a two-level sense-reversing
 * barrier in place of libgomp's centralised one.  Threads are pinned to cores
 * in order (OMP_PLACES=cores, bench.py), so t_grp consecutive thread ids share
 * a socket / an L3: they meet on a counter of their own cache line, the last
 * one of each group meets the other groups' last ones on the top counter, and
 * the release travels back the same way - per barrier every thread touches
 * one line that lives in its own L3, and only n_groups threads touch the line
 * that crosses the sockets (the centralised barrier: all of them, 150 times
 * per PCApply).  PCDO_TEAM_GROUP sets the group size; 0 (the default) = the
 * runtime's own barrier.  MEASURED on this pool's 2 x 64-core hosts
 * (bench.py cpu_baseline, "sweep_with_the_two_level_barrier", round 6): the
 * two-level form LOSES at every thread count where it differs - level 6, 32
 * threads: 196-212 PCApply/s against 243-273 with libgomp's barrier (whose
 * waiters spin on a generation word of its own cache line and fall back to a
 * futex) - also with count and release word on separate lines.  Kept as the
 * A/B switch; the collapse of the sweep past 32 threads is not the barrier's
 * centralisation. */
#define T_MAX_GROUPS 64
/* (arrivals and the release word on cache lines of their own: an arrival
 * must not invalidate the line the group's waiters spin on - the first form
 * of this barrier kept both in one line and LOST to the runtime's barrier,
 * 212 against 273 PCApply/s at 32 threads) */
typedef struct {
  volatile int count; char pad0[60];
  volatile int sense; char pad1[60];
} t_node_t;
static t_node_t t_grp_node[T_MAX_GROUPS] __attribute__((aligned(128)));
static t_node_t t_top_node __attribute__((aligned(128)));
static int t_grp = 0;          /* measured (round 6): the runtime's barrier wins - see below */
static __thread int t_hsense = 0;
static inline void t_full_barrier(void) {
  const int nth = omp_get_num_threads();
  if (t_grp <= 0 || nth <= t_grp || (nth + t_grp - 1) / t_grp > T_MAX_GROUPS) {
    _Pragma("omp barrier")
    return;
  }
  const int tid = omp_get_thread_num(), g = tid / t_grp, ng = (nth + t_grp - 1) / t_grp;
  const int gsize = (g + 1) * t_grp <= nth ? t_grp : nth - g * t_grp;
  t_hsense = !t_hsense;
  if (__atomic_add_fetch(&t_grp_node[g].count, 1, __ATOMIC_ACQ_REL) == gsize) {
    __atomic_store_n(&t_grp_node[g].count, 0, __ATOMIC_RELAXED);
    if (__atomic_add_fetch(&t_top_node.count, 1, __ATOMIC_ACQ_REL) == ng) {
      __atomic_store_n(&t_top_node.count, 0, __ATOMIC_RELAXED);
      __atomic_store_n(&t_top_node.sense, t_hsense, __ATOMIC_RELEASE);
    } else {
      while (__atomic_load_n(&t_top_node.sense, __ATOMIC_ACQUIRE) != t_hsense) __builtin_ia32_pause();
    }
    __atomic_store_n(&t_grp_node[g].sense, t_hsense, __ATOMIC_RELEASE);
  } else {
    while (__atomic_load_n(&t_grp_node[g].sense, __ATOMIC_ACQUIRE) != t_hsense) __builtin_ia32_pause();
  }
}

/* Every parallel region starts from a known barrier state: t_dirty / t_sense
 * live in the pool threads and would otherwise survive a region that ended on
 * a short loop (the next, larger team would then take t_sync()'s barrier with
 * some of its threads only).  t_region() runs on the encountering thread
 * before the fork, t_begin() is the first statement of every team thread. */
static inline void t_region(void) {
  __atomic_store_n(&t_sb_count, 0, __ATOMIC_RELAXED);
  __atomic_store_n(&t_sb_sense, 0, __ATOMIC_RELEASE);
  for (int g = 0; g < T_MAX_GROUPS; ++g) {
    __atomic_store_n(&t_grp_node[g].count, 0, __ATOMIC_RELAXED);
    __atomic_store_n(&t_grp_node[g].sense, 0, __ATOMIC_RELAXED);
  }
  __atomic_store_n(&t_top_node.count, 0, __ATOMIC_RELAXED);
  __atomic_store_n(&t_top_node.sense, 0, __ATOMIC_RELEASE);
}
static inline void t_begin(void) { t_dirty = 0; t_sense = 0; t_hsense = 0; }

static inline void t_sync(void) {
  if (t_dirty) {
    t_full_barrier();
    t_dirty = 0;
  }
}
static inline void t_sub_barrier(int k) {
  t_sense = !t_sense;
  if (__atomic_add_fetch(&t_sb_count, 1, __ATOMIC_ACQ_REL) == k) {
    __atomic_store_n(&t_sb_count, 0, __ATOMIC_RELAXED);
    __atomic_store_n(&t_sb_sense, t_sense, __ATOMIC_RELEASE);
  } else {
    while (__atomic_load_n(&t_sb_sense, __ATOMIC_ACQUIRE) != t_sense) __builtin_ia32_pause();
  }
}
static inline int t_width(int64_t n) {
  const int nth = omp_get_num_threads();
  return (n >= t_big || nth <= t_sub) ? nth : t_sub;
}
static inline int64_t t_lo(int64_t n) {
  const int nth = omp_get_num_threads(), tid = omp_get_thread_num(), k = t_width(n);
  if (k == nth) t_sync();
  return tid < k ? n * tid / k : 0;
}
static inline int64_t t_hi(int64_t n) {
  const int tid = omp_get_thread_num(), k = t_width(n);
  return tid < k ? n * (tid + 1) / k : 0;
}
static inline void t_end(int64_t n) {
  const int nth = omp_get_num_threads(), k = t_width(n);
  if (k == nth) {
    t_full_barrier();
  } else {
    if (omp_get_thread_num() < k) t_sub_barrier(k);
    t_dirty = 1;
  }
}
#define T_LOOP(i, n)                                                            \
  for (int64_t i##_n = (n), i##_lo = t_lo(i##_n), i##_hi = t_hi(i##_n), i##_go = 1; \
       i##_go; i##_go = 0, t_end(i##_n))                                        \
    for (int64_t i = i##_lo; i < i##_hi; ++i)

typedef struct {
  int levels, nu_pre, nu_post;
  csr_t A[MG_MAX_LEVELS], P[MG_MAX_LEVELS], R[MG_MAX_LEVELS];
  double emin[MG_MAX_LEVELS], emax[MG_MAX_LEVELS];
  double *x[MG_MAX_LEVELS], *t0[MG_MAX_LEVELS], *t1[MG_MAX_LEVELS],
         *r[MG_MAX_LEVELS], *b[MG_MAX_LEVELS];
} team_mg_t;

typedef struct {
  int threads;
  csr_t mat[MAT_COUNT];
  team_mg_t mg[SLOT_COUNT];
  double *cw[SLOT_COUNT][3];          /* Chebyshev ring per slot */
  double *w0, *w1, *wu, *xs, *ys;
  int32_t *perm;
  double pq, rz;                      /* shared reduction targets (CG) */
} team_t;

static team_t *g_team[64];
static pcdo_t *g_team_owner[64];

static void *t_alloc(size_t bytes) { return malloc(bytes ? bytes : 8); }

/* first-touch copy: the thread that owns a row block copies its rows */
static void t_copy_csr(csr_t *d, const csr_t *s, int threads) {
  memset(d, 0, sizeof *d);
  if (!s->set) return;
  d->nrows = s->nrows; d->ncols = s->ncols; d->nnz = s->nnz; d->set = 1;
  d->rowptr = (int32_t *)t_alloc(sizeof(int32_t) * (s->nrows + 1));
  d->col = (int32_t *)t_alloc(sizeof(int32_t) * s->nnz);
  d->val = (double *)t_alloc(sizeof(double) * s->nnz);
  if (s->dinv) d->dinv = (double *)t_alloc(sizeof(double) * s->nrows);
  t_region();
#pragma omp parallel num_threads(threads)
  {
    t_begin();
    T_LOOP(i, s->nrows) {
      d->rowptr[i] = s->rowptr[i];
      for (int32_t k = s->rowptr[i]; k < s->rowptr[i + 1]; ++k) {
        d->col[k] = s->col[k]; d->val[k] = s->val[k];
      }
      if (s->dinv) d->dinv[i] = s->dinv[i];
    }
  }
  d->rowptr[s->nrows] = s->rowptr[s->nrows];
}

static double *t_vec(int64_t n, int threads) {
  double *v = (double *)t_alloc(sizeof(double) * n);
  t_region();
#pragma omp parallel num_threads(threads)
  {
    t_begin();
    T_LOOP(i, n) v[i] = 0.0;
  }
  return v;
}

/* ---- team kernels: called by EVERY thread of the region ------------------ */
/* mode 0: y = A x; 1: y = add + A x; 2: y = add - A x */
static void t_spmv(const csr_t *A, const double *x, double *y, int mode,
                   const double *add) {
  T_LOOP(i, A->nrows) {
    double s = 0.0;
    for (int32_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k)
      s += A->val[k] * x[A->col[k]];
    y[i] = mode == 0 ? s : (mode == 1 ? add[i] + s : add[i] - s);
  }
}

/* pn = c0 pm + c1 pk + c2 dinv (b - A pk) in one pass */
static void t_cheb_step(const csr_t *A, const double *b, const double *pm,
                        const double *pk, double *pn, double c0, double c1,
                        double c2) {
  T_LOOP(i, A->nrows) {
    double s = 0.0;
    for (int32_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k)
      s += A->val[k] * pk[A->col[k]];
    double v = c1 * pk[i] + c2 * (A->dinv[i] * (b[i] - s));
    if (c0 != 0.0) v += c0 * pm[i];
    pn[i] = v;
  }
}

static void t_scale_dinv(const csr_t *A, const double *b, double s, double *x) {
  T_LOOP(i, A->nrows) x[i] = s * (A->dinv[i] * b[i]);
}

/* nu Chebyshev-Jacobi steps; iterates rotate through bufs; returns the
 * buffer that holds the result (mg_smooth of the serial path) */
static double *t_smooth(const csr_t *A, double emin, double emax, int nu,
                        const double *b, double *bufs[3], int zero_guess) {
  double scale = 2.0 / (emax + emin), alpha = 1.0 - scale * emin;
  double mu = 1.0 / alpha, omegaprod = 2.0 / alpha, c_km1 = 1.0, c_k = mu;
  int cur = 0, have_pm = 0;
  if (nu == 0) return bufs[0];
  if (zero_guess) {
    t_scale_dinv(A, b, scale, bufs[0]);
  } else {
    t_cheb_step(A, b, bufs[0], bufs[0], bufs[1], 0.0, 1.0, scale);
    cur = 1; have_pm = 1;
  }
  for (int it = 0; it < nu - 1; ++it) {
    double c_kp1 = 2.0 * mu * c_k - c_km1, omega = omegaprod * c_k / c_kp1;
    double *pk = bufs[cur % 3], *pn = bufs[(cur + 1) % 3];
    double *pm = have_pm ? bufs[(cur + 2) % 3] : pk;
    t_cheb_step(A, b, pm, pk, pn, have_pm ? 1.0 - omega : 0.0, omega,
                omega * scale);
    c_km1 = c_k; c_k = c_kp1; ++cur; have_pm = 1;
  }
  return bufs[cur % 3];
}

static double *t_vcycle(team_mg_t *g, const csr_t *Afine, int l,
                        const double *b) {
  if (l == 0) { t_spmv(&g->A[0], b, g->x[0], 0, NULL); return g->x[0]; }
  const csr_t *A = (l == g->levels - 1) ? Afine : &g->A[l];
  double *bufs[3] = {g->x[l], g->t0[l], g->t1[l]};
  double *px = t_smooth(A, g->emin[l], g->emax[l], g->nu_pre, b, bufs, 1);
  if (g->nu_pre == 0) {
    T_LOOP(i, A->nrows) px[i] = 0.0;
  }
  const double *r = b;
  if (g->nu_pre > 0) { t_spmv(A, px, g->r[l], 2, b); r = g->r[l]; }
  t_spmv(&g->R[l], r, g->b[l - 1], 0, NULL);
  double *pe = t_vcycle(g, Afine, l - 1, g->b[l - 1]);
  double *post[3]; int j = 0;
  post[0] = px;
  for (int q = 0; q < 3; ++q) if (bufs[q] != px) post[++j] = bufs[q];
  t_spmv(&g->P[l], pe, post[0], 1, px);            /* x += P e (in place) */
  return t_smooth(A, g->emin[l], g->emax[l], g->nu_post, b, post, 0);
}

/* x = solve(b); x is written by the last loop */
static void t_inner(pcdo_t *h, team_t *T, int slot, const double *b, double *x) {
  const csr_t *A = &T->mat[slot_mat[slot]];
  inner_t *s = &h->inner[slot];
  int64_t n = A->nrows;
  if (s->pc == PC_MG) {
    int its = (s->ksp == KSP_PREONLY) ? 1 : s->max_it;
    for (int it = 0; it < its; ++it) {
      const double *r = b;
      if (it > 0) { t_spmv(A, x, T->cw[slot][0], 2, b); r = T->cw[slot][0]; }
      double *z = t_vcycle(&T->mg[slot], A, T->mg[slot].levels - 1, r);
      if (it == 0) { T_LOOP(i, n) x[i] = z[i]; }
      else { T_LOOP(i, n) x[i] += z[i]; }
    }
    return;
  }
  if (s->ksp == KSP_CHEBYSHEV) {
    double *ring[3] = {T->cw[slot][0], T->cw[slot][1], T->cw[slot][2]};
    double scale = 2.0 / (s->emax + s->emin), alpha = 1.0 - scale * s->emin;
    double mu = 1.0 / alpha, omegaprod = 2.0 / alpha, c_km1 = 1.0, c_k = mu;
    if (s->pc == PC_JACOBI) t_scale_dinv(A, b, scale, ring[0]);
    else { T_LOOP(i, n) ring[0][i] = scale * b[i]; }
    for (int it = 0; it < s->max_it; ++it) {
      double c_kp1 = 2.0 * mu * c_k - c_km1, omega = omegaprod * c_k / c_kp1;
      double *pk = ring[it % 3], *pn = ring[(it + 1) % 3];
      double *pm = it ? ring[(it + 2) % 3] : pk;
      t_cheb_step(A, b, pm, pk, pn, it ? 1.0 - omega : 0.0, omega, omega * scale);
      c_km1 = c_k; c_k = c_kp1;
    }
    const double *res = ring[s->max_it % 3];
    T_LOOP(i, n) x[i] = res[i];
    return;
  }
  if (s->ksp == KSP_PREONLY) { t_scale_dinv(A, b, 1.0, x); return; }
  if (s->ksp == KSP_RICHARDSON) {
    t_scale_dinv(A, b, 1.0, x);
    for (int it = 1; it < s->max_it; ++it) {
      t_cheb_step(A, b, x, x, T->cw[slot][0], 0.0, 1.0, 1.0);
      T_LOOP(i, n) x[i] = T->cw[slot][0][i];
    }
    return;
  }
  /* KSP_CG + Jacobi, natural norm; the reductions go through two shared
   * scalars, every thread takes the same decision */
  double *r = T->cw[slot][0], *z = T->cw[slot][1], *p = T->cw[slot][2];
  double *q = T->mg[slot].x[MG_MAX_LEVELS - 1];
  t_sync();
  T_LOOP(i, n) {
    x[i] = 0.0; r[i] = b[i]; z[i] = A->dinv[i] * b[i]; p[i] = z[i];
  }
  t_sync();
#pragma omp single
  T->rz = 0.0;
  { double acc = 0.0;
    _Pragma("omp for schedule(static) nowait")
    for (int64_t i = 0; i < n; ++i) acc += r[i] * z[i];
    _Pragma("omp atomic") T->rz += acc; }
#pragma omp barrier
  double rz = T->rz, rz0 = rz;
  for (int it = 0; it < s->max_it; ++it) {
    if (rz == 0.0) break;
    if (s->rtol > 0.0 && sqrt(fabs(rz)) <= s->rtol * sqrt(fabs(rz0))) break;
  t_sync();
#pragma omp single
    T->pq = 0.0;
    { double acc = 0.0;
      _Pragma("omp for schedule(static) nowait")
      for (int64_t i = 0; i < n; ++i) {
        double sum = 0.0;
        for (int32_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k)
          sum += A->val[k] * p[A->col[k]];
        q[i] = sum; acc += p[i] * sum;
      }
      _Pragma("omp atomic") T->pq += acc; }
#pragma omp barrier
    double alpha_ = rz / T->pq;
#pragma omp barrier
  t_sync();
#pragma omp single
    T->rz = 0.0;
    { double acc = 0.0;
      _Pragma("omp for schedule(static) nowait")
      for (int64_t i = 0; i < n; ++i) {
        x[i] += alpha_ * p[i];
        double ri = r[i] - alpha_ * q[i], zi = A->dinv[i] * ri;
        r[i] = ri; z[i] = zi; acc += ri * zi;
      }
      _Pragma("omp atomic") T->rz += acc; }
#pragma omp barrier
    double rz_new = T->rz, beta = rz_new / rz;
    rz = rz_new;
    T_LOOP(i, n) p[i] = z[i] + beta * p[i];
  }
}

static void t_pcd(pcdo_t *h, team_t *T, const double *x, double *y) {
  int64_t n = h->n_p;
  double *z = T->w0, *z1 = T->w1;
  if (h->variant == BRM1 || h->variant == RBRM1) {
    T_LOOP(i, n) z[i] = x[i];
  t_sync();
#pragma omp single
    for (int64_t i = 0; i < h->n_bc; ++i) z[h->bc_idx[i]] = h->bc_val[i];
    t_inner(h, T, SLOT_AP, z, y);
    t_spmv(&T->mat[MAT_KP], y, z1, 1, x);              /* z1 = Kp y + x */
    t_inner(h, T, SLOT_MP, z1, y);
    if (h->variant == RBRM1) {
      t_inner(h, T, SLOT_RP, x, z);
      T_LOOP(i, n) y[i] = -(y[i] + z[i]);
    } else { T_LOOP(i, n) y[i] = -y[i]; }
  } else {
    t_inner(h, T, SLOT_MP, x, y);
    t_spmv(&T->mat[MAT_KP], y, z1, 0, NULL);
  t_sync();
#pragma omp single
    for (int64_t i = 0; i < h->n_bc; ++i) z1[h->bc_idx[i]] = h->bc_val[i];
    t_inner(h, T, SLOT_AP, z1, z);
    if (h->variant == RBRM2) {
      T_LOOP(i, n) y[i] += z[i];
      t_inner(h, T, SLOT_RP, x, z);
    }
    T_LOOP(i, n) y[i] = -(y[i] + z[i]);
  }
}

static int team_slot(pcdo_t *h) {
  for (int i = 0; i < 64; ++i) if (g_team_owner[i] == h) return i;
  return -1;
}

/* first-touch copies of everything the apply streams; freezes the threads */
static void t_free_csr(csr_t *m) {
  free(m->rowptr); free(m->col); free(m->val); free(m->dinv);
  memset(m, 0, sizeof *m);
}
static void t_free_team(team_t *T) {
  for (int m = 0; m < MAT_COUNT; ++m) t_free_csr(&T->mat[m]);
  for (int s = 0; s < SLOT_COUNT; ++s) {
    for (int k = 0; k < 3; ++k) free(T->cw[s][k]);
    team_mg_t *g = &T->mg[s];
    for (int l = 0; l < MG_MAX_LEVELS; ++l) {
      t_free_csr(&g->A[l]); t_free_csr(&g->P[l]); t_free_csr(&g->R[l]);
      free(g->x[l]); free(g->t0[l]); free(g->t1[l]); free(g->r[l]); free(g->b[l]);
    }
  }
  free(T->w0); free(T->w1); free(T->wu); free(T->xs); free(T->ys); free(T->perm);
  free(T);
}

int pcdo_team_prepare(pcdo_t *h, int threads) {
  if (!h->ready || !h->mat[MAT_A00].set)
    return fail(4, "team_prepare: system/setup missing");
  if (threads < 1) threads = omp_get_max_threads();
  { const char *e = getenv("PCDO_TEAM_BIG"); if (e) t_big = atoi(e);
    e = getenv("PCDO_TEAM_SUB"); if (e && atoi(e) > 0) t_sub = atoi(e);
    e = getenv("PCDO_TEAM_GROUP"); t_grp = e ? atoi(e) : 0; }
  int slot = team_slot(h);
  if (slot < 0) for (int i = 0; i < 64 && slot < 0; ++i) if (!g_team_owner[i]) slot = i;
  if (slot < 0) return fail(3, "team_prepare: too many engines");
  /* a re-prepare frees the previous team's copies (bench.py sweeps ten thread
   * counts: at config 5's size every leaked team was ~12 GB - round 6) */
  if (g_team_owner[slot] == h && g_team[slot]) { t_free_team(g_team[slot]); g_team[slot] = NULL; }
  team_t *T = (team_t *)calloc(1, sizeof *T);
  T->threads = threads;
  for (int m = 0; m < MAT_COUNT; ++m) t_copy_csr(&T->mat[m], &h->mat[m], threads);
  for (int s = 0; s < SLOT_COUNT; ++s) {
    inner_t *in = &h->inner[s];
    const csr_t *A = &h->mat[slot_mat[s]];
    if (!A->set) continue;
    for (int k = 0; k < 3; ++k) T->cw[s][k] = t_vec(A->nrows, threads);
    T->mg[s].x[MG_MAX_LEVELS - 1] = t_vec(A->nrows, threads);   /* CG's q */
    if (in->pc != PC_MG) continue;
    team_mg_t *g = &T->mg[s];
    g->levels = in->mg_levels; g->nu_pre = in->nu_pre; g->nu_post = in->nu_post;
    for (int l = 0; l < in->mg_levels; ++l) {
      t_copy_csr(&g->A[l], &in->mgA[l], threads);
      t_copy_csr(&g->P[l], &in->mgP[l], threads);
      t_copy_csr(&g->R[l], &in->mgR[l], threads);
      g->emin[l] = in->mg_emin[l]; g->emax[l] = in->mg_emax[l];
      int64_t nl = (l == in->mg_levels - 1) ? A->nrows : in->mgA[l].nrows;
      g->x[l] = t_vec(nl, threads); g->t0[l] = t_vec(nl, threads);
      g->t1[l] = t_vec(nl, threads); g->r[l] = t_vec(nl, threads);
      g->b[l] = t_vec(nl, threads);
    }
  }
  int64_t n = h->n_u + h->n_p;
  T->w0 = t_vec(h->n_p, threads); T->w1 = t_vec(h->n_p, threads);
  T->wu = t_vec(h->n_u, threads);
  T->xs = t_vec(n, threads); T->ys = t_vec(n, threads);
  T->perm = (int32_t *)t_alloc(sizeof(int32_t) * n);
  t_region();
#pragma omp parallel num_threads(threads)
  {
    t_begin();
    T_LOOP(i, n) T->perm[i] = h->perm[i];
  }
  g_team[slot] = T; g_team_owner[slot] = h;
  return 0;
}

int pcdo_team_fieldsplit_apply(pcdo_t *h, const double *x, double *y) {
  int slot = team_slot(h);
  if (slot < 0) return fail(4, "team_fieldsplit_apply: call pcdo_team_prepare first");
  team_t *T = g_team[slot];
  int64_t nu = h->n_u, n = h->n_u + h->n_p;
  t_region();
#pragma omp parallel num_threads(T->threads)
  {
    t_begin();
    T_LOOP(i, n) T->xs[i] = x[T->perm[i]];
    t_pcd(h, T, T->xs + nu, T->ys + nu);                /* y_p = S^-1 x_p */
    t_spmv(&T->mat[MAT_A01], T->ys + nu, T->wu, 2, T->xs);   /* x_u - A01 y_p */
    t_inner(h, T, SLOT_A00, T->wu, T->ys);
    T_LOOP(i, n) y[T->perm[i]] = T->ys[i];
  }
  ++h->num_fs;
  return 0;
}

/* STREAM triad a = b + s*c on first-touched arrays: GB/s (24 B per entry) */
double pcdo_stream_triad(int64_t n, int reps, int threads) {
  if (threads < 1) threads = omp_get_max_threads();
  double *a = t_vec(n, threads), *b = t_vec(n, threads), *c = t_vec(n, threads);
  double best = 0.0;
  for (int r = 0; r < reps + 1; ++r) {
    double t0 = omp_get_wtime();
  t_region();
#pragma omp parallel num_threads(threads)
    {
      t_begin();
      T_LOOP(i, n) a[i] = b[i] + 3.0 * c[i];
    }
    double dt = omp_get_wtime() - t0;
    double gbs = 24.0 * (double)n / dt / 1e9;
    if (r > 0 && gbs > best) best = gbs;
    b[r % n] += a[(r * 7) % n];          /* keep the loop alive */
  }
  free(a); free(b); free(c);
  return best;
}
#endif /* _OPENMP */

int pcdo_synchronize(pcdo_t *h) { (void)h; return 0; }

/* threads of the OpenMP timing build; returns the count in effect (1 when
 * built without OpenMP) */
int pcdo_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}
