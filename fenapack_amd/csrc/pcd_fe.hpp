// Device operator producer: the matrices that change every nonlinear iteration
// (convection part of the velocity block on every multigrid level, Kp) are
// assembled in HBM from the iterate, straight into the engine's operators.
//
// Replaces, for the fixed P2/P1 Picard forms, the per-iteration callbacks of
// fenapack/assembling.py:151-189 (`system_matrix`, `kp`) that the reference
// hands to DOLFIN's assembler [ext]; SURVEY.md section 8(f1).
//
// Design (MI355X): element loops are embarrassingly parallel and tiny per
// cell, so the layout is what matters -
//   * every per-cell array is stored component-major ([entry][cell]) so that
//     consecutive lanes (= consecutive cells) read and write unit-stride;
//   * one thread per (cell, local row) keeps the register footprint at one
//     element-matrix row (10 doubles in 3-D) instead of the full 10 x 10;
//   * the scatter is a GATHER: every CSR entry sums its (<= ~6 in 2-D)
//     element contributions from a precomputed list, in a fixed order - no
//     atomics, bitwise reproducible, and the Dirichlet treatment
//     (`keep` mask + diagonal values) rides on the same pass.
// Included through pcd_internal.hpp by every translation unit of the engine.
#pragma once

namespace pcd {

struct FeTables {
  int nq;
  const double* qw;     // nq          reference weights (sum = 1)
  const double* phi;    // nq * NA     P2 basis at the quadrature points
  const double* dphi;   // nq * NA * NV  d phi_a / d lambda_k
  const double* psi;    // nq * NV     P1 basis (= barycentric coordinates)
  const double* phic;   // NA          P2 basis at the cell midpoint (SUPG)
  // the streamline-diffusion term has degree 6 with a P2 wind: its own rule
  int nq_s;
  const double* qw_s;   // nq_s
  const double* phi_s;  // nq_s * NA
  const double* dphi_s; // nq_s * NA * NV
};

// One thread per (cell c, local row a): row a of the P2 convection matrix
//   C[a][b] = sum_q w_q |K| phi_a(q) (w(q) . grad phi_b(q)),
//   w(q) = sum_a phi_a(q) U_a.
// dofs2 [a][c], gradlam [(k*DIM+d)][c], cells out [(a*NA+b)][c].
// With `cells_s` (optional): the streamline-diffusion term of the SUPG
// preconditioner matrix (demo_navier-stokes-pcd.py:122-125),
//   S[a][b] = delta_c sum_q w_q |K| (w.grad phi_a)(w.grad phi_b),
//   Pe = |w_mid| h rho / (2 nu),  delta_c = Pe > 1 ? h (1 - 1/Pe) / (2 |w_mid|) : 0
// (fenapack/stabilization.py:66-67, rho = 1, h = `cell_h`).
template <int DIM>
__global__ __launch_bounds__(kBlock) void k_fe_convection_p2(
    int nc, const int* __restrict__ dofs2, const double* __restrict__ gradlam,
    const double* __restrict__ measure, const FeTables T,
    const double* __restrict__ U, double* __restrict__ cells,
    const double* __restrict__ cell_h, double nu, double* __restrict__ cells_s) {
  constexpr int NV = DIM + 1, NA = DIM == 2 ? 6 : 10;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= (int64_t)NA * nc) return;
  const int a = (int)(t / nc), c = (int)(t % nc);
  double g[NV][DIM], Uc[NA][DIM], acc[NA];
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int d = 0; d < DIM; ++d) g[k][d] = gradlam[(int64_t)(k * DIM + d) * nc + c];
#pragma unroll
  for (int b = 0; b < NA; ++b) {
    const int node = dofs2[(int64_t)b * nc + c];
#pragma unroll
    for (int d = 0; d < DIM; ++d) Uc[b][d] = U[(int64_t)DIM * node + d];
    acc[b] = 0.0;
  }
  const double meas = measure[c];
  double delta = 0.0, accs[NA];
  if (cells_s) {
    double wm[DIM], n2 = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) wm[d] = 0.0;
#pragma unroll
    for (int b = 0; b < NA; ++b)
#pragma unroll
      for (int d = 0; d < DIM; ++d) wm[d] += T.phic[b] * Uc[b][d];
#pragma unroll
    for (int d = 0; d < DIM; ++d) n2 += wm[d] * wm[d];
    const double wn = sqrt(n2), hc = cell_h[c];
    const double pe = 0.5 * wn * hc / nu;
    if (pe > 1.0) delta = 0.5 * hc * (1.0 - 1.0 / pe) / wn;
#pragma unroll
    for (int b = 0; b < NA; ++b) accs[b] = 0.0;
  }
  for (int q = 0; q < T.nq; ++q) {
    double w[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) w[d] = 0.0;
#pragma unroll
    for (int b = 0; b < NA; ++b) {
      const double ph = T.phi[q * NA + b];
#pragma unroll
      for (int d = 0; d < DIM; ++d) w[d] += ph * Uc[b][d];
    }
    double wl[NV];                       // w . grad lambda_k
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double s = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) s += w[d] * g[k][d];
      wl[k] = s;
    }
    const double f = T.qw[q] * meas * T.phi[q * NA + a];
#pragma unroll
    for (int b = 0; b < NA; ++b) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < NV; ++k) s += T.dphi[(q * NA + b) * NV + k] * wl[k];
      acc[b] += f * s;
    }
  }
  if (cells_s && delta != 0.0) {
    for (int q = 0; q < T.nq_s; ++q) {
      double w[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d) w[d] = 0.0;
#pragma unroll
      for (int b = 0; b < NA; ++b) {
        const double ph = T.phi_s[q * NA + b];
#pragma unroll
        for (int d = 0; d < DIM; ++d) w[d] += ph * Uc[b][d];
      }
      double wl[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) s += w[d] * g[k][d];
        wl[k] = s;
      }
      double sa = 0.0;                   // (w . grad phi_a)(q)
#pragma unroll
      for (int k = 0; k < NV; ++k) sa += T.dphi_s[(q * NA + a) * NV + k] * wl[k];
      sa *= T.qw_s[q] * meas * delta;
#pragma unroll
      for (int b = 0; b < NA; ++b) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NV; ++k) s += T.dphi_s[(q * NA + b) * NV + k] * wl[k];
        accs[b] += sa * s;
      }
    }
  }
#pragma unroll
  for (int b = 0; b < NA; ++b) cells[(int64_t)(a * NA + b) * nc + c] = acc[b];
  if (cells_s) {
#pragma unroll
    for (int b = 0; b < NA; ++b) cells_s[(int64_t)(a * NA + b) * nc + c] = accs[b];
  }
}

// Newton linearisation (demo_navier-stokes-pcd.py:113-116, `derivative(F, w)`):
// the term ((du . grad) w, v) couples the velocity components,
//   N_ij[a][b] = sum_q w_q |K| phi_a(q) phi_b(q) d_j w_i(q),
// d x d scalar matrices on the pattern of F.  One thread per (cell, local row
// a); the velocity gradient at a quadrature point is recomputed per column b
// (a handful of flops) so that only d*d accumulators live in registers.
// cells out [((i*DIM+j)*NA*NA + a*NA+b)][c].
template <int DIM>
__global__ __launch_bounds__(kBlock) void k_fe_newton_p2(
    int nc, const int* __restrict__ dofs2, const double* __restrict__ gradlam,
    const double* __restrict__ measure, const FeTables T,
    const double* __restrict__ U, double* __restrict__ cells) {
  constexpr int NV = DIM + 1, NA = DIM == 2 ? 6 : 10;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= (int64_t)NA * nc) return;
  const int a = (int)(t / nc), c = (int)(t % nc);
  double g[NV][DIM], Uc[NA][DIM];
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int d = 0; d < DIM; ++d) g[k][d] = gradlam[(int64_t)(k * DIM + d) * nc + c];
#pragma unroll
  for (int b = 0; b < NA; ++b) {
    const int node = dofs2[(int64_t)b * nc + c];
#pragma unroll
    for (int d = 0; d < DIM; ++d) Uc[b][d] = U[(int64_t)DIM * node + d];
  }
  const double meas = measure[c];
  const int64_t plane = (int64_t)NA * NA * nc;
  for (int b = 0; b < NA; ++b) {
    double acc[DIM][DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i)
#pragma unroll
      for (int j = 0; j < DIM; ++j) acc[i][j] = 0.0;
    for (int q = 0; q < T.nq; ++q) {
      double gw[DIM][DIM];               // gw[i][j] = d_j w_i at q
#pragma unroll
      for (int i = 0; i < DIM; ++i)
#pragma unroll
        for (int j = 0; j < DIM; ++j) gw[i][j] = 0.0;
#pragma unroll
      for (int e = 0; e < NA; ++e) {
        double ge[DIM];                  // grad phi_e at q
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < NV; ++k) s += T.dphi[(q * NA + e) * NV + k] * g[k][j];
          ge[j] = s;
        }
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int j = 0; j < DIM; ++j) gw[i][j] += Uc[e][i] * ge[j];
      }
      const double f = T.qw[q] * meas * T.phi[q * NA + a] * T.phi[q * NA + b];
#pragma unroll
      for (int i = 0; i < DIM; ++i)
#pragma unroll
        for (int j = 0; j < DIM; ++j) acc[i][j] += f * gw[i][j];
    }
#pragma unroll
    for (int i = 0; i < DIM; ++i)
#pragma unroll
      for (int j = 0; j < DIM; ++j)
        cells[(int64_t)(i * DIM + j) * plane + (int64_t)(a * NA + b) * nc + c] = acc[i][j];
  }
}

// the ncomp = d*d scalar matrices of the Newton term gathered on the pattern of
// F: out[m*nnz + k] = sum of the element contributions of plane m (list order),
// Dirichlet rows/columns removed (keep); `unc` (optional) without the mask
static __global__ __launch_bounds__(kBlock) void k_fe_gather_blocks(
    int64_t nnz, int ncomp, int64_t plane, const int* __restrict__ ptr,
    const int* __restrict__ src, const double* __restrict__ cells,
    const unsigned char* __restrict__ keep, double* unc, double* out) {
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz;
       k += (int64_t)gridDim.x * kBlock) {
    const bool kept = !(keep && !keep[k]);
    for (int m = 0; m < ncomp; ++m) {
      const double* cm = cells + (int64_t)m * plane;
      double s = 0.0;
      for (int t = ptr[k]; t < ptr[k + 1]; ++t) s += cm[src[t]];
      if (unc) unc[(int64_t)m * nnz + k] = s;
      out[(int64_t)m * nnz + k] = kept ? s : 0.0;
    }
  }
}

// dst[pos[(i*d+j)*nnz + k]] = delta_ij F[k] + N[(i*d+j)*nnz + k]: the coupled
// velocity block F x I_d + N written into a general CSR (F, N optional)
template <class I>
__global__ __launch_bounds__(kBlock) void k_fe_scatter_blocks(
    int64_t nnz, int d, const I* __restrict__ pos, const double* __restrict__ F,
    const double* __restrict__ N, double* dst) {
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz;
       k += (int64_t)gridDim.x * kBlock) {
    const double f = F ? F[k] : 0.0;
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j) {
        const int64_t m = (int64_t)(i * d + j) * nnz + k;
        dst[pos[m]] = (i == j ? f : 0.0) + (N ? N[m] : 0.0);
      }
  }
}

// One thread per cell: P1 convection matrix of the pressure space
//   K[i][j] = scale * sum_q w_q |K| psi_i(q) (w(q) . grad lambda_j)
// (grad lambda_j is constant on the cell); cells out [(i*NV+j)][c].
template <int DIM>
__global__ __launch_bounds__(kBlock) void k_fe_convection_p1(
    int nc, const int* __restrict__ dofs2, const double* __restrict__ gradlam,
    const double* __restrict__ measure, const FeTables T,
    const double* __restrict__ U, double scale, double* __restrict__ cells) {
  constexpr int NV = DIM + 1, NA = DIM == 2 ? 6 : 10;
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= nc) return;
  double g[NV][DIM], Uc[NA][DIM], m[NV][DIM];
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      g[k][d] = gradlam[(int64_t)(k * DIM + d) * nc + c];
      m[k][d] = 0.0;
    }
#pragma unroll
  for (int b = 0; b < NA; ++b) {
    const int node = dofs2[(int64_t)b * nc + c];
#pragma unroll
    for (int d = 0; d < DIM; ++d) Uc[b][d] = U[(int64_t)DIM * node + d];
  }
  const double meas = measure[c] * scale;
  for (int q = 0; q < T.nq; ++q) {
    double w[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) w[d] = 0.0;
#pragma unroll
    for (int b = 0; b < NA; ++b) {
      const double ph = T.phi[q * NA + b];
#pragma unroll
      for (int d = 0; d < DIM; ++d) w[d] += ph * Uc[b][d];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const double f = T.qw[q] * meas * T.psi[q * NV + i];
#pragma unroll
      for (int d = 0; d < DIM; ++d) m[i][d] += f * w[d];
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      double s = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) s += m[i][d] * g[j][d];
      cells[(int64_t)(i * NV + j) * nc + c] = s;
    }
}

// entry k = cst[k] + sum of its element contributions (list order); `unc`
// (optional) receives the unconstrained value, `out` the one with Dirichlet
// rows/columns removed (keep[k] == 0)
// `cells_s` / `out_s` (optional): the same entry with the stabilisation
// contributions added (the preconditioner's operator); `out` may then be null
static __global__ __launch_bounds__(kBlock) void k_fe_gather(
    int64_t nnz, const int* __restrict__ ptr, const int* __restrict__ src,
    const double* __restrict__ cells, const double* __restrict__ cst,
    const unsigned char* __restrict__ keep, double* unc, double* out,
    const double* __restrict__ cells_s, double* out_s) {
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz;
       k += (int64_t)gridDim.x * kBlock) {
    double s = cst ? cst[k] : 0.0;
    for (int t = ptr[k]; t < ptr[k + 1]; ++t) s += cells[src[t]];
    const bool kept = !(keep && !keep[k]);
    if (unc) unc[k] = s;
    if (out) out[k] = kept ? s : 0.0;
    if (out_s) {
      double s2 = s;
      for (int t = ptr[k]; t < ptr[k + 1]; ++t) s2 += cells_s[src[t]];
      out_s[k] = kept ? s2 : 0.0;
    }
  }
}

// out[k] = sum_t w[t] * in[src[t]]: one stage of the numeric Galerkin triple
// product P^T F P on fixed patterns (B = F P, then F_c = P^T B); the lists
// carry the prolongation weights, so the product is two weighted gathers
static __global__ __launch_bounds__(kBlock) void k_fe_wgather(
    int64_t nnz, const int64_t* __restrict__ ptr, const int* __restrict__ src,
    const double* __restrict__ w, const double* __restrict__ in, double* out) {
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz;
       k += (int64_t)gridDim.x * kBlock) {
    double s = 0.0;
    for (int64_t t = ptr[k]; t < ptr[k + 1]; ++t) s += w[t] * in[src[t]];
    out[k] = s;
  }
}

// C = A B on FIXED patterns - the numeric phase of a sparse product whose
// symbolic phase ran once on the host (libpcd_host pcdh_spgemm_*): what PETSc's
// MatMatMult / transposeMatMult(..., result=) do when the reference refreshes
// R_p (fenapack/field_split_backend.py:160-166) and hypre / GAMG when they
// re-form the coarse operators of a hierarchy whose aggregation is kept.  The
// two Galerkin stages B = F P and F_c = P^T B of the device operator producer.
//
// A GROUP of G lanes owns a row i of C: its sorted column list and one
// accumulator per entry live in LDS (CAP entries per pass; a longer row takes
// several passes over A's row, each restricted to its column window).  The
// entries k of A's row are taken IN ORDER; for each one the group's lanes run
// along B's row j = col_A[k] - the columns of one row are distinct, so the
// lanes of a step never meet in an accumulator (no atomics) - find the
// column in the LDS list by bisection and add a_ik * b_jJ there.  Every entry
// of C is thus summed in the order of A's row: reproducible, the same bits on
// every rank that computes it, and the order of the gather plans this kernel
// replaced (pcdh_product_plan_*: ~12 B per TERM resident in HBM, hundreds of
// millions of terms for a 3-D hierarchy; here: nothing but the patterns).
// LDS traffic of one wave is ordered (DS operations execute in issue order),
// which is what lets step k + 1 read what step k wrote without a barrier.
constexpr int kSpgemmSlots = 4096;       // C entries a workgroup holds (16 KB of columns + 32 KB of sums)
template <int G>
static __global__ __launch_bounds__(kBlock) void k_spgemm_fixed(
    int nrows, const int* __restrict__ arp, const int* __restrict__ ac, const double* __restrict__ av,
    const int* __restrict__ brp, const int* __restrict__ bc, const double* __restrict__ bv,
    const int* __restrict__ crp, const int* __restrict__ cc, double* __restrict__ cv) {
  constexpr int NG = kBlock / G;                 // groups per workgroup
  constexpr int CAP = kSpgemmSlots / NG;         // C entries of a row per pass
  static_assert(G >= 8 && G <= 64 && (G & (G - 1)) == 0, "a group is a power-of-two part of a wave");
  __shared__ int s_col[kSpgemmSlots];
  __shared__ double s_acc[kSpgemmSlots];
  const int g = threadIdx.x / G, lane = threadIdx.x % G;
  int* col = s_col + g * CAP;
  double* acc = s_acc + g * CAP;
  for (int i = blockIdx.x * NG + g; i < nrows; i += gridDim.x * NG) {
    const int c0 = crp[i], nC = crp[i + 1] - c0;
    const int a0 = arp[i], a1 = arp[i + 1];
    for (int p0 = 0; p0 < nC; p0 += CAP) {
      const int np = min(CAP, nC - p0);
      for (int t = lane; t < np; t += G) { col[t] = cc[c0 + p0 + t]; acc[t] = 0.0; }
      __builtin_amdgcn_wave_barrier();
      const int lo = col[0], hi = col[np - 1];
      for (int k = a0; k < a1; ++k) {
        const int j = ac[k];
        const double a = av[k];
        const int q1 = brp[j + 1];
        for (int q = brp[j] + lane; q < q1; q += G) {
          const int J = bc[q];
          if (J < lo || J > hi) continue;          // (another pass' window)
          int l = 0, r = np - 1;
          while (l < r) {
            const int m = (l + r) >> 1;
            if (col[m] < J) l = m + 1; else r = m;
          }
          // (a structural pattern holds every J; the test keeps a foreign
          // pattern from corrupting a neighbour's sum)
          if (col[l] == J) acc[l] += a * bv[q];
        }
        __builtin_amdgcn_wave_barrier();
      }
      for (int t = lane; t < np; t += G) cv[c0 + p0 + t] = acc[t];
      __builtin_amdgcn_wave_barrier();
    }
  }
}

static __global__ __launch_bounds__(kBlock) void k_fe_set(
    int n, const int* __restrict__ pos, const double* __restrict__ val,
    double* out) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    out[pos[i]] = val[i];
}

// dst[pos[c * nnz + k]] = F[k], c < ncomp: the scalar operator written into
// every component's entries of an interleaved (F x I) matrix
template <class I>
__global__ __launch_bounds__(kBlock) void k_fe_scatter(
    int64_t nnz, int ncomp, const I* __restrict__ pos,
    const double* __restrict__ F, double* dst) {
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz;
       k += (int64_t)gridDim.x * kBlock) {
    const double v = F[k];
    for (int c = 0; c < ncomp; ++c) dst[pos[(int64_t)c * nnz + k]] = v;
  }
}

// wind of the next coarser level by injection (P2 spaces are nested)
static __global__ __launch_bounds__(kBlock) void k_fe_inject(
    int64_t nn, int dim, const int* __restrict__ inject,
    const double* __restrict__ Uf, double* Uc) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nn;
       i += (int64_t)gridDim.x * kBlock) {
    const int64_t j = inject[i];
    for (int d = 0; d < dim; ++d) Uc[dim * i + d] = Uf[dim * j + d];
  }
}

// power iteration helper: y *= dinv (if given); parts[block] = sum y^2
static __global__ __launch_bounds__(kBlock) void k_fe_scale_sqnorm(
    int64_t n, const double* __restrict__ dinv, double* y, double* parts) {
  __shared__ double sm[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    const double v = dinv ? y[i] * dinv[i] : y[i];
    y[i] = v;
    s += v * v;
  }
  s = block_sum(s, sm);
  if (threadIdx.x == 0) parts[blockIdx.x] = s;
}

// ---- dense inverse of the coarsest scalar operator: Gauss-Jordan with
// partial pivoting on [F | I] (row-major, ld = 2n), rows never physically
// swapped: step k picks the unused row p with the largest |W[i][k]|, eliminates
// column k from every other row, and row p becomes row k of the inverse after a
// final scaling.  ONE launch per step: every workgroup (= one row) first finds
// the pivot itself from `colcur` (the column's moduli as left by the previous
// step, -1 for rows already used), so there is no separate pivot kernel and no
// grid-wide synchronisation.  n ~ 10^3: n short launches (~2 ms) replace a
// device->host->device round trip around LAPACK that cost 20x more.
static __global__ __launch_bounds__(kBlock) void k_gj_init(
    int n, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, double* W, double* colcur) {
  const int i = blockIdx.x;
  const int ld = 2 * n;
  for (int j = threadIdx.x; j < ld; j += kBlock) W[(int64_t)i * ld + j] = (j == n + i) ? 1.0 : 0.0;
  __syncthreads();
  for (int k = rowptr[i] + threadIdx.x; k < rowptr[i + 1]; k += kBlock)
    W[(int64_t)i * ld + col[k]] = val[k];
  __syncthreads();
  if (threadIdx.x == 0) colcur[i] = fabs(W[(int64_t)i * ld]);
}

static __global__ __launch_bounds__(kBlock) void k_gj_step(
    int n, int k, double* W, const double* __restrict__ colcur, double* colnext,
    int* pivrow, int* singular) {
  __shared__ double sv[kBlock];
  __shared__ int si[kBlock];
  const int i = blockIdx.x;
  const int ld = 2 * n;
  // pivot: largest modulus among the unused rows, lowest index on ties
  double best = -1.0;
  int bi = -1;
  for (int r = threadIdx.x; r < n; r += kBlock) {
    const double a = colcur[r];
    if (a > best) { best = a; bi = r; }
  }
  sv[threadIdx.x] = best; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const double o = sv[threadIdx.x + s];
      const int oi = si[threadIdx.x + s];
      if (o > sv[threadIdx.x] || (o == sv[threadIdx.x] && oi >= 0 && (si[threadIdx.x] < 0 || oi < si[threadIdx.x]))) {
        sv[threadIdx.x] = o; si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  const int p = si[0];
  if (!(sv[0] > 0.0) || p < 0) { if (i == 0 && threadIdx.x == 0) *singular = 1; return; }
  const bool used = colcur[i] < 0.0;
  if (i == p) {                            // the pivot row stays as it is
    if (threadIdx.x == 0) { pivrow[k] = p; colnext[i] = -1.0; }
    return;
  }
  const double f = W[(int64_t)i * ld + k] / W[(int64_t)p * ld + k];
  __syncthreads();                          // every lane has read f's operands
  if (f != 0.0) {
    for (int j = threadIdx.x; j < ld; j += kBlock)
      if (j != k) W[(int64_t)i * ld + j] -= f * W[(int64_t)p * ld + j];
    if (threadIdx.x == 0) W[(int64_t)i * ld + k] = 0.0;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    colnext[i] = used ? -1.0 : (k + 1 < n ? fabs(W[(int64_t)i * ld + k + 1]) : 0.0);
}

// dense (d n)^2 row-major out = inv(F) x I_d: row k of inv(F) is the right half
// of the row that served as pivot of column k, divided by that pivot
static __global__ __launch_bounds__(kBlock) void k_gj_store(int n, int d, const double* __restrict__ W,
                                                      const int* __restrict__ pivrow, double* out) {
  const int64_t N = (int64_t)n * d, total = N * N;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int64_t r = t / N, cidx = t % N;
    const int k = (int)(r / d), ci = (int)(r % d), j = (int)(cidx / d), cj = (int)(cidx % d);
    double v = 0.0;
    if (ci == cj) {
      const int64_t p = pivrow[k];
      v = W[p * 2 * n + n + j] / W[p * 2 * n + k];
    }
    out[t] = v;
  }
}

// the same without the entries that couple no components: row (k, c) of
// inv(F) x I_d keeps its n entries (j, c), j ascending - the CSR order
static __global__ __launch_bounds__(kBlock) void k_gj_store_compact(int n, int d, const double* __restrict__ W,
                                                              const int* __restrict__ pivrow, double* out) {
  const int64_t total = (int64_t)n * d * n;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int64_t r = t / n;
    const int j = (int)(t % n), k = (int)(r / d);
    const int64_t p = pivrow[k];
    out[t] = W[p * 2 * n + n + j] / W[p * 2 * n + k];
  }
}

// ---- BRM2 boundary term of Kp: - (1/nu) int_{inflow} (w.n) p q ds ------------
// (demo_navier-stokes-pcd.py:131-135).  One thread per boundary edge (2-D):
// the P2 wind along the edge from its end points and midpoint, 3-point Gauss
// rule (degree 5; the integrand has degree 4), local 2 x 2 matrix
// loc[(i*2+j)][e] = L_e sum_q w_q (w.n)(q) psi_i(q) psi_j(q).
static __global__ __launch_bounds__(kBlock) void k_fe_robin_edges(
    int nb, const int* __restrict__ nodes, const double* __restrict__ normal,
    const double* __restrict__ length, const double* __restrict__ U, double* loc) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= nb) return;
  const double gx[3] = {0.5 - 0.5 * 0.7745966692414834, 0.5, 0.5 + 0.5 * 0.7745966692414834};
  const double gw[3] = {0.5 * 5.0 / 9.0, 0.5 * 8.0 / 9.0, 0.5 * 5.0 / 9.0};
  const double nx = normal[e], ny = normal[nb + e];
  double un[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int64_t node = nodes[(int64_t)k * nb + e];
    un[k] = U[2 * node] * nx + U[2 * node + 1] * ny;
  }
  double m00 = 0.0, m01 = 0.0, m11 = 0.0;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const double lb = gx[q], la = 1.0 - lb;
    const double wn = la * (2 * la - 1) * un[0] + lb * (2 * lb - 1) * un[1] + 4 * la * lb * un[2];
    const double f = gw[q] * wn * length[e];
    m00 += f * la * la; m01 += f * la * lb; m11 += f * lb * lb;
  }
  loc[e] = m00; loc[(int64_t)nb + e] = m01; loc[(int64_t)2 * nb + e] = m01; loc[(int64_t)3 * nb + e] = m11;
}

// The same term in space: one thread per boundary FACE, the P2 wind from its
// six nodes (three vertices, then the midpoints of the edges 01, 02, 12),
// 6-point rule of degree 4 on the triangle (the integrand has degree 4), local
// 3 x 3 matrix loc[(i*3+j)][f] = |f| sum_q w_q (w.n)(q) lam_i(q) lam_j(q).
static __global__ __launch_bounds__(kBlock) void k_fe_robin_faces(
    int nb, const int* __restrict__ nodes, const double* __restrict__ normal,
    const double* __restrict__ area, const double* __restrict__ U, double* loc) {
  const int f = blockIdx.x * kBlock + threadIdx.x;
  if (f >= nb) return;
  const double qa[2] = {0.445948490915965, 0.091576213509771};
  const double qw[2] = {0.223381589678011, 0.109951743655322};
  double un[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int64_t node = nodes[(int64_t)k * nb + f];
    un[k] = U[3 * node] * normal[f] + U[3 * node + 1] * normal[nb + f] + U[3 * node + 2] * normal[2 * (int64_t)nb + f];
  }
  double m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = 0.0;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double L[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) L[i] = i == k ? 1.0 - 2.0 * qa[g] : qa[g];
      const double wn = L[0] * (2 * L[0] - 1) * un[0] + L[1] * (2 * L[1] - 1) * un[1] +
                        L[2] * (2 * L[2] - 1) * un[2] + 4 * L[0] * L[1] * un[3] +
                        4 * L[0] * L[2] * un[4] + 4 * L[1] * L[2] * un[5];
      const double w = qw[g] * wn * area[f];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) m[i * 3 + j] += w * L[i] * L[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) loc[(int64_t)i * nb + f] = m[i];
}

// out[pos[i]] += vals[i]   (positions distinct)
static __global__ __launch_bounds__(kBlock) void k_fe_add_at(
    int n, const int* __restrict__ pos, const double* __restrict__ vals, double* out) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    out[pos[i]] += vals[i];
}

// ---- nonlinear residual on the device ---------------------------------------
// v = x_u with the Dirichlet entries replaced by their boundary values g
// (v = x_u - d, d = the boundary defect of the iterate)
static __global__ __launch_bounds__(kBlock) void k_fe_bc_replace(
    int n, const int* __restrict__ idx, const double* __restrict__ g, double* v) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    v[idx[i]] = g[i];
}
// Dirichlet rows of the residual: F_u[idx] = mult * (x_u[idx] - g)
static __global__ __launch_bounds__(kBlock) void k_fe_bc_rows(
    int n, const int* __restrict__ idx, const double* __restrict__ g,
    const double* __restrict__ mult, const double* __restrict__ xu, double* Fu) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    Fu[idx[i]] = mult[i] * (xu[idx[i]] - g[i]);
}

// deterministic start vector of the power iteration
static __global__ __launch_bounds__(kBlock) void k_fe_seed(int64_t n, double* v) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
    z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    v[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}

}  // namespace pcd
