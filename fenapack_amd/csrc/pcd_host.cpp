// pcd_host.cpp - libpcd_host.so: host-side (OpenMP) set-up helpers of the PCD
// engine; C ABI in include/pcd_host.h.  Patterns, contribution lists, sub-matrix
// extraction, sparse products, sums of element contributions; no HIP.
//
// Build: g++ -O3 -fopenmp -std=c++17 -shared -fPIC pcd_host.cpp -o libpcd_host.so
#include "../../include/pcd_host.h"

#include <omp.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <utility>
#include <limits>
#include <vector>

static thread_local char g_err[512] = "";
static int g_threads = 0;

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

// Nothing throws across the ABI (include/pcd_host.h; SURVEY 8(b) "Errors").
// Every export is a function-try-block closed by PCDH_ABI_CATCH: bad_alloc ->
// PCDH_ERR_NOMEM, anything else -> PCDH_ERR_STATE, message in pcdh_last_error().
#define PCDH_ABI_CATCH(name)                                                         \
  catch (const std::bad_alloc&) { return fail(PCDH_ERR_NOMEM, #name ": out of host memory"); } \
  catch (const std::exception& e_) { return fail(PCDH_ERR_STATE, #name ": %s", e_.what()); }   \
  catch (...) { return fail(PCDH_ERR_STATE, #name ": unknown exception"); }

// An exception must not leave an OpenMP structured block either (that is
// std::terminate, not a status code): what allocates inside a parallel region
// - the threads' scratch, the rows' buffers - sits between OMP_TRY and
// OMP_CATCH, which record the failure in a flag of the team; the loops skip
// their remaining iterations (every thread still meets every worksharing
// construct) and the export returns PCDH_ERR_NOMEM after the region.
struct OmpFail {
  int v = 0;
  void set() {
#pragma omp atomic write
    v = 1;
  }
  bool on() const {
    int r;
#pragma omp atomic read
    r = v;
    return r != 0;
  }
};
#define OMP_TRY try {
#define OMP_CATCH(f) } catch (...) { (f).set(); }

// Threads for a job of `work` items: one per 32 k items, at most 32 (or what
// pcdh_set_threads asked for).  Small set-ups - the test suite builds
// thousands of tiny patterns - stay serial: waking a 256-thread team for a few
// thousand entries costs more than the work, and idle OpenMP workers spinning
// next to the other runtimes of the process (BLAS, torch) slow everything.
static int nthreads(int64_t work = INT64_MAX) {
  int cap = g_threads > 0 ? g_threads : std::min(omp_get_max_threads(), 32);
  int64_t t = std::min<int64_t>(cap, work / 32768);
  return (int)std::max<int64_t>(1, std::min<int64_t>(t, 256));
}

// ---------------------------------------------------------------- grouping
struct pcdh_group_s {
  int64_t n = 0, nrows_kept = 0, row0 = 0, nnz = 0, kept = 0;
  std::vector<int64_t> indptr;   // nrows_kept + 1
  std::vector<int64_t> ucols;    // nnz
  std::vector<int64_t> ptr;      // nnz + 1
  std::vector<int64_t> order;    // kept
};

// Stable grouping of items 0..n-1 by (row(i), col(i)); rows outside
// [row0, row1) are dropped.  Two-level counting sort: items are first
// partitioned (stably) into B contiguous row chunks with per-thread
// histograms of size B, then every chunk is counting-sorted by row and each
// row's short segment is sorted by column - all passes parallel, O(n) memory.
template <class RowF, class ColF>
static int group_impl(int64_t n, RowF row, ColF col, int64_t nrows, int64_t row0,
                      int64_t row1, pcdh_group_s* g) {
  const int T = nthreads(n);
  const int64_t nk = row1 - row0;
  g->n = n; g->nrows_kept = nk; g->row0 = row0;
  g->indptr.assign(nk + 1, 0);
  if (nk <= 0 || n <= 0) { g->ptr.assign(1, 0); return 0; }
  const int64_t B = std::max<int64_t>(1, std::min<int64_t>(nk, (int64_t)T * 8));
  const int64_t W = (nk + B - 1) / B;
  // pass 1: per-thread histograms over the chunks
  std::vector<int64_t> hist((size_t)T * B, 0);
  int bad = 0;
  // (a loop over the T input slices, not over omp_get_thread_num(): num_threads
  // is an upper bound - OMP_THREAD_LIMIT, a nested region or a failed thread
  // start give a smaller team, and every slice must still be processed)
#pragma omp parallel for schedule(static, 1) num_threads(T) reduction(| : bad)
  for (int t = 0; t < T; ++t) {
    const int64_t i0 = n * t / T, i1 = n * (t + 1) / T;
    int64_t* h = hist.data() + (size_t)t * B;
    for (int64_t i = i0; i < i1; ++i) {
      const int64_t r = row(i);
      if (r < 0 || r >= nrows) { bad |= 1; continue; }
      if (r < row0 || r >= row1) continue;
      ++h[(r - row0) / W];
    }
  }
  if (bad) return fail(PCDH_ERR_ARG, "group: row index outside [0, %lld)", (long long)nrows);
  // offsets: chunk-major, thread-minor (keeps ascending input position)
  std::vector<int64_t> chunk_off(B + 1, 0);
  {
    int64_t run = 0;
    for (int64_t b = 0; b < B; ++b) {
      chunk_off[b] = run;
      for (int t = 0; t < T; ++t) {
        const int64_t c = hist[(size_t)t * B + b];
        hist[(size_t)t * B + b] = run;
        run += c;
      }
    }
    chunk_off[B] = run;
  }
  const int64_t kept = chunk_off[B];
  g->kept = kept;
  std::vector<int64_t> part(kept), sorted(kept);
#pragma omp parallel for schedule(static, 1) num_threads(T)
  for (int t = 0; t < T; ++t) {
    const int64_t i0 = n * t / T, i1 = n * (t + 1) / T;
    int64_t* h = hist.data() + (size_t)t * B;
    for (int64_t i = i0; i < i1; ++i) {
      const int64_t r = row(i);
      if (r < row0 || r >= row1) continue;
      part[h[(r - row0) / W]++] = i;
    }
  }
  // pass 2: per chunk, counting sort by row, then sort each row by column
  std::vector<int64_t> rowlen(nk, 0);            // groups per row
  std::vector<std::vector<int64_t>> c_ucols(B), c_gsize(B);
  OmpFail oom;
#pragma omp parallel num_threads(T)
  {
    std::vector<int64_t> cnt;
    std::vector<std::pair<int64_t, int64_t>> seg;
#pragma omp for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; ++b) {
      const int64_t r_lo = b * W, r_hi = std::min<int64_t>(nk, r_lo + W);
      if (r_lo >= r_hi || oom.on()) continue;
      OMP_TRY
      const int64_t p0 = chunk_off[b], p1 = chunk_off[b + 1];
      cnt.assign(r_hi - r_lo + 1, 0);
      for (int64_t p = p0; p < p1; ++p) ++cnt[row(part[p]) - row0 - r_lo + 1];
      for (int64_t r = 0; r < r_hi - r_lo; ++r) cnt[r + 1] += cnt[r];
      {
        std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1);
        for (int64_t p = p0; p < p1; ++p) {
          const int64_t i = part[p];
          sorted[p0 + cur[row(i) - row0 - r_lo]++] = i;
        }
      }
      auto& uc = c_ucols[b];
      auto& gs = c_gsize[b];
      for (int64_t r = 0; r < r_hi - r_lo; ++r) {
        const int64_t s0 = p0 + cnt[r], s1 = p0 + cnt[r + 1];
        if (s0 == s1) continue;
        seg.resize(s1 - s0);
        for (int64_t p = s0; p < s1; ++p) seg[p - s0] = {col(sorted[p]), sorted[p]};
        // (col, position): positions are distinct, so this IS the stable order
        std::sort(seg.begin(), seg.end());
        int64_t groups = 0;
        for (int64_t p = s0; p < s1; ++p) {
          sorted[p] = seg[p - s0].second;
          if (p == s0 || seg[p - s0].first != seg[p - s0 - 1].first) {
            uc.push_back(seg[p - s0].first); gs.push_back(0); ++groups;
          }
          ++gs.back();
        }
        rowlen[r_lo + r] = groups;
      }
      OMP_CATCH(oom)
    }
  }
  if (oom.on()) return fail(PCDH_ERR_NOMEM, "group: out of host memory");
  for (int64_t r = 0; r < nk; ++r) g->indptr[r + 1] = g->indptr[r] + rowlen[r];
  const int64_t nnz = g->indptr[nk];
  g->nnz = nnz;
  g->ucols.resize(nnz);
  g->ptr.resize(nnz + 1);
  std::vector<int64_t> goff(B + 1, 0);
  for (int64_t b = 0; b < B; ++b) goff[b + 1] = goff[b] + (int64_t)c_ucols[b].size();
#pragma omp parallel for schedule(dynamic, 1) num_threads(T)
  for (int64_t b = 0; b < B; ++b) {
    int64_t run = chunk_off[b];
    for (size_t k = 0; k < c_ucols[b].size(); ++k) {
      g->ucols[goff[b] + k] = c_ucols[b][k];
      g->ptr[goff[b] + k] = run;
      run += c_gsize[b][k];
    }
  }
  g->ptr[nnz] = kept;
  g->order.swap(sorted);
  return 0;
}

extern "C" {

const char* pcdh_last_error(void) { return g_err; }


int pcdh_set_threads(int n) try {
  if (n < 0) return fail(PCDH_ERR_ARG, "set_threads: negative count");
  g_threads = n;
  return 0;
} PCDH_ABI_CATCH(pcdh_set_threads)
int pcdh_get_threads(void) { return nthreads(); }

int pcdh_group_pairs(int64_t n, const int64_t* rows, const int64_t* cols,
                     int64_t nrows, int64_t row0, int64_t row1, pcdh_group* out) try {
  if (!out || n < 0 || (n && !rows) || nrows < 0 || row0 < 0 || row1 > nrows || row0 > row1)
    return fail(PCDH_ERR_ARG, "group_pairs: bad arguments");
  pcdh_group_s* g = new (std::nothrow) pcdh_group_s();
  if (!g) return fail(PCDH_ERR_NOMEM, "group_pairs: out of memory");
  int rc;
  try {
    if (cols)
      rc = group_impl(n, [rows](int64_t i) { return rows[i]; },
                      [cols](int64_t i) { return cols[i]; }, nrows, row0, row1, g);
    else
      rc = group_impl(n, [rows](int64_t i) { return rows[i]; },
                      [](int64_t) { return (int64_t)0; }, nrows, row0, row1, g);
  } catch (const std::bad_alloc&) {
    rc = fail(PCDH_ERR_NOMEM, "group_pairs: out of memory");
  }
  if (rc) { delete g; return rc; }
  *out = g;
  return 0;
} PCDH_ABI_CATCH(pcdh_group_pairs)

int pcdh_pattern_cells(int64_t ncell, int nr, const int64_t* rdofs, int nc,
                       const int64_t* cdofs, int64_t nrows, int64_t row0,
                       int64_t row1, pcdh_group* out) try {
  if (!out || ncell < 0 || nr < 1 || nc < 1 || (ncell && (!rdofs || !cdofs)) ||
      nrows < 0 || row0 < 0 || row1 > nrows || row0 > row1)
    return fail(PCDH_ERR_ARG, "pattern_cells: bad arguments");
  pcdh_group_s* g = new (std::nothrow) pcdh_group_s();
  if (!g) return fail(PCDH_ERR_NOMEM, "pattern_cells: out of memory");
  const int64_t per = (int64_t)nr * nc;
  int rc;
  try {
    rc = group_impl(
        ncell * per,
        [=](int64_t i) { const int64_t c = i / per; return rdofs[c * nr + (i - c * per) / nc]; },
        [=](int64_t i) { const int64_t c = i / per; return cdofs[c * nc + (i - c * per) % nc]; },
        nrows, row0, row1, g);
  } catch (const std::bad_alloc&) {
    rc = fail(PCDH_ERR_NOMEM, "pattern_cells: out of memory");
  }
  if (rc) { delete g; return rc; }
  *out = g;
  return 0;
} PCDH_ABI_CATCH(pcdh_pattern_cells)

int64_t pcdh_group_nnz(pcdh_group g) { return g ? g->nnz : -1; }
int64_t pcdh_group_kept(pcdh_group g) { return g ? g->kept : -1; }

int pcdh_group_export(pcdh_group g, int64_t* indptr, int64_t* ucols, int64_t* inv,
                      int64_t* ptr, int64_t* order) try {
  if (!g) return fail(PCDH_ERR_ARG, "group_export: null group");
  const int T = nthreads(g->n);
  if (indptr) std::copy(g->indptr.begin(), g->indptr.end(), indptr);
  if (ucols) std::copy(g->ucols.begin(), g->ucols.end(), ucols);
  if (ptr) std::copy(g->ptr.begin(), g->ptr.end(), ptr);
  if (order) std::copy(g->order.begin(), g->order.end(), order);
  if (inv) {
#pragma omp parallel for schedule(static) num_threads(T)
    for (int64_t i = 0; i < g->n; ++i) inv[i] = -1;
#pragma omp parallel for schedule(static, 1024) num_threads(T)
    for (int64_t k = 0; k < g->nnz; ++k)
      for (int64_t p = g->ptr[k]; p < g->ptr[k + 1]; ++p) inv[g->order[p]] = k;
  }
  return 0;
} PCDH_ABI_CATCH(pcdh_group_export)

void pcdh_group_free(pcdh_group g) { delete g; }

// ------------------------------------------------------ sub-matrix extraction
int pcdh_extract_count(int64_t nr, const int32_t* rows, const int32_t* rowptr,
                       const int32_t* col, const int32_t* colmap, int32_t* orp) try {
  if (nr < 0 || !rows || !rowptr || !col || !colmap || !orp)
    return fail(PCDH_ERR_ARG, "extract_count: bad arguments");
  const int T = nthreads(nr * 8);
  orp[0] = 0;
#pragma omp parallel for schedule(static, 4096) num_threads(T)
  for (int64_t i = 0; i < nr; ++i) {
    int32_t c = 0;
    for (int32_t k = rowptr[rows[i]]; k < rowptr[rows[i] + 1]; ++k) c += colmap[col[k]] >= 0;
    orp[i + 1] = c;
  }
  int64_t run = 0;
  for (int64_t i = 0; i < nr; ++i) {
    run += orp[i + 1];
    if (run > INT32_MAX) return fail(PCDH_ERR_ARG, "extract_count: block exceeds int32 indexing");
    orp[i + 1] = (int32_t)run;
  }
  return 0;
} PCDH_ABI_CATCH(pcdh_extract_count)

int pcdh_extract_fill(int64_t nr, const int32_t* rows, const int32_t* rowptr,
                      const int32_t* col, const int32_t* colmap, const int32_t* orp,
                      int32_t* oc, int64_t* osrc) try {
  if (nr < 0 || !rows || !rowptr || !col || !colmap || !orp || (orp[nr] && (!oc || !osrc)))
    return fail(PCDH_ERR_ARG, "extract_fill: bad arguments");
  const int T = nthreads(nr * 8);
  OmpFail oom;
#pragma omp parallel num_threads(T)
  {
    std::vector<std::pair<int32_t, int64_t>> tmp;
#pragma omp for schedule(static, 4096)
    for (int64_t i = 0; i < nr; ++i) {
      if (oom.on()) continue;
      OMP_TRY
      tmp.clear();
      bool sorted = true;
      for (int32_t k = rowptr[rows[i]]; k < rowptr[rows[i] + 1]; ++k) {
        const int32_t c = colmap[col[k]];
        if (c < 0) continue;
        if (!tmp.empty() && c < tmp.back().first) sorted = false;
        tmp.emplace_back(c, (int64_t)k);
      }
      if (!sorted) std::sort(tmp.begin(), tmp.end());
      int64_t q = orp[i];
      for (auto& t : tmp) { oc[q] = t.first; osrc[q] = t.second; ++q; }
      OMP_CATCH(oom)
    }
  }
  if (oom.on()) return fail(PCDH_ERR_NOMEM, "extract_fill: out of host memory");
  return 0;
} PCDH_ABI_CATCH(pcdh_extract_fill)

// -------------------------------------------------------------------- transpose
int pcdh_transpose(int64_t nr, int64_t nc, const int32_t* rp, const int32_t* ci,
                   const double* va, int32_t* trp, int32_t* tc, double* tv) try {
  if (nr < 0 || nc < 0 || !rp || !trp || (rp[nr] && (!ci || !tc)))
    return fail(PCDH_ERR_ARG, "transpose: bad arguments");
  const int64_t nnz = rp[nr];
  for (int64_t c = 0; c <= nc; ++c) trp[c] = 0;
  for (int64_t k = 0; k < nnz; ++k) {
    if (ci[k] < 0 || ci[k] >= nc) return fail(PCDH_ERR_ARG, "transpose: column outside [0, %lld)", (long long)nc);
    ++trp[ci[k] + 1];
  }
  for (int64_t c = 0; c < nc; ++c) trp[c + 1] += trp[c];
  std::vector<int32_t> fill(trp, trp + nc);
  for (int64_t i = 0; i < nr; ++i)
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
      const int32_t q = fill[ci[k]]++;
      tc[q] = (int32_t)i;
      if (va && tv) tv[q] = va[k];
    }
  return 0;
} PCDH_ABI_CATCH(pcdh_transpose)

// ----------------------------------------------------------------------- SpGEMM
int pcdh_spgemm_count(int64_t row0, int64_t row1, int64_t b_cols,
                      const int32_t* arp, const int32_t* ac, const int32_t* brp,
                      const int32_t* bc, int64_t* crp) try {
  if (row0 < 0 || row1 < row0 || b_cols < 0 || !arp || !brp || !crp)
    return fail(PCDH_ERR_ARG, "spgemm_count: bad arguments");
  const int T = nthreads((row1 - row0) * 64);
  crp[0] = 0;
  OmpFail oom;
#pragma omp parallel num_threads(T)
  {
    std::vector<int32_t> mark;
    OMP_TRY mark.assign(b_cols, -1); OMP_CATCH(oom)
#pragma omp for schedule(dynamic, 256)
    for (int64_t i = row0; i < row1; ++i) {
      if (oom.on()) continue;
      int64_t cnt = 0;
      for (int32_t k = arp[i]; k < arp[i + 1]; ++k) {
        const int32_t j = ac[k];
        for (int32_t q = brp[j]; q < brp[j + 1]; ++q)
          if (mark[bc[q]] != (int32_t)(i - row0)) { mark[bc[q]] = (int32_t)(i - row0); ++cnt; }
      }
      crp[i - row0 + 1] = cnt;
    }
  }
  if (oom.on()) return fail(PCDH_ERR_NOMEM, "spgemm_count: out of host memory");
  for (int64_t i = 0; i < row1 - row0; ++i) crp[i + 1] += crp[i];
  return 0;
} PCDH_ABI_CATCH(pcdh_spgemm_count)

int pcdh_spgemm_fill(int64_t row0, int64_t row1, int64_t b_cols, const int32_t* arp,
                     const int32_t* ac, const double* av, const int32_t* brp,
                     const int32_t* bc, const double* bv, const int64_t* crp,
                     int32_t* cc, double* cv) try {
  if (row0 < 0 || row1 < row0 || !arp || !brp || !crp || (crp[row1 - row0] && (!cc || !cv)))
    return fail(PCDH_ERR_ARG, "spgemm_fill: bad arguments");
  const int T = nthreads((row1 - row0) * 64);
  OmpFail oom;
#pragma omp parallel num_threads(T)
  {
    std::vector<int64_t> where;                 // column -> slot of the current row
    std::vector<std::pair<int32_t, double>> rowbuf;
    OMP_TRY where.assign(b_cols, -1); OMP_CATCH(oom)
#pragma omp for schedule(dynamic, 256)
    for (int64_t i = row0; i < row1; ++i) {
      if (oom.on()) continue;
      OMP_TRY
      const int64_t base = crp[i - row0];
      int64_t fillp = base;
      for (int32_t k = arp[i]; k < arp[i + 1]; ++k) {
        const int32_t j = ac[k];
        const double a = av ? av[k] : 1.0;
        for (int32_t q = brp[j]; q < brp[j + 1]; ++q) {
          const int32_t c = bc[q];
          const double v = a * (bv ? bv[q] : 1.0);
          if (where[c] < base) { where[c] = fillp; cc[fillp] = c; cv[fillp] = v; ++fillp; }
          else cv[where[c]] += v;
        }
      }
      // sort the row by column (accumulation order per entry stays k-major)
      const int64_t len = fillp - base;
      rowbuf.resize(len);
      for (int64_t p = 0; p < len; ++p) rowbuf[p] = {cc[base + p], cv[base + p]};
      std::sort(rowbuf.begin(), rowbuf.end(),
                [](const std::pair<int32_t, double>& x, const std::pair<int32_t, double>& y) { return x.first < y.first; });
      for (int64_t p = 0; p < len; ++p) { cc[base + p] = rowbuf[p].first; cv[base + p] = rowbuf[p].second; where[rowbuf[p].first] = -1; }
      OMP_CATCH(oom)
    }
  }
  if (oom.on()) return fail(PCDH_ERR_NOMEM, "spgemm_fill: out of host memory");
  return 0;
} PCDH_ABI_CATCH(pcdh_spgemm_fill)

// ---------------------------------------------------- plan of a sparse product
// C = A B on a fixed pattern, as a GATHER PLAN: for every entry of C the list
// of products that sum to it, so that later numeric products are one weighted
// gather on the device (pcd_fe_set_level_galerkin).  Entry (i, J) of C lists
// its terms in the order of A's row i (then B's row): the order the numpy
// builder's stable grouping yields.  mode 0: term = (index of the A entry,
// weight = the B value) - B = F P with F changing, P fixed; mode 1: term =
// (index of the B entry, weight = the A value) - F_c = P^T B with B changing.
int pcdh_product_plan_count(int64_t nrows, int64_t b_cols, const int32_t* arp,
                            const int32_t* ac, const int32_t* brp, const int32_t* bc,
                            int64_t* crp, int64_t* trp) try {
  if (nrows < 0 || b_cols < 0 || !arp || !brp || !crp || !trp)
    return fail(PCDH_ERR_ARG, "product_plan_count: bad arguments");
  const int T = nthreads(nrows * 64);
  crp[0] = 0; trp[0] = 0;
  OmpFail oom;
#pragma omp parallel num_threads(T)
  {
    std::vector<int64_t> mark;
    OMP_TRY mark.assign(b_cols, -1); OMP_CATCH(oom)
#pragma omp for schedule(dynamic, 256)
    for (int64_t i = 0; i < nrows; ++i) {
      if (oom.on()) continue;
      int64_t cnt = 0, terms = 0;
      for (int32_t k = arp[i]; k < arp[i + 1]; ++k) {
        const int32_t j = ac[k];
        terms += brp[j + 1] - brp[j];
        for (int32_t q = brp[j]; q < brp[j + 1]; ++q)
          if (mark[bc[q]] != i) { mark[bc[q]] = i; ++cnt; }
      }
      crp[i + 1] = cnt; trp[i + 1] = terms;
    }
  }
  if (oom.on()) return fail(PCDH_ERR_NOMEM, "product_plan_count: out of host memory");
  for (int64_t i = 0; i < nrows; ++i) { crp[i + 1] += crp[i]; trp[i + 1] += trp[i]; }
  return 0;
} PCDH_ABI_CATCH(pcdh_product_plan_count)

int pcdh_product_plan_fill(int64_t nrows, int64_t b_cols, const int32_t* arp, const int32_t* ac,
                           const double* av, const int32_t* brp, const int32_t* bc,
                           const double* bv, int mode, const int64_t* crp, const int64_t* trp,
                           int32_t* cc, int64_t* ptr, int32_t* src, double* w) try {
  if (nrows < 0 || !arp || !brp || !crp || !trp || !ptr || (mode != 0 && mode != 1) ||
      (mode == 0 ? !bv : !av) || (crp[nrows] && !cc) || (trp[nrows] && (!src || !w)))
    return fail(PCDH_ERR_ARG, "product_plan_fill: bad arguments");
  const int T = nthreads(nrows * 64);
  (void)b_cols;
  OmpFail oom;
#pragma omp parallel num_threads(T)
  {
    struct Term { int32_t col, src; double w; };
    std::vector<Term> buf;
#pragma omp for schedule(dynamic, 256)
    for (int64_t i = 0; i < nrows; ++i) {
      if (oom.on()) continue;
      OMP_TRY
      buf.clear();
      for (int32_t k = arp[i]; k < arp[i + 1]; ++k) {
        const int32_t j = ac[k];
        for (int32_t q = brp[j]; q < brp[j + 1]; ++q)
          buf.push_back(Term{bc[q], mode == 0 ? k : q, mode == 0 ? bv[q] : av[k]});
      }
      std::stable_sort(buf.begin(), buf.end(), [](const Term& x, const Term& y) { return x.col < y.col; });
      int64_t e = crp[i], t = trp[i];
      for (size_t p = 0; p < buf.size(); ++p) {
        if (p == 0 || buf[p].col != buf[p - 1].col) { cc[e] = buf[p].col; ptr[e] = t; ++e; }
        src[t] = buf[p].src; w[t] = buf[p].w; ++t;
      }
      OMP_CATCH(oom)
    }
  }
  if (oom.on()) return fail(PCDH_ERR_NOMEM, "product_plan_fill: out of host memory");
  ptr[crp[nrows]] = trp[nrows];
  return 0;
} PCDH_ABI_CATCH(pcdh_product_plan_fill)

// ----------------------------------------------------------------------- SpMV
int pcdh_spmv(int64_t nrows, const int32_t* rowptr, const int32_t* col, const double* val,
              const double* x, const double* scale, double* y) try {
  if (nrows < 0 || !rowptr || (rowptr[nrows] && (!col || !val)) || !x || !y)
    return fail(PCDH_ERR_ARG, "spmv: bad arguments");
  const int T = nthreads((int64_t)rowptr[nrows]);
#pragma omp parallel for schedule(static, 2048) num_threads(T)
  for (int64_t i = 0; i < nrows; ++i) {
    double s = 0.0;                        // (ascending k: scipy's csr_matvec order)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += val[k] * x[col[k]];
    y[i] = scale ? scale[i] * s : s;
  }
  return 0;
} PCDH_ABI_CATCH(pcdh_spmv)

// y = scale .* (A X) for nvec interleaved vectors (X, Y: nrows x nvec row-major):
// the power iteration of F (x) I_nvec without the expanded matrix; each of the
// nvec row sums runs in ascending entry order (= the expanded operator's sums)
int pcdh_spmm(int64_t nrows, const int32_t* rowptr, const int32_t* col, const double* val,
              int nvec, const double* x, const double* scale, double* y) try {
  if (nrows < 0 || nvec < 1 || nvec > 8 || !rowptr || (rowptr[nrows] && (!col || !val)) || !x || !y)
    return fail(PCDH_ERR_ARG, "spmm: bad arguments");
  const int T = nthreads((int64_t)rowptr[nrows] * nvec);
#pragma omp parallel for schedule(static, 2048) num_threads(T)
  for (int64_t i = 0; i < nrows; ++i) {
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const double a = val[k];
      const double* xr = x + (int64_t)col[k] * nvec;
      for (int c = 0; c < nvec; ++c) s[c] += a * xr[c];
    }
    for (int c = 0; c < nvec; ++c) y[i * nvec + c] = scale ? scale[i * nvec + c] * s[c] : s[c];
  }
  return 0;
} PCDH_ABI_CATCH(pcdh_spmm)

// ------------------------------------------------------- F (x) I_nc, both ways
int pcdh_kron_factor(int64_t nrows, const int32_t* rowptr, const int32_t* col, const double* val,
                     int nc, int32_t* f_rowptr, int32_t* f_col, double* f_val) try {
  if (nrows < 0 || nc < 2 || !rowptr || !f_rowptr) return fail(PCDH_ERR_ARG, "kron_factor: bad arguments");
  if (nrows % nc || rowptr[nrows] % nc) return PCDH_NOT_KRON;
  const int64_t ns = nrows / nc;
  const int T = nthreads((int64_t)rowptr[nrows]);
  int bad = 0;
#pragma omp parallel for schedule(static, 2048) num_threads(T) reduction(| : bad)
  for (int64_t s = 0; s < ns; ++s) {
    const int32_t a = rowptr[nc * s], len = rowptr[nc * s + 1] - a;
    // component rows of one node are stored back to back: row c starts at a + c len
    for (int c = 1; c < nc; ++c)
      if (rowptr[nc * s + c] != a + c * len || rowptr[nc * s + c + 1] - rowptr[nc * s + c] != len) bad = 1;
    if (bad) continue;
    for (int32_t k = 0; k < len; ++k) {
      if (col[a + k] % nc) { bad = 1; break; }
      for (int c = 1; c < nc; ++c)
        if (col[a + c * len + k] != col[a + k] + c || val[a + c * len + k] != val[a + k]) bad = 1;
    }
  }
  if (bad) return PCDH_NOT_KRON;
  for (int64_t s = 0; s <= ns; ++s) f_rowptr[s] = rowptr[s < ns ? nc * s : nrows] / nc;
  if (!f_col || !f_val) return 0;
#pragma omp parallel for schedule(static, 2048) num_threads(T)
  for (int64_t s = 0; s < ns; ++s) {
    const int32_t a = rowptr[nc * s], len = rowptr[nc * s + 1] - a, o = f_rowptr[s];
    for (int32_t k = 0; k < len; ++k) { f_col[o + k] = col[a + k] / nc; f_val[o + k] = val[a + k]; }
  }
  return 0;
} PCDH_ABI_CATCH(pcdh_kron_factor)

int pcdh_kron_expand(int64_t ns, const int32_t* f_rowptr, const int32_t* f_col, const double* f_val,
                     int nc, int32_t* rowptr, int32_t* col, double* val) try {
  if (ns < 0 || nc < 1 || !f_rowptr || !rowptr || (f_rowptr[ns] && (!f_col || !f_val || !col || !val)))
    return fail(PCDH_ERR_ARG, "kron_expand: bad arguments");
  if ((int64_t)f_rowptr[ns] * nc > INT32_MAX) return fail(PCDH_ERR_ARG, "kron_expand: more than 2^31 entries");
  const int T = nthreads((int64_t)f_rowptr[ns] * nc);
#pragma omp parallel for schedule(static, 2048) num_threads(T)
  for (int64_t s = 0; s < ns; ++s) {
    const int32_t o = f_rowptr[s], len = f_rowptr[s + 1] - o;
    for (int c = 0; c < nc; ++c) {
      const int32_t a = nc * o + c * len;
      rowptr[nc * s + c] = a;
      for (int32_t k = 0; k < len; ++k) { col[a + k] = nc * f_col[o + k] + c; val[a + k] = f_val[o + k]; }
    }
  }
  rowptr[nc * ns] = nc * f_rowptr[ns];
  return 0;
} PCDH_ABI_CATCH(pcdh_kron_expand)

// ------------------------------------------------- sums of grouped contributions
// out[g] = sum of vals[members[k]], k in [ptr[g], ptr[g+1]), added in ascending k
// (members ascend within a group: numpy.bincount's order of additions)
int pcdh_gather_sum(int64_t ngroups, const int64_t* ptr, const int64_t* members,
                    const double* vals, double* out) try {
  if (ngroups < 0 || !ptr || (ptr[ngroups] && (!members || !vals)) || !out)
    return fail(PCDH_ERR_ARG, "gather_sum: bad arguments");
  const int T = nthreads(ptr[ngroups]);
#pragma omp parallel for schedule(static, 4096) num_threads(T)
  for (int64_t g = 0; g < ngroups; ++g) {
    double s = 0.0;
    for (int64_t k = ptr[g]; k < ptr[g + 1]; ++k) s += vals[members[k]];
    out[g] = s;
  }
  return 0;
} PCDH_ABI_CATCH(pcdh_gather_sum)

// ------------------------------------------ wind x barycentric gradients per cell
// out[c, m, k] = |T_c| * sum_d U[dofs[c, m], d] * gradlam[c, k, d], the sum in
// ascending d with every product rounded before it is added (numpy's
// elementwise chain, bitwise): the left factor of the convection element
// matrices (taylor_hood.py p2_convection_nodal), one pass instead of eight
int pcdh_wind_gradlam(int64_t ncell, int na, int nvl, int dim, const int64_t* dofs,
                      const double* U, const double* gradlam, const double* area,
                      double* out) try {
  if (ncell < 0 || na < 1 || nvl < 1 || dim < 1 || dim > 3 ||
      (ncell && (!dofs || !U || !gradlam || !area || !out)))
    return fail(PCDH_ERR_ARG, "wind_gradlam: bad arguments");
  const int T = nthreads(ncell * na * nvl);
#pragma omp parallel for schedule(static, 2048) num_threads(T)
  for (int64_t c = 0; c < ncell; ++c) {
    const double* g = gradlam + c * nvl * dim;
    const double a = area[c];
    for (int m = 0; m < na; ++m) {
      const double* u = U + dofs[c * na + m] * dim;
      double* o = out + (c * na + m) * nvl;
      for (int k = 0; k < nvl; ++k) {
        volatile double p = u[0] * g[k * dim];          // (volatile: no contraction
        double s = p;                                   //  into a fused multiply-add)
        for (int d = 1; d < dim; ++d) {
          p = u[d] * g[k * dim + d];
          s += p;
        }
        o[k] = s * a;
      }
    }
  }
  return 0;
} PCDH_ABI_CATCH(pcdh_wind_gradlam)

// ------------------------------------------------------- union of mapped blocks
int pcdh_union_count(int64_t n, int nb, const int64_t* nr, const int32_t* const* rowmap,
                     const int32_t* const* indptr, int64_t* out) try {
  if (n < 0 || nb < 1 || !nr || !rowmap || !indptr || !out)
    return fail(PCDH_ERR_ARG, "union_count: bad arguments");
  for (int64_t i = 0; i <= n; ++i) out[i] = 0;
  for (int b = 0; b < nb; ++b)
    for (int64_t i = 0; i < nr[b]; ++i) {
      const int32_t g = rowmap[b][i];
      if (g < 0 || g >= n) return fail(PCDH_ERR_ARG, "union_count: row map of block %d leaves [0, %lld)", b, (long long)n);
      out[g + 1] += indptr[b][i + 1] - indptr[b][i];
    }
  for (int64_t i = 0; i < n; ++i) out[i + 1] += out[i];
  return 0;
} PCDH_ABI_CATCH(pcdh_union_count)

int pcdh_union_fill(int64_t n, int nb, const int64_t* nr, const int32_t* const* rowmap,
                    const int32_t* const* colmap, const int32_t* const* indptr,
                    const int32_t* const* indices, const int64_t* data_off,
                    const int64_t* out_indptr, int32_t* out_indices, int64_t* out_order) try {
  if (n < 0 || nb < 1 || !nr || !rowmap || !colmap || !indptr || !indices || !data_off ||
      !out_indptr || !out_indices || !out_order)
    return fail(PCDH_ERR_ARG, "union_fill: bad arguments");
  const int T = nthreads(n * 16);
  // global row -> (block, block row) lists: a row may appear in several blocks
  std::vector<int32_t> inv((size_t)nb * n, -1);
  for (int b = 0; b < nb; ++b)
    for (int64_t i = 0; i < nr[b]; ++i) inv[(size_t)b * n + rowmap[b][i]] = (int32_t)i;
  int dup = 0;
  OmpFail oom;
#pragma omp parallel num_threads(T)
  {
    std::vector<std::pair<int32_t, int64_t>> tmp;
#pragma omp for schedule(static, 2048)
    for (int64_t g = 0; g < n; ++g) {
      if (oom.on()) continue;
      OMP_TRY
      tmp.clear();
      for (int b = 0; b < nb; ++b) {
        const int32_t i = inv[(size_t)b * n + g];
        if (i < 0) continue;
        for (int32_t k = indptr[b][i]; k < indptr[b][i + 1]; ++k)
          tmp.emplace_back(colmap[b][indices[b][k]], data_off[b] + k);
      }
      std::sort(tmp.begin(), tmp.end());
      int64_t q = out_indptr[g];
      for (size_t t = 0; t < tmp.size(); ++t) {
        if (t && tmp[t].first == tmp[t - 1].first) dup = 1;
        out_indices[q] = tmp[t].first; out_order[q] = tmp[t].second; ++q;
      }
      OMP_CATCH(oom)
    }
  }
  if (oom.on()) return fail(PCDH_ERR_NOMEM, "union_fill: out of host memory");
  if (dup) return fail(PCDH_ERR_ARG, "union_fill: the blocks overlap");
  return 0;
} PCDH_ABI_CATCH(pcdh_union_fill)

// ------------------------------------------------------------ gathers of values
// out[i] = concat(seg[0], seg[1], ...)[idx[i]] without forming the
// concatenation: seg_off[s] = first index of segment s (nseg + 1 entries).
// The monolithic system's values from its blocks (an assembly that is a pure
// permutation) and a sub-matrix's values from its parent's, on threads: what
// numpy does with concatenate + fancy indexing on ONE thread (at config 5's
// size 1.1e9 entries per call, twice per nonlinear step).
int pcdh_take_segments(int64_t n, const int64_t* idx, int nseg, const double* const* seg,
                       const int64_t* seg_off, double* out) try {
  if (n < 0 || nseg < 1 || nseg > 16 || !seg || !seg_off || (n && (!idx || !out)))
    return fail(PCDH_ERR_ARG, "take_segments: bad arguments");
  for (int s_ = 0; s_ < nseg; ++s_)
    if (seg_off[s_ + 1] < seg_off[s_] || (seg_off[s_ + 1] > seg_off[s_] && !seg[s_]))
      return fail(PCDH_ERR_ARG, "take_segments: bad segment %d", s_);
  const int64_t total = seg_off[nseg];
  const int T = nthreads(n);
  int bad = 0;
#pragma omp parallel for schedule(static, 1 << 16) num_threads(T) reduction(| : bad)
  for (int64_t i = 0; i < n; ++i) {
    const int64_t k = idx[i];
    if (k < seg_off[0] || k >= total) { bad |= 1; out[i] = 0.0; continue; }
    int s_ = 0;
    while (k >= seg_off[s_ + 1]) ++s_;
    out[i] = seg[s_][k - seg_off[s_]];
  }
  if (bad) return fail(PCDH_ERR_ARG, "take_segments: an index outside the segments");
  return 0;
} PCDH_ABI_CATCH(pcdh_take_segments)

// ------------------------------------------------------- positions of entries
// pos[q] = position of entry (qrow[q], qcol[q]) in the CSR (rowptr, col) whose
// columns are sorted in every row; PCDH_ERR_ARG when one is missing.
int pcdh_locate(int64_t nq, const int64_t* qrow, const int64_t* qcol, int64_t nrows,
                const int64_t* rowptr, const int32_t* col, int64_t* pos) try {
  if (nq < 0 || nrows < 0 || !rowptr || (nq && (!qrow || !qcol || !pos)) || (rowptr[nrows] && !col))
    return fail(PCDH_ERR_ARG, "locate: bad arguments");
  const int T = nthreads(nq * 4);
  int bad = 0;
#pragma omp parallel for schedule(static, 8192) num_threads(T) reduction(| : bad)
  for (int64_t q = 0; q < nq; ++q) {
    const int64_t r = qrow[q];
    if (r < 0 || r >= nrows) { bad |= 1; pos[q] = -1; continue; }
    const int32_t* b = col + rowptr[r];
    const int32_t* e = col + rowptr[r + 1];
    const int32_t* p = std::lower_bound(b, e, (int32_t)qcol[q]);
    if (p == e || *p != qcol[q] || qcol[q] > INT32_MAX) { bad |= 1; pos[q] = -1; continue; }
    pos[q] = (int64_t)(p - col);
  }
  if (bad) return fail(PCDH_ERR_ARG, "locate: an entry is not in the pattern");
  return 0;
} PCDH_ABI_CATCH(pcdh_locate)

// pos[p * nq + q] = position of the entry (is_u[d rows[q] + ci[p]],
// is_u[d cols[q] + cj[p]]) in the CSR (rowptr, col): where the d (Picard:
// ci = cj = 0 .. d-1) or d*d (Newton) component entries of the scalar pattern
// entry q sit in the monolithic system's values - locate() for all components
// at once, without the index arrays numpy would build for each of them.
int pcdh_locate_blocks(int64_t nq, const int32_t* rows, const int32_t* cols, int d, int npairs,
                       const int32_t* ci, const int32_t* cj, int64_t n_u, const int64_t* is_u,
                       int64_t nrows, const int64_t* rowptr, const int32_t* col, int64_t* pos) try {
  if (nq < 0 || d < 1 || npairs < 1 || !ci || !cj || !is_u || !rowptr || (nq && (!rows || !cols || !pos)) ||
      (rowptr[nrows] && !col))
    return fail(PCDH_ERR_ARG, "locate_blocks: bad arguments");
  for (int p_ = 0; p_ < npairs; ++p_)
    if (ci[p_] < 0 || ci[p_] >= d || cj[p_] < 0 || cj[p_] >= d)
      return fail(PCDH_ERR_ARG, "locate_blocks: component outside [0, d)");
  const int T = nthreads(nq * 4 * npairs);
  int bad = 0;
#pragma omp parallel for schedule(static, 8192) num_threads(T) reduction(| : bad)
  for (int64_t q = 0; q < nq; ++q) {
    for (int p_ = 0; p_ < npairs; ++p_) {
      const int64_t ur = (int64_t)d * rows[q] + ci[p_], uc = (int64_t)d * cols[q] + cj[p_];
      int64_t& out = pos[(int64_t)p_ * nq + q];
      out = -1;
      if (rows[q] < 0 || cols[q] < 0 || ur >= n_u || uc >= n_u) { bad |= 1; continue; }
      const int64_t r = is_u[ur], c = is_u[uc];
      if (r < 0 || r >= nrows || c > INT32_MAX) { bad |= 1; continue; }
      const int32_t* b = col + rowptr[r];
      const int32_t* e = col + rowptr[r + 1];
      const int32_t* f = std::lower_bound(b, e, (int32_t)c);
      if (f == e || *f != c) { bad |= 1; continue; }
      out = (int64_t)(f - col);
    }
  }
  if (bad) return fail(PCDH_ERR_ARG, "locate_blocks: an entry is not in the pattern");
  return 0;
} PCDH_ABI_CATCH(pcdh_locate_blocks)

// src[t] = (order[t] % nloc2) * ncells + order[t] / nloc2: the element storage
// position (component-major: ab * ncells + cell) of every member of a
// contribution list whose members are element entries laid out (cell, ab).
int pcdh_contribution_src(int64_t n, const int64_t* order, int64_t nloc2, int64_t ncells, int32_t* src) try {
  if (n < 0 || nloc2 < 1 || ncells < 1 || (n && (!order || !src)))
    return fail(PCDH_ERR_ARG, "contribution_src: bad arguments");
  if (nloc2 * ncells > INT32_MAX) return fail(PCDH_ERR_ARG, "contribution_src: element storage exceeds int32 indexing");
  const int T = nthreads(n);
  int bad = 0;
#pragma omp parallel for schedule(static, 1 << 16) num_threads(T) reduction(| : bad)
  for (int64_t t = 0; t < n; ++t) {
    const int64_t o = order[t];
    if (o < 0 || o >= nloc2 * ncells) { bad |= 1; src[t] = 0; continue; }
    src[t] = (int32_t)((o % nloc2) * ncells + o / nloc2);
  }
  if (bad) return fail(PCDH_ERR_ARG, "contribution_src: a member outside the element storage");
  return 0;
} PCDH_ABI_CATCH(pcdh_contribution_src)

// ------------------------------------------------- distance-2 independent set
int pcdh_mis2_degrees(int64_t n, const int32_t* rowptr, const int32_t* col, int64_t* deg) try {
  if (n < 0 || !rowptr || (rowptr[n] && !col) || (n && !deg))
    return fail(PCDH_ERR_ARG, "mis2_degrees: bad arguments");
  if (n > INT32_MAX) return fail(PCDH_ERR_ARG, "mis2_degrees: more than 2^31 vertices");
  const int T = nthreads((int64_t)rowptr[n] * 8);
  OmpFail oom;
#pragma omp parallel num_threads(T)
  {
    // stamp[k] == i: vertex k already counted for vertex i
    std::vector<int32_t> stamp;
    OMP_TRY stamp.assign((size_t)n, -1); OMP_CATCH(oom)
#pragma omp for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; ++i) {
      if (oom.on()) continue;
      int64_t cnt = 0;
      auto visit = [&](int32_t j) {
        for (int32_t q = rowptr[j]; q < rowptr[j + 1]; ++q) {
          const int32_t k = col[q];
          if (k != i && stamp[k] != (int32_t)i) { stamp[k] = (int32_t)i; ++cnt; }
        }
      };
      visit((int32_t)i);
      for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
        const int32_t j = col[p];
        if (j == i) continue;
        // (j itself is within one edge: counted by visit(i) above)
        visit(j);
      }
      deg[i] = cnt;
    }
  }
  if (oom.on()) return fail(PCDH_ERR_NOMEM, "mis2_degrees: out of host memory");
  return 0;
} PCDH_ABI_CATCH(pcdh_mis2_degrees)

int pcdh_mis2(int64_t n, const int32_t* rowptr, const int32_t* col, const double* w,
              int8_t* in_set, int64_t* rounds) try {
  if (n < 0 || !rowptr || (rowptr[n] && !col) || (n && (!w || !in_set)))
    return fail(PCDH_ERR_ARG, "mis2: bad arguments");
  if (n > INT32_MAX) return fail(PCDH_ERR_ARG, "mis2: more than 2^31 vertices");
  const double ninf = -std::numeric_limits<double>::infinity();
  const int T = nthreads((int64_t)rowptr[n] * 4);
  std::vector<int8_t> state((size_t)n, 0);          // 0 undecided, 1 in, -1 out
  // the two largest priorities of DISTINCT undecided vertices in the closed
  // one-hop neighbourhood of every vertex (v1 >= v2; i1 / i2 = -1: none)
  std::vector<double> v1((size_t)n), v2((size_t)n);
  std::vector<int32_t> i1((size_t)n), i2((size_t)n);
  std::vector<int8_t> win((size_t)n), near((size_t)n);
  int64_t nround = 0;
  for (;;) {
    int64_t nund = 0;
#pragma omp parallel for schedule(static, 4096) num_threads(T) reduction(+ : nund)
    for (int64_t i = 0; i < n; ++i) nund += state[i] == 0;
    if (!nund) break;
    ++nround;
#pragma omp parallel for schedule(static, 2048) num_threads(T)
    for (int64_t j = 0; j < n; ++j) {
      double a1 = ninf, a2 = ninf;
      int32_t b1 = -1, b2 = -1;
      auto offer = [&](int32_t k) {
        const double x = state[k] == 0 ? w[k] : ninf;
        // (ties keep the first comer: which of two equal candidates is "the
        // largest" does not matter, both block each other below)
        if (b1 < 0 || x > a1) { a2 = a1; b2 = b1; a1 = x; b1 = k; }
        else if (b2 < 0 || x > a2) { a2 = x; b2 = k; }
      };
      offer((int32_t)j);
      for (int32_t q = rowptr[j]; q < rowptr[j + 1]; ++q)
        if (col[q] != j) offer(col[q]);
      v1[j] = a1; i1[j] = b1; v2[j] = b2 < 0 ? ninf : a2; i2[j] = b2;
    }
    int64_t nwin = 0;
#pragma omp parallel for schedule(static, 2048) num_threads(T) reduction(+ : nwin)
    for (int64_t i = 0; i < n; ++i) {
      win[i] = 0;
      if (state[i] != 0) continue;
      // max over the two-hop neighbourhood, the vertex itself left out
      auto other = [&](int32_t j) { return i1[j] != (int32_t)i ? v1[j] : v2[j]; };
      double nb = other((int32_t)i);
      for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p)
        if (col[p] != i) nb = std::max(nb, other(col[p]));
      if (w[i] > nb) { win[i] = 1; ++nwin; }
    }
    if (!nwin) {                               // tied priorities: break by index
      int64_t best = -1;
      for (int64_t i = 0; i < n; ++i)
        if (state[i] == 0 && (best < 0 || w[i] > w[best])) best = i;
      win[best] = 1;
    }
    // vertices within two edges of a new member drop out
#pragma omp parallel for schedule(static, 2048) num_threads(T)
    for (int64_t j = 0; j < n; ++j) {
      int8_t any = win[j];
      for (int32_t q = rowptr[j]; q < rowptr[j + 1] && !any; ++q) any = win[col[q]];
      near[j] = any;
    }
#pragma omp parallel for schedule(static, 2048) num_threads(T)
    for (int64_t i = 0; i < n; ++i) {
      if (win[i]) { state[i] = 1; continue; }
      if (state[i] != 0) continue;
      int8_t any = near[i];
      for (int32_t p = rowptr[i]; p < rowptr[i + 1] && !any; ++p) any = near[col[p]];
      if (any) state[i] = -1;
    }
  }
  for (int64_t i = 0; i < n; ++i) in_set[i] = state[i] == 1;
  if (rounds) *rounds = nround;
  return 0;
} PCDH_ABI_CATCH(pcdh_mis2)

}  // extern "C"
