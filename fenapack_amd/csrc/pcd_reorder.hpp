// pcd_reorder.hpp - locality-preserving renumbering inside the engine.
//
// The kernels gather the input vector through the column indices: how many
// cache lines a row block touches is decided by the numbering the caller
// hands over.  The repository's own producer numbers nodes lexicographically
// (geometric slabs); a caller's dof order (DOLFIN's dofmap as it comes,
// _field_split_utils.py:39-50) need not be local at all.  At pcd_set_system the
// engine therefore measures the caller's numbering and, when it is not local,
// renumbers the velocity nodes by reverse Cuthill-McKee on the node graph of
// A00 (components of a node stay together, so F (x) I_d survives) and the
// pressure dofs by the velocity nodes they couple to; coarse multigrid levels
// inherit their order from the level above through the prolongation.  The
// permutation lives entirely inside the engine: it is folded into the index
// sets (hence into the entry gather / exit scatter that exist anyway) and
// applied to every operator, index list and field vector that crosses the ABI
// in the caller's field numbering.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace pcd {

// mean |row - col| / n over the stored entries: ~0.33 for a random numbering,
// <= a few per cent for a banded (geometric) one
inline double locality_metric(int64_t n, const int32_t* rowptr, const int32_t* col) {
  if (n <= 1 || rowptr[n] == 0) return 0.0;
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += std::abs((double)col[k] - (double)i);
  return s / (double)rowptr[n] / (double)n;
}

// reverse Cuthill-McKee of the symmetrised pattern; returns new -> old
inline std::vector<int32_t> rcm_order(int64_t n, const int32_t* rowptr, const int32_t* col) {
  // symmetrised adjacency (pattern + transpose), self loops dropped
  std::vector<int32_t> deg(n, 0);
  for (int64_t i = 0; i < n; ++i)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (col[k] != i && col[k] >= 0 && col[k] < n) { ++deg[i]; ++deg[col[k]]; }
  std::vector<int64_t> ap(n + 1, 0);
  for (int64_t i = 0; i < n; ++i) ap[i + 1] = ap[i] + deg[i];
  std::vector<int32_t> adj(ap[n]);
  {
    std::vector<int64_t> fill(ap.begin(), ap.end() - 1);
    for (int64_t i = 0; i < n; ++i)
      for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int32_t j = col[k];
        if (j == i || j < 0 || j >= n) continue;
        adj[fill[i]++] = j; adj[fill[j]++] = (int32_t)i;
      }
  }
  // (duplicates - an entry stored in both triangles - only repeat a visit test)
  std::vector<int32_t> order;
  order.reserve(n);
  std::vector<char> seen(n, 0);
  std::vector<int32_t> nbr;
  // start nodes: lowest degree first
  std::vector<int32_t> by_deg(n);
  std::iota(by_deg.begin(), by_deg.end(), 0);
  std::stable_sort(by_deg.begin(), by_deg.end(), [&](int32_t a, int32_t b) { return deg[a] < deg[b]; });
  auto bfs = [&](int32_t start, std::vector<int32_t>& out, bool mark) -> int32_t {
    // breadth-first from `start`, neighbours in order of increasing degree;
    // returns the last node reached (a far end of the component)
    const size_t first = out.size();
    out.push_back(start);
    seen[start] = 1;
    for (size_t q = first; q < out.size(); ++q) {
      const int32_t v = out[q];
      nbr.clear();
      for (int64_t k = ap[v]; k < ap[v + 1]; ++k)
        if (!seen[adj[k]]) { seen[adj[k]] = 1; nbr.push_back(adj[k]); }
      std::sort(nbr.begin(), nbr.end(), [&](int32_t a, int32_t b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
      out.insert(out.end(), nbr.begin(), nbr.end());
    }
    const int32_t last = out.back();
    if (!mark) { for (size_t q = first; q < out.size(); ++q) seen[out[q]] = 0; out.resize(first); }
    return last;
  };
  for (int32_t s : by_deg) {
    if (seen[s]) continue;
    // pseudo-peripheral start: the far end of a sweep from the low-degree node
    std::vector<int32_t> tmp;
    const int32_t far = bfs(s, tmp, false);
    (void)bfs(far, order, true);
  }
  std::reverse(order.begin(), order.end());
  return order;
}

// CLUSTER order for the vector-tile kernels: consecutive rows should touch as
// few DISTINCT columns as possible (a block's tile holds its distinct columns
// once).  A lexicographic numbering puts a pencil of nodes into a row block -
// in space a pencil of 37 P2 nodes touches ~512 nodes; a graph BALL of as many
// nodes touches far fewer.  Greedy: seeds in the caller's order (which keeps
// the global, slab-wise locality the caches rely on), every seed collects up
// to `k` unnumbered nodes breadth-first.  Returns new -> old.
inline std::vector<int32_t> cluster_order(int64_t n, const int32_t* rowptr, const int32_t* col, int k) {
  std::vector<int32_t> deg(n, 0);
  for (int64_t i = 0; i < n; ++i)
    for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q)
      if (col[q] != i && col[q] >= 0 && col[q] < n) { ++deg[i]; ++deg[col[q]]; }
  std::vector<int64_t> ap(n + 1, 0);
  for (int64_t i = 0; i < n; ++i) ap[i + 1] = ap[i] + deg[i];
  std::vector<int32_t> adj(ap[n]);
  {
    std::vector<int64_t> fill(ap.begin(), ap.end() - 1);
    for (int64_t i = 0; i < n; ++i)
      for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
        const int32_t j = col[q];
        if (j == i || j < 0 || j >= n) continue;
        adj[fill[i]++] = j; adj[fill[j]++] = (int32_t)i;
      }
  }
  std::vector<int32_t> order;
  order.reserve(n);
  std::vector<char> seen(n, 0);
  for (int64_t s = 0; s < n; ++s) {
    if (seen[s]) continue;
    const size_t first = order.size();
    order.push_back((int32_t)s);
    seen[s] = 1;
    for (size_t q = first; q < order.size() && order.size() - first < (size_t)k; ++q) {
      const int32_t v = order[q];
      for (int64_t e = ap[v]; e < ap[v + 1] && order.size() - first < (size_t)k; ++e)
        if (!seen[adj[e]]) { seen[adj[e]] = 1; order.push_back(adj[e]); }
    }
    // (within a cluster: ascending old numbers - neighbouring clusters then
    // share runs of consecutive columns)
    std::sort(order.begin() + first, order.end());
  }
  return order;
}

// order of the coarse dofs induced by an order of the fine ones: a coarse dof
// goes where the first fine dof it interpolates to goes (P: fine x coarse)
inline std::vector<int32_t> induced_order(int64_t nfine, int64_t ncoarse, const int32_t* prowptr,
                                          const int32_t* pcol, const int32_t* fine_old2new) {
  std::vector<int64_t> key(ncoarse, INT64_MAX);
  for (int64_t i = 0; i < nfine; ++i) {
    const int64_t pos = fine_old2new ? fine_old2new[i] : i;
    for (int32_t k = prowptr[i]; k < prowptr[i + 1]; ++k) key[pcol[k]] = std::min(key[pcol[k]], pos);
  }
  std::vector<int32_t> order(ncoarse);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
  return order;
}

inline std::vector<int32_t> invert_perm(const std::vector<int32_t>& new2old) {
  std::vector<int32_t> inv(new2old.size());
  for (size_t i = 0; i < new2old.size(); ++i) inv[new2old[i]] = (int32_t)i;
  return inv;
}

// node order -> dof order for `nc` interleaved components per node
inline std::vector<int32_t> expand_nodes(const std::vector<int32_t>& node_new2old, int nc) {
  std::vector<int32_t> d(node_new2old.size() * nc);
  for (size_t i = 0; i < node_new2old.size(); ++i)
    for (int c = 0; c < nc; ++c) d[i * nc + c] = node_new2old[i] * nc + c;
  return d;
}

// B = A(rows new2old_r, cols renumbered by old2new_c) with sorted columns;
// src[k] = position of entry k of B in A's arrays.  Null maps are identities.
struct PermCsr {
  std::vector<int32_t> rp, ci;
  std::vector<int64_t> src;
};
inline void permute_csr(int64_t nr, const int32_t* rowptr, const int32_t* col,
                        const int32_t* new2old_r, const int32_t* old2new_c, PermCsr& out) {
  out.rp.assign(nr + 1, 0);
  for (int64_t i = 0; i < nr; ++i) {
    const int64_t o = new2old_r ? new2old_r[i] : i;
    out.rp[i + 1] = out.rp[i] + (rowptr[o + 1] - rowptr[o]);
  }
  out.ci.resize(out.rp[nr]);
  out.src.resize(out.rp[nr]);
  std::vector<std::pair<int32_t, int64_t>> tmp;
  for (int64_t i = 0; i < nr; ++i) {
    const int64_t o = new2old_r ? new2old_r[i] : i;
    tmp.clear();
    for (int32_t k = rowptr[o]; k < rowptr[o + 1]; ++k)
      tmp.emplace_back(old2new_c ? old2new_c[col[k]] : col[k], (int64_t)k);
    std::sort(tmp.begin(), tmp.end());
    int64_t q = out.rp[i];
    for (auto& t : tmp) { out.ci[q] = t.first; out.src[q] = t.second; ++q; }
  }
}

}  // namespace pcd
