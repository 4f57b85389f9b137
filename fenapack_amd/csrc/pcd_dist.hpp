// pcd_dist.hpp - row partition, matrix localisation and communication
// backends of the multi-GPU path.
//
// One process drives one GPU.  DOF rows are split into contiguous blocks of a
// locality-preserving ordering (geometric strips), each rank keeps the rows it
// owns of every operator; a column it does not own becomes a *ghost* column
// whose value arrives in a halo exchange before the SpMV (the MPIAIJ picture
// PETSc gives the reference, restated for RCCL over xGMI).  Every rank is
// handed the same global matrices at set-up (or its own rows only:
// pcd_set_csr_local) and looks at the rows it owns; what the other ranks need
// from it arrives in a set-up handshake (localize_owned).  Set-up and the hot
// path need the same two primitives:
//   exchange()  - neighbour halo: grouped ncclSend/ncclRecv
//   allreduce() - dot products / norms: ncclAllReduce(sum, fp64) in place
// A second backend with the same two primitives runs R "ranks" as R threads of
// one process on one GPU (device-to-device copies + a barrier); it exists so
// that the partition / localisation / halo logic can be tested on a single-GPU
// box, where RCCL refuses two ranks on one device.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace pcd {

// ---- index spaces --------------------------------------------------------
// A space is one field or the [u; p] system.  Field f occupies global indices
// [goff[f], goff[f] + n[f]) and is cut at bounds[f][0..R]; the local layout of
// a rank is its slice of field 0 followed by its slice of field 1.
struct Space {
  int nf = 0;
  int64_t goff[2] = {0, 0};
  std::vector<int64_t> bounds[2];

  // cuts fall on multiples of `block` (velocity: the components of a node
  // stay on one rank)
  static std::vector<int64_t> cut(int64_t n, int R, int block) {
    std::vector<int64_t> b(R + 1);
    for (int r = 0; r <= R; ++r) {
      int64_t v = n * r / R;
      if (block > 1) v -= v % block;
      b[r] = v;
    }
    b[R] = n;
    return b;
  }
  static Space field(int64_t n, int R, int block) {
    Space s; s.nf = 1; s.bounds[0] = cut(n, R, block); return s;
  }
  static Space system(const Space& u, const Space& p) {
    Space s; s.nf = 2; s.bounds[0] = u.bounds[0]; s.bounds[1] = p.bounds[0];
    s.goff[1] = u.bounds[0].back();
    return s;
  }
  int64_t total() const {
    int64_t t = 0;
    for (int f = 0; f < nf; ++f) t += bounds[f].back();
    return t;
  }
  int64_t nloc(int r) const {
    int64_t t = 0;
    for (int f = 0; f < nf; ++f) t += bounds[f][r + 1] - bounds[f][r];
    return t;
  }
  int field_of(int64_t g) const { return (nf == 2 && g >= goff[1]) ? 1 : 0; }
  int owner(int64_t g) const {
    const int f = field_of(g);
    const int64_t i = g - goff[f];
    const auto& b = bounds[f];
    return (int)(std::upper_bound(b.begin(), b.end(), i) - b.begin()) - 1;
  }
  // local index of an owned global index
  int64_t local(int64_t g, int r) const {
    const int f = field_of(g);
    int64_t off = 0;
    for (int k = 0; k < f; ++k) off += bounds[k][r + 1] - bounds[k][r];
    return off + (g - goff[f]) - bounds[f][r];
  }
  // global index of local index i of rank r
  int64_t global(int64_t i, int r) const {
    for (int f = 0; f < nf; ++f) {
      const int64_t len = bounds[f][r + 1] - bounds[f][r];
      if (i < len) return goff[f] + bounds[f][r] + i;
      i -= len;
    }
    return -1;
  }
};

// ---- halo plan -------------------------------------------------------------
struct HaloPlan {
  int nghost = 0;
  std::vector<int> peers_recv, recv_off;    // recv_off has peers+1 entries
  std::vector<int> peers_send, send_off;
  std::vector<int32_t> send_idx;            // local indices to pack, per peer
};

// Cut rank `me`'s rows out of a global CSR (rows in `rs`, columns in `cs`),
// renumber columns (owned -> local, others -> nloc + ghost slot) and derive the
// halo plan.  `src_in` (optional) carries value provenance through.
inline void localize(const Space& rs, const Space& cs, int me, int R,
                     const int32_t* rowptr, const int32_t* col,
                     const double* val, const int64_t* src_in,
                     std::vector<int32_t>& orp, std::vector<int32_t>& oc,
                     std::vector<double>& ov, std::vector<int64_t>& osrc,
                     HaloPlan& plan) {
  const int64_t nrow_loc = rs.nloc(me), ncol_loc = cs.nloc(me);
  // needed[q] = sorted unique global columns rank q needs from another owner
  std::vector<std::vector<int64_t>> need_from_me(R);   // by requesting rank
  std::vector<int64_t> my_ghosts;
  const int64_t nrows = rs.total();
  std::vector<int> row_owner(nrows);
  for (int64_t g = 0; g < nrows; ++g) row_owner[g] = rs.owner(g);
  for (int64_t g = 0; g < nrows; ++g) {
    const int q = row_owner[g];
    for (int32_t k = rowptr[g]; k < rowptr[g + 1]; ++k) {
      const int o = cs.owner(col[k]);
      if (o == q) continue;
      if (q == me) my_ghosts.push_back(col[k]);
      else if (o == me) need_from_me[q].push_back(col[k]);
    }
  }
  auto uniq = [](std::vector<int64_t>& v) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
  };
  uniq(my_ghosts);
  // ghosts ordered by (owner, global)
  std::stable_sort(my_ghosts.begin(), my_ghosts.end(),
                   [&](int64_t a, int64_t b) { return cs.owner(a) < cs.owner(b); });
  plan = HaloPlan();
  plan.nghost = (int)my_ghosts.size();
  plan.recv_off.push_back(0);
  for (size_t i = 0; i < my_ghosts.size();) {
    const int o = cs.owner(my_ghosts[i]);
    size_t j = i;
    while (j < my_ghosts.size() && cs.owner(my_ghosts[j]) == o) ++j;
    plan.peers_recv.push_back(o);
    plan.recv_off.push_back((int)j);
    i = j;
  }
  plan.send_off.push_back(0);
  for (int q = 0; q < R; ++q) {
    if (q == me || need_from_me[q].empty()) continue;
    uniq(need_from_me[q]);
    plan.peers_send.push_back(q);
    for (int64_t g : need_from_me[q]) plan.send_idx.push_back((int32_t)cs.local(g, me));
    plan.send_off.push_back((int)plan.send_idx.size());
  }
  // local matrix
  orp.assign(nrow_loc + 1, 0);
  oc.clear(); ov.clear(); osrc.clear();
  std::vector<std::pair<int32_t, int64_t>> tmp;
  for (int64_t i = 0; i < nrow_loc; ++i) {
    const int64_t g = rs.global(i, me);
    tmp.clear();
    for (int32_t k = rowptr[g]; k < rowptr[g + 1]; ++k) {
      const int64_t cg = col[k];
      int32_t lc;
      if (cs.owner(cg) == me) lc = (int32_t)cs.local(cg, me);
      else {
        // position among my ghosts (sorted by owner then global)
        const int o = cs.owner(cg);
        size_t pi = 0;
        while (plan.peers_recv[pi] != o) ++pi;
        auto b = my_ghosts.begin() + plan.recv_off[pi];
        auto e = my_ghosts.begin() + plan.recv_off[pi + 1];
        lc = (int32_t)(ncol_loc + (std::lower_bound(b, e, cg) - my_ghosts.begin()));
      }
      tmp.emplace_back(lc, (int64_t)k);
    }
    std::sort(tmp.begin(), tmp.end());
    for (auto& t : tmp) {
      oc.push_back(t.first);
      if (val) ov.push_back(val[t.second]);
      osrc.push_back(src_in ? src_in[t.second] : t.second);
    }
    orp[i + 1] = (int32_t)oc.size();
  }
}

// ---- communication backends -----------------------------------------------
struct Msg { int peer; double* ptr; size_t count; };

struct CommBackend {
  int rank = 0, nranks = 1;
  std::string err;
  virtual ~CommBackend() {}
  // one-shot peer-write halos / reductions available (pcd_peer.hpp)
  virtual bool peer() const { return false; }
  // in-place sum over ranks of a device buffer, enqueued on `s`
  virtual int allreduce(double* dbuf, size_t count, hipStream_t s) = 0;
  // all sends and receives of one halo exchange, enqueued on `s`
  virtual int exchange(const std::vector<Msg>& sends,
                       const std::vector<Msg>& recvs, hipStream_t s) = 0;
};

// Rank-local localisation: the rows this rank OWNS are all it looks at (the
// owned-rows-only construction of SubfieldBC.h:136-155).  `span(i)` gives the
// entry range of local row i in `col` / `val` (GLOBAL column ids) - a view of
// a global CSR or arrays that hold this rank's rows only.  What the other
// ranks need from this one cannot be read off its own rows (rectangular and
// non-symmetric operators): it arrives in a set-up handshake over the
// communicator - one all-reduce of the R x R request counts, one neighbour
// exchange of the requested column ids (carried as doubles: exact below
// 2^53).  Collective: every rank calls it for every operator in the same
// order.  The ghost numbering (by owner, then global id) and the send lists
// (ascending global id per peer) equal those of `localize`.
template <class Span>
inline int localize_owned(const Space& rs, const Space& cs, int me, int R,
                          int64_t nrow_loc, Span span, const int32_t* col,
                          const double* val, const int64_t* src_in,
                          CommBackend* comm, hipStream_t stream,
                          std::vector<int32_t>& orp, std::vector<int32_t>& oc,
                          std::vector<double>& ov, std::vector<int64_t>& osrc,
                          HaloPlan& plan, std::string& err) {
  const int64_t ncol_loc = cs.nloc(me);
  std::vector<int64_t> my_ghosts;
  for (int64_t i = 0; i < nrow_loc; ++i) {
    const auto be = span(i);
    for (int64_t k = be.first; k < be.second; ++k)
      if (cs.owner(col[k]) != me) my_ghosts.push_back(col[k]);
  }
  std::sort(my_ghosts.begin(), my_ghosts.end());
  my_ghosts.erase(std::unique(my_ghosts.begin(), my_ghosts.end()), my_ghosts.end());
  std::stable_sort(my_ghosts.begin(), my_ghosts.end(),
                   [&](int64_t a, int64_t b) { return cs.owner(a) < cs.owner(b); });
  plan = HaloPlan();
  plan.nghost = (int)my_ghosts.size();
  plan.recv_off.push_back(0);
  for (size_t i = 0; i < my_ghosts.size();) {
    const int o = cs.owner(my_ghosts[i]);
    size_t j = i;
    while (j < my_ghosts.size() && cs.owner(my_ghosts[j]) == o) ++j;
    plan.peers_recv.push_back(o);
    plan.recv_off.push_back((int)j);
    i = j;
  }
  // ---- handshake: who needs what from me
  {
    auto hip_fail = [&](hipError_t e, const char* what) {
      if (e == hipSuccess) return false;
      err = std::string(what) + ": " + hipGetErrorString(e);
      return true;
    };
    std::vector<double> cnt((size_t)R * R, 0.0);
    for (size_t p = 0; p < plan.peers_recv.size(); ++p)
      cnt[(size_t)me * R + plan.peers_recv[p]] = (double)(plan.recv_off[p + 1] - plan.recv_off[p]);
    double* dcnt = nullptr;
    if (hip_fail(hipMalloc((void**)&dcnt, cnt.size() * sizeof(double)), "hipMalloc")) return 1;
    bool bad = hip_fail(hipMemcpyAsync(dcnt, cnt.data(), cnt.size() * sizeof(double), hipMemcpyHostToDevice, stream), "memcpy");
    if (!bad && comm->allreduce(dcnt, cnt.size(), stream)) { err = comm->err; bad = true; }
    if (!bad) bad = hip_fail(hipMemcpyAsync(cnt.data(), dcnt, cnt.size() * sizeof(double), hipMemcpyDeviceToHost, stream), "memcpy");
    if (!bad) bad = hip_fail(hipStreamSynchronize(stream), "sync");
    (void)hipFree(dcnt);
    if (bad) return 1;
    // requests: my ghost ids go to their owners, theirs come to me
    size_t nsend = my_ghosts.size(), nrecv = 0;
    std::vector<int> req_peers; std::vector<size_t> req_off(1, 0);
    for (int q = 0; q < R; ++q) {
      const size_t c = (size_t)cnt[(size_t)q * R + me];
      if (q == me || !c) continue;
      req_peers.push_back(q); nrecv += c; req_off.push_back(nrecv);
    }
    double *dsend = nullptr, *drecv = nullptr;
    if (hip_fail(hipMalloc((void**)&dsend, std::max<size_t>(nsend, 1) * sizeof(double)), "hipMalloc")) return 1;
    if (hip_fail(hipMalloc((void**)&drecv, std::max<size_t>(nrecv, 1) * sizeof(double)), "hipMalloc")) { (void)hipFree(dsend); return 1; }
    std::vector<double> ids(my_ghosts.begin(), my_ghosts.end()), got(nrecv);
    if (nsend) bad = hip_fail(hipMemcpyAsync(dsend, ids.data(), nsend * sizeof(double), hipMemcpyHostToDevice, stream), "memcpy");
    std::vector<Msg> sends, recvs;
    for (size_t p = 0; p < plan.peers_recv.size(); ++p)
      sends.push_back(Msg{plan.peers_recv[p], dsend + plan.recv_off[p], (size_t)(plan.recv_off[p + 1] - plan.recv_off[p])});
    for (size_t p = 0; p < req_peers.size(); ++p)
      recvs.push_back(Msg{req_peers[p], drecv + req_off[p], req_off[p + 1] - req_off[p]});
    if (!bad && comm->exchange(sends, recvs, stream)) { err = comm->err; bad = true; }
    if (!bad && nrecv) bad = hip_fail(hipMemcpyAsync(got.data(), drecv, nrecv * sizeof(double), hipMemcpyDeviceToHost, stream), "memcpy");
    if (!bad) bad = hip_fail(hipStreamSynchronize(stream), "sync");
    (void)hipFree(dsend); (void)hipFree(drecv);
    if (bad) return 1;
    plan.send_off.push_back(0);
    for (size_t p = 0; p < req_peers.size(); ++p) {
      plan.peers_send.push_back(req_peers[p]);
      for (size_t k = req_off[p]; k < req_off[p + 1]; ++k) {
        const int64_t g = (int64_t)got[k];
        if (cs.owner(g) != me) { err = "set-up handshake: asked for a column this rank does not own"; return 1; }
        plan.send_idx.push_back((int32_t)cs.local(g, me));
      }
      plan.send_off.push_back((int)plan.send_idx.size());
    }
  }
  // ---- local matrix (columns: owned -> local, ghosts -> ncol_loc + slot)
  orp.assign(nrow_loc + 1, 0);
  oc.clear(); ov.clear(); osrc.clear();
  std::vector<std::pair<int32_t, int64_t>> tmp;
  for (int64_t i = 0; i < nrow_loc; ++i) {
    const auto be = span(i);
    tmp.clear();
    for (int64_t k = be.first; k < be.second; ++k) {
      const int64_t cg = col[k];
      int32_t lc;
      const int o = cs.owner(cg);
      if (o == me) lc = (int32_t)cs.local(cg, me);
      else {
        size_t pi = 0;
        while (plan.peers_recv[pi] != o) ++pi;
        auto b = my_ghosts.begin() + plan.recv_off[pi];
        auto e = my_ghosts.begin() + plan.recv_off[pi + 1];
        lc = (int32_t)(ncol_loc + (std::lower_bound(b, e, cg) - my_ghosts.begin()));
      }
      tmp.emplace_back(lc, k);
    }
    std::sort(tmp.begin(), tmp.end());
    for (auto& t : tmp) {
      oc.push_back(t.first);
      if (val) ov.push_back(val[t.second]);
      osrc.push_back(src_in ? src_in[t.second] : t.second);
    }
    orp[i + 1] = (int32_t)oc.size();
  }
  return 0;
}

// RCCL, bound at run time so that the library loads without it
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t,
                            ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t,
                       hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t,
                       hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;

  bool load(std::string& err) {
    if (lib) return true;
    // a copy that is already mapped wins (PyTorch ships its own librccl.so
    // with the SONAME librccl.so.1: two copies in one process tear their
    // static state down twice at exit), else the ROCm one
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
      if (lib) break;
    }
    if (!lib)
      for (const char* name : {"librccl.so.1", "librccl.so",
                               "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
      }
    if (!lib) { err = std::string("dlopen librccl: ") + dlerror(); return false; }
#define PCD_SYM(field, name)                                            \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, name));        \
    if (!field) { err = std::string("dlsym ") + name; return false; }
    PCD_SYM(GetUniqueId, "ncclGetUniqueId")
    PCD_SYM(CommInitRank, "ncclCommInitRank")
    PCD_SYM(CommDestroy, "ncclCommDestroy")
    PCD_SYM(AllReduce, "ncclAllReduce")
    PCD_SYM(Send, "ncclSend")
    PCD_SYM(Recv, "ncclRecv")
    PCD_SYM(GroupStart, "ncclGroupStart")
    PCD_SYM(GroupEnd, "ncclGroupEnd")
    PCD_SYM(GetErrorString, "ncclGetErrorString")
#undef PCD_SYM
    return true;
  }
};

inline RcclApi& rccl_api() { static RcclApi api; return api; }

struct RcclBackend : CommBackend {
  ncclComm_t comm = nullptr;
  ~RcclBackend() override { if (comm) (void)rccl_api().CommDestroy(comm); }
  int check(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return 0;
    err = std::string(what) + ": " + rccl_api().GetErrorString(r);
    return 1;
  }
  int allreduce(double* dbuf, size_t count, hipStream_t s) override {
    return check(rccl_api().AllReduce(dbuf, dbuf, count, ncclDouble, ncclSum, comm, s),
                 "ncclAllReduce");
  }
  int exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs,
               hipStream_t s) override {
    if (sends.empty() && recvs.empty()) return 0;
    RcclApi& a = rccl_api();
    if (check(a.GroupStart(), "ncclGroupStart")) return 1;
    for (const Msg& m : recvs)
      if (check(a.Recv(m.ptr, m.count, ncclDouble, m.peer, comm, s), "ncclRecv")) return 1;
    for (const Msg& m : sends)
      if (check(a.Send(m.ptr, m.count, ncclDouble, m.peer, comm, s), "ncclSend")) return 1;
    return check(a.GroupEnd(), "ncclGroupEnd");
  }
};

// Host transport supplied by the caller (pcd_comm_init_host): two callbacks on
// HOST buffers - what an application that already owns an MPI communicator
// (the reference: PETSc's, through mpi4py) or a torch.distributed process
// group plugs in.  Device buffers are staged through the host, so this
// backend alone is a slow path; its role is the BOOTSTRAP of the peer
// protocol (pcd_peer.hpp), which then carries the hot path between the
// processes of a node - including two processes that share ONE GPU, where
// RCCL refuses to build a communicator.
typedef int (*host_allreduce_fn)(void* ctx, double* buf, int64_t count);
typedef int (*host_exchange_fn)(void* ctx, int nsend, const int* send_peers,
                                double* const* send_bufs, const int64_t* send_counts,
                                int nrecv, const int* recv_peers, double* const* recv_bufs,
                                const int64_t* recv_counts);
struct HostBackend : CommBackend {
  host_allreduce_fn ar = nullptr;
  host_exchange_fn ex = nullptr;
  void* ctx = nullptr;
  int fail(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    err = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
  }
  int allreduce(double* dbuf, size_t count, hipStream_t s) override {
    std::vector<double> hb(count);
    if (fail(hipMemcpyAsync(hb.data(), dbuf, count * sizeof(double), hipMemcpyDeviceToHost, s), "memcpy") ||
        fail(hipStreamSynchronize(s), "sync")) return 1;
    if (ar(ctx, hb.data(), (int64_t)count)) { err = "host all-reduce callback failed"; return 1; }
    if (fail(hipMemcpyAsync(dbuf, hb.data(), count * sizeof(double), hipMemcpyHostToDevice, s), "memcpy") ||
        fail(hipStreamSynchronize(s), "sync")) return 1;
    return 0;
  }
  int exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs,
               hipStream_t s) override {
    std::vector<std::vector<double>> sb(sends.size()), rb(recvs.size());
    std::vector<int> sp, rp;
    std::vector<double*> sptr, rptr;
    std::vector<int64_t> sc, rc;
    for (size_t i = 0; i < sends.size(); ++i) {
      sb[i].resize(sends[i].count);
      if (sends[i].count &&
          fail(hipMemcpyAsync(sb[i].data(), sends[i].ptr, sends[i].count * sizeof(double),
                              hipMemcpyDeviceToHost, s), "memcpy")) return 1;
      sp.push_back(sends[i].peer); sptr.push_back(sb[i].data()); sc.push_back((int64_t)sends[i].count);
    }
    for (size_t i = 0; i < recvs.size(); ++i) {
      rb[i].resize(recvs[i].count);
      rp.push_back(recvs[i].peer); rptr.push_back(rb[i].data()); rc.push_back((int64_t)recvs[i].count);
    }
    if (fail(hipStreamSynchronize(s), "sync")) return 1;
    if (ex(ctx, (int)sends.size(), sp.data(), sptr.data(), sc.data(), (int)recvs.size(), rp.data(),
           rptr.data(), rc.data())) { err = "host exchange callback failed"; return 1; }
    for (size_t i = 0; i < recvs.size(); ++i)
      if (recvs[i].count &&
          fail(hipMemcpyAsync(recvs[i].ptr, rb[i].data(), recvs[i].count * sizeof(double),
                              hipMemcpyHostToDevice, s), "memcpy")) return 1;
    return fail(hipStreamSynchronize(s), "sync");
  }
};

// In-process stand-in: R threads, one engine each, one GPU.  Used by the tests
// only (RCCL refuses two ranks on one device).  Nothing synchronises a stream:
// the ranks meet at host barriers to publish pointers, and the ordering of the
// device work across their streams is carried by events - a rank's copy of a
// peer's send buffer waits for the event the peer recorded after packing it,
// and the peer's next pack waits for the event recorded after that copy.
struct ThreadGroup {
  int nranks;
  std::atomic<int> arrived{0};
  std::atomic<uint64_t> phase{0};
  std::vector<double*> ar_buf;                       // allreduce operands
  std::vector<std::vector<Msg>> sends;               // posted sends per rank
  std::vector<hipEvent_t> ev_ready, ev_done;         // per rank
  std::vector<hipStream_t> stream;                   // the stream each rank enqueues on
  std::vector<char*> arenas;                         // peer protocol (pcd_peer.hpp): every rank's arena
  explicit ThreadGroup(int n)
      : nranks(n), ar_buf(n, nullptr), sends(n), ev_ready(n, nullptr), ev_done(n, nullptr),
        stream(n, nullptr), arenas(n, nullptr) {}
  // all ranks on ONE in-order stream (the tests' default: the null stream):
  // enqueue order is execution order, the host barriers alone order the work
  bool one_stream() const {
    for (int r = 1; r < nranks; ++r) if (stream[r] != stream[0]) return false;
    return true;
  }
  // sense-reversing spin barrier: the ranks meet hundreds of thousands of
  // times in a test run, a condition variable costs tens of microseconds each
  // A rank that failed (or left) never arrives: the others give up after
  // PCD_THREAD_BARRIER_TIMEOUT_S (default 120 s) and the group stays failed -
  // every later barrier returns at once, every rank leaves with an error
  // instead of spinning forever.
  std::atomic<bool> failed{false};
  bool barrier() {
    if (failed.load(std::memory_order_acquire)) return false;
    const uint64_t ph = phase.load(std::memory_order_acquire);
    if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == nranks) {
      arrived.store(0, std::memory_order_relaxed);
      phase.store(ph + 1, std::memory_order_release);
      return true;
    }
    static const double limit = [] {
      const char* e = getenv("PCD_THREAD_BARRIER_TIMEOUT_S");
      return e ? atof(e) : 120.0;
    }();
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (phase.load(std::memory_order_acquire) == ph) {
      if (failed.load(std::memory_order_acquire)) return false;
      if (++spins > 2000) {
        std::this_thread::yield();
        spins = 0;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
          failed.store(true, std::memory_order_release);
          return false;
        }
      }
    }
    return true;
  }
  void fail_group() { failed.store(true, std::memory_order_release); }
};

struct ThreadBackend : CommBackend {
  ThreadGroup* g = nullptr;
  double* tmp = nullptr;
  size_t tmp_n = 0;
  ~ThreadBackend() override { if (tmp) (void)hipFree(tmp); }
  // (a rank-local failure inside a collective fails the group at once: the
  // others would otherwise wait for the barrier's time-out)
  int fail(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    err = std::string(what) + ": " + hipGetErrorString(e);
    if (g) g->fail_group();
    return 1;
  }
  int gone() {
    err = "thread ranks: a rank failed or did not arrive at a collective";
    return 1;
  }
  int events() {
    if (!g->ev_ready[rank] &&
        fail(hipEventCreateWithFlags(&g->ev_ready[rank], hipEventDisableTiming), "event")) return 1;
    if (!g->ev_done[rank] &&
        fail(hipEventCreateWithFlags(&g->ev_done[rank], hipEventDisableTiming), "event")) return 1;
    return 0;
  }
  int allreduce(double* dbuf, size_t count, hipStream_t s) override {
    if (nranks > 16) { err = "thread backend: at most 16 ranks"; return 1; }
    if (events()) return 1;
    if (tmp_n < count) {
      if (tmp) (void)hipFree(tmp);
      tmp = nullptr; tmp_n = 0;
      if (fail(hipMalloc((void**)&tmp, count * sizeof(double)), "hipMalloc")) return 1;
      tmp_n = count;
    }
    g->ar_buf[rank] = dbuf;
    g->stream[rank] = s;
    if (!g->barrier()) return gone();                // operands + streams published
    const bool ordered = g->one_stream();            // (same answer on every rank)
    if (!ordered) {
      if (fail(hipEventRecord(g->ev_ready[rank], s), "record")) return 1;
      if (!g->barrier()) return gone();              // every event recorded
    }
    RankBufs bufs;
    bufs.n = nranks;
    for (int r = 0; r < nranks; ++r) {
      bufs.p[r] = g->ar_buf[r];
      if (!ordered && r != rank && fail(hipStreamWaitEvent(s, g->ev_ready[r], 0), "wait")) return 1;
    }
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((count + 255) / 256, 1024));
    hipLaunchKernelGGL(k_sum_ranks, dim3(grid), dim3(256), 0, s, bufs, (int64_t)count, tmp);
    if (!ordered && fail(hipEventRecord(g->ev_done[rank], s), "record")) return 1;
    if (!g->barrier()) return gone();                // every rank has (enqueued its) read of every operand
    if (!ordered)
      for (int r = 0; r < nranks; ++r)
        if (r != rank && fail(hipStreamWaitEvent(s, g->ev_done[r], 0), "wait")) return 1;
    return fail(hipMemcpyAsync(dbuf, tmp, count * sizeof(double), hipMemcpyDeviceToDevice, s), "memcpy");
  }
  int exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs,
               hipStream_t s) override {
    if (events()) return 1;
    g->sends[rank] = sends;
    g->stream[rank] = s;
    if (!g->barrier()) return gone();                // send buffers + streams published
    const bool ordered = g->one_stream();
    if (!ordered) {
      if (fail(hipEventRecord(g->ev_ready[rank], s), "record")) return 1;
      if (!g->barrier()) return gone();              // every event recorded
    }
    // several messages between one pair of ranks (grouped halos of several
    // operators) match in posting order, as grouped ncclSend / ncclRecv do
    std::vector<int> taken(nranks, 0);
    int bad = 0;
    for (const Msg& m : recvs) {
      const Msg* src = nullptr;
      int seen = 0;
      for (const Msg& q : g->sends[m.peer])
        if (q.peer == rank && seen++ == taken[m.peer]) { src = &q; break; }
      ++taken[m.peer];
      if (!src || src->count != m.count) {
        char buf[200];
        snprintf(buf, sizeof buf, "halo mismatch: rank %d expects %zu doubles from rank %d as its message "
                 "%d, found %s%zu (rank %d posted %zu sends, this rank %zu receives)", rank, m.count,
                 m.peer, taken[m.peer] - 1, src ? "" : "none; ", src ? src->count : (size_t)0, m.peer,
                 g->sends[m.peer].size(), recvs.size());
        err = buf; bad = 1; break;
      }
      if ((!ordered && fail(hipStreamWaitEvent(s, g->ev_ready[m.peer], 0), "wait")) ||
          fail(hipMemcpyAsync(m.ptr, src->ptr, m.count * sizeof(double),
                              hipMemcpyDeviceToDevice, s), "memcpy")) { bad = 1; break; }
    }
    if (!ordered && !bad && fail(hipEventRecord(g->ev_done[rank], s), "record")) bad = 1;
    if (!g->barrier()) return gone();                // all copies are enqueued
    if (bad) return 1;
    // my send buffers may be packed again only after my readers' copies
    if (!ordered)
      for (const Msg& m : sends)
        if (fail(hipStreamWaitEvent(s, g->ev_done[m.peer], 0), "wait")) return 1;
    return 0;
  }
};

}  // namespace pcd
