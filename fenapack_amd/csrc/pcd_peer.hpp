// pcd_peer.hpp - one-shot peer-write halos and reductions (the
// payloads of this path are latency-bound - 16 B dot products, halos of a few
// kB to ~1 MB - so "prefer one-shot peer-write reductions and minimise the
// NUMBER of collectives").
//
// The grouped ncclSend / ncclRecv halo costs a proxy hand-shake and its own
// kernel per exchange, cannot be captured into a hipGraph together with the
// engine's kernels, and is issued from the host.  Here every rank maps the
// other ranks' ARENA - one uncached device allocation per rank, shared through
// HIP IPC between the processes of a node (hipIpcGetMemHandle /
// hipIpcOpenMemHandle, as RCCL itself does for its intra-node transport), or
// addressed directly by the thread ranks of the single-GPU tests - and a halo
// exchange is ONE kernel of the engine's own stream:
//
//   pack   x[send_idx] -> the neighbour's landing buffer (remote stores)
//   signal release-store of the exchange's sequence number into the
//          neighbour's flag, after the last workgroup's data is out
//   wait   acquire-spin on this rank's flags until every neighbour's
//          sequence number arrived (bounded: a dead peer sets an error flag
//          instead of hanging the GPU)
//   land   landing buffer -> the operator's ghost segment (ordinary cached
//          memory at the address the SpMV kernels were launched with)
//
// Landing buffers are double-buffered by the parity of the sequence number,
// which lives in device memory and is advanced by the kernel itself: a
// captured hipGraph replays without host-side state, and a neighbour's
// exchange n+2 cannot overwrite data of exchange n that is still being read
// (it follows that neighbour's wait for this rank's exchange n+1, which this
// rank issues after its consumers of exchange n - stream order).  Every
// neighbour pair of this engine exchanges in both directions between two
// uses of a one-directional halo (the smoother steps between a restriction
// and the next one), which is what the argument needs.
//
// allreduce (<= kPeerMaxCount doubles: CG / GMRES dot products): every rank
// stores its operand into its slot of every rank's arena, signals, waits for
// all, and sums the slots in rank order - the same bits on every rank.
//
// Set-up (arena exchange, registration of a halo: where a neighbour's landing
// buffer and flag live) goes through the bootstrap backend (RCCL or the
// thread group); bulk transfers (whole-vector gathers of the host-pointer
// API) stay there too.
#pragma once
#include "pcd_dist.hpp"

#include <set>
#include <utility>

namespace pcd {

constexpr int kPeerMaxPeers = 16;
constexpr int kPeerMaxCount = 65536;        // doubles per peer all-reduce (a replicated coarse level)
constexpr int kPeerXchgGrid = 32;           // workgroups of one exchange kernel

// one halo channel, device view (passed by value)
struct PeerHaloDev {
  int nsend = 0, nsp = 0, nrecv = 0, nrp = 0;
  const int* send_idx = nullptr;            // local indices to pack (nsend)
  int send_off[kPeerMaxPeers + 1];          // per send peer
  int recv_off[kPeerMaxPeers + 1];          // per recv peer
  double* dst[kPeerMaxPeers];               // remote landing buffers (2 x count each)
  unsigned long long* rflag[kPeerMaxPeers]; // remote flags
  double* land[kPeerMaxPeers];              // local landing buffers (2 x count each)
  unsigned long long* lflag[kPeerMaxPeers]; // local flags
  double* ghost = nullptr;                  // the operator's ghost segment
  unsigned long long* seq = nullptr;        // sequence number (device)
  unsigned* ctr = nullptr;                  // [0] arrived, [1] finished workgroups
  int* err = nullptr;                       // set when a wait gave up
};

static __global__ __launch_bounds__(kBlock) void k_halo_xchg(PeerHaloDev d, const double* x,
                                                       long long spin_limit) {
  __shared__ int s_last;
  const unsigned long long seq_now = *d.seq + 1;      // (advanced by the last finisher)
  const size_t p = (size_t)(seq_now & 1);
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.nsend; i += gridDim.x * kBlock) {
    int q = 0;
    while (i >= d.send_off[q + 1]) ++q;
    const int cnt = d.send_off[q + 1] - d.send_off[q];
    d.dst[q][p * cnt + (i - d.send_off[q])] = x[d.send_idx[i]];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&d.ctr[0], 1u) == gridDim.x - 1;
  __syncthreads();
  if (s_last) {                                        // every workgroup's data is out
    __threadfence_system();
    if ((int)threadIdx.x < d.nsp)
      __hip_atomic_store(d.rflag[threadIdx.x], seq_now, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) d.ctr[0] = 0;
  }
  if ((int)threadIdx.x < d.nrp) {
    // bounded by the wall clock (s_memrealtime, 100 MHz); sticky: once a wait
    // gave up, later ones do not spin - one time-out per failure, not one per
    // exchange
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(d.lflag[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq_now) {
      if (*(volatile int*)d.err) break;
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > spin_limit) { *d.err = 1; break; }
    }
  }
  __syncthreads();
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < d.nrecv; e += gridDim.x * kBlock) {
    int q = 0;
    while (e >= d.recv_off[q + 1]) ++q;
    const int cnt = d.recv_off[q + 1] - d.recv_off[q];
    d.ghost[e] = __hip_atomic_load(&d.land[q][p * cnt + (e - d.recv_off[q])], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&d.ctr[1], 1u) == gridDim.x - 1) {
    d.ctr[1] = 0;
    *d.seq = seq_now;
  }
}

// The same exchange in two kernels, for consumers that compute the rows
// needing no ghost column in between (PCD_OVERLAP=1): k_halo_send packs,
// stores into the neighbours' landing buffers and signals; k_halo_wait waits
// for the neighbours' signals, lands the data in the ghost segment and
// advances the sequence number.  Both read *d.seq before anyone advances it
// (same stream, in order), so they agree on the exchange's number.
static __global__ __launch_bounds__(kBlock) void k_halo_send(PeerHaloDev d, const double* x) {
  __shared__ int s_last;
  const unsigned long long seq_now = *d.seq + 1;
  const size_t p = (size_t)(seq_now & 1);
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.nsend; i += gridDim.x * kBlock) {
    int q = 0;
    while (i >= d.send_off[q + 1]) ++q;
    const int cnt = d.send_off[q + 1] - d.send_off[q];
    d.dst[q][p * cnt + (i - d.send_off[q])] = x[d.send_idx[i]];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&d.ctr[0], 1u) == gridDim.x - 1;
  __syncthreads();
  if (s_last) {                                        // every workgroup's data is out
    __threadfence_system();
    if ((int)threadIdx.x < d.nsp)
      __hip_atomic_store(d.rflag[threadIdx.x], seq_now, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) d.ctr[0] = 0;
  }
}

static __global__ __launch_bounds__(kBlock) void k_halo_wait(PeerHaloDev d, long long spin_limit) {
  const unsigned long long seq_now = *d.seq + 1;
  const size_t p = (size_t)(seq_now & 1);
  if ((int)threadIdx.x < d.nrp) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(d.lflag[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq_now) {
      if (*(volatile int*)d.err) break;
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > spin_limit) { *d.err = 1; break; }
    }
  }
  __syncthreads();
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < d.nrecv; e += gridDim.x * kBlock) {
    int q = 0;
    while (e >= d.recv_off[q + 1]) ++q;
    const int cnt = d.recv_off[q + 1] - d.recv_off[q];
    d.ghost[e] = __hip_atomic_load(&d.land[q][p * cnt + (e - d.recv_off[q])], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&d.ctr[1], 1u) == gridDim.x - 1) {
    d.ctr[1] = 0;
    *d.seq = seq_now;
  }
}

// all-reduce channel, device view
struct PeerReduceDev {
  int nranks = 0, rank = 0;
  double* slot[kPeerMaxPeers];              // rank r's slot array [2][nranks][kPeerMaxCount]
  unsigned long long* flag[kPeerMaxPeers];  // rank r's flags [nranks]
  unsigned long long* seq = nullptr;
  unsigned* ctr = nullptr;                  // [0] arrived, [1] finished workgroups
  int* err = nullptr;
};

static __global__ __launch_bounds__(kBlock) void k_peer_allreduce(PeerReduceDev d, double* buf, int count,
                                                            long long spin_limit) {
  __shared__ int s_last;
  const unsigned long long seq_now = *d.seq + 1;       // (advanced by the last finisher)
  const size_t p = (size_t)(seq_now & 1);
  const size_t mine = (p * d.nranks + d.rank) * kPeerMaxCount;
  const long long total = (long long)count * d.nranks;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (long long)gridDim.x * kBlock) {
    const int r = (int)(i / count), k = (int)(i % count);
    d.slot[r][mine + k] = buf[k];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&d.ctr[0], 1u) == gridDim.x - 1;
  __syncthreads();
  if (s_last) {
    __threadfence_system();
    if ((int)threadIdx.x < d.nranks)
      __hip_atomic_store(&d.flag[threadIdx.x][d.rank], seq_now, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) d.ctr[0] = 0;
  }
  if ((int)threadIdx.x < d.nranks) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(&d.flag[d.rank][threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq_now) {
      if (*(volatile int*)d.err) break;
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > spin_limit) { *d.err = 1; break; }
    }
  }
  __syncthreads();
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < count; k += gridDim.x * kBlock) {
    double s = 0.0;
    for (int r = 0; r < d.nranks; ++r)
      s += __hip_atomic_load(&d.slot[d.rank][(p * d.nranks + r) * kPeerMaxCount + k], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
    buf[k] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&d.ctr[1], 1u) == gridDim.x - 1) {
    d.ctr[1] = 0;
    *d.seq = seq_now;
  }
}

// host side of one halo channel
struct PeerBackend;
struct PeerHalo {
  bool ready = false;
  PeerHaloDev dev;
  int grid = 1;
  // what the channel holds of its rank's arena (offset, bytes): handed back
  // when the operator is released (peer_give_back)
  PeerBackend* owner = nullptr;
  std::vector<std::pair<size_t, size_t>> takes;
};
// backends alive in this process: an operator may outlive the communicator it
// registered its halo with (pcd_comm_init on an engine that holds operators)
inline std::mutex& peer_live_mu() { static std::mutex m; return m; }
inline std::set<PeerBackend*>& peer_live() { static std::set<PeerBackend*> s; return s; }

inline void peer_give_back(PeerHalo& ph);  // (defined below PeerBackend)

struct PeerBackend : CommBackend {
  CommBackend* boot = nullptr;              // RCCL / thread group: set-up and bulk
  char* arena = nullptr;                    // this rank's arena (uncached)
  size_t cap = 0, used = 0;
  std::vector<char*> peer_arena;            // every rank's arena in this address space
  std::vector<void*> opened;                // IPC mappings to close
  PeerReduceDev red;
  unsigned long long* dseq = nullptr;       // counters of the all-reduce channel (ordinary memory)
  int* derr = nullptr;                      // error word read by the host at synchronisation points
  long long spin_limit = 3000000000ll;      // 30 s of the 100 MHz wall clock
  bool process_mode = false;
  int device = 0;                           // the device the arena lives on (init)
  long boot_calls = 0;                      // hot-path calls that went to the bootstrap backend
  long peer_calls = 0;                      // exchanges / reductions issued as kernels of the stream
  long declined = 0;                        // halo channels that did not fit an arena (bootstrap path instead)
  std::vector<std::pair<size_t, size_t>> freed;   // (offset, bytes) handed back by released operators
  std::vector<std::pair<size_t, size_t>> taking;  // ... taken by the registration in progress

  PeerBackend() {
    std::lock_guard<std::mutex> lk(peer_live_mu());
    peer_live().insert(this);
  }
  ~PeerBackend() override {
    {
      std::lock_guard<std::mutex> lk(peer_live_mu());
      peer_live().erase(this);
    }
    for (void* p : opened) (void)hipIpcCloseMemHandle(p);
    if (arena) (void)hipFree(arena);
    if (dseq) (void)hipFree(dseq);
    if (derr) (void)hipFree(derr);
    delete boot;
  }
  bool peer() const override { return true; }
  int fail(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    err = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
  }
  // sub-allocation (256-byte granules); 0 when the arena is full
  // A region handed back by a released operator is used again for a request
  // of the same size (a hierarchy pushed again after a pattern change asks
  // for the sizes it asked for before); its flags start from zero like fresh
  // arena memory - zeroed here, before the registration's collective steps
  // tell any peer where the region is.
  char* take(size_t bytes) {
    const size_t b = (bytes + 255) / 256 * 256;
    for (size_t i = 0; i < freed.size(); ++i)
      if (freed[i].second == b && b) {
        char* p = arena + freed[i].first;
        if (hipMemset(p, 0, b) != hipSuccess) return nullptr;
        taking.push_back(freed[i]);
        freed.erase(freed.begin() + (long)i);
        return p;
      }
    if (used + b > cap) return nullptr;
    char* p = arena + used;
    used += b;
    if (b) taking.push_back({(size_t)(p - arena), b});
    return p;
  }
  // (the caller has drained the OWNER's device first - peer_give_back below:
  // the regions may be handed out again at once, so nothing of this rank's
  // may still be in flight on them; its last exchange on the channel has then
  // completed, and with it every peer's store into these buffers)
  void give_back(PeerHalo& ph) {
    for (auto& t : ph.takes) freed.push_back(t);
    ph.takes.clear(); ph.owner = nullptr;
  }
  char* remote(int r, const char* mine_off_base, size_t off) const {
    (void)mine_off_base;
    return peer_arena[r] + off;
  }

  // collective; `arenas_of_group` (thread mode) is where the ranks of one
  // process publish their arena pointers, indexed by rank
  int init(size_t capacity, char** arenas_of_group, void (*group_barrier)(void*), void* group,
           hipStream_t s) {
    cap = capacity;
    (void)hipGetDevice(&device);
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, cap, hipDeviceMallocUncached);
    if (e != hipSuccess) e = hipExtMallocWithFlags(&p, cap, hipDeviceMallocFinegrained);
    if (fail(e, "arena (hipExtMallocWithFlags)")) return 1;
    arena = (char*)p;
    if (fail(hipMemset(arena, 0, cap), "memset")) return 1;
    if (fail(hipMalloc((void**)&dseq, 64), "hipMalloc")) return 1;
    if (fail(hipMalloc((void**)&derr, 64), "hipMalloc")) return 1;
    if (fail(hipMemset(dseq, 0, 64), "memset") || fail(hipMemset(derr, 0, 64), "memset")) return 1;
    if (fail(hipDeviceSynchronize(), "sync")) return 1;
    peer_arena.assign(nranks, nullptr);
    if (arenas_of_group) {
      arenas_of_group[rank] = arena;
      group_barrier(group);
      for (int r = 0; r < nranks; ++r) peer_arena[r] = arenas_of_group[r];
      group_barrier(group);
    } else {
      // one process per GPU: the IPC handles travel as doubles (one per byte)
      // through the bootstrap all-reduce
      process_mode = true;
      hipIpcMemHandle_t hnd;
      if (fail(hipIpcGetMemHandle(&hnd, arena), "hipIpcGetMemHandle")) return 1;
      const size_t hb = sizeof hnd, n = hb * nranks;
      std::vector<double> hv(n, 0.0);
      const unsigned char* raw = reinterpret_cast<const unsigned char*>(&hnd);
      for (size_t i = 0; i < hb; ++i) hv[(size_t)rank * hb + i] = (double)raw[i];
      double* dv = nullptr;
      if (fail(hipMalloc((void**)&dv, n * sizeof(double)), "hipMalloc")) return 1;
      bool bad = fail(hipMemcpyAsync(dv, hv.data(), n * sizeof(double), hipMemcpyHostToDevice, s), "memcpy");
      if (!bad && boot->allreduce(dv, n, s)) { err = boot->err; bad = true; }
      if (!bad) bad = fail(hipMemcpyAsync(hv.data(), dv, n * sizeof(double), hipMemcpyDeviceToHost, s), "memcpy");
      if (!bad) bad = fail(hipStreamSynchronize(s), "sync");
      (void)hipFree(dv);
      if (bad) return 1;
      for (int r = 0; r < nranks; ++r) {
        if (r == rank) { peer_arena[r] = arena; continue; }
        hipIpcMemHandle_t h2;
        unsigned char* w = reinterpret_cast<unsigned char*>(&h2);
        for (size_t i = 0; i < hb; ++i) w[i] = (unsigned char)hv[(size_t)r * hb + i];
        void* q = nullptr;
        if (fail(hipIpcOpenMemHandle(&q, h2, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle")) return 1;
        opened.push_back(q);
        peer_arena[r] = (char*)q;
      }
    }
    // the all-reduce channel sits at the same offsets in every arena
    char* slots = take((size_t)2 * nranks * kPeerMaxCount * sizeof(double));
    char* flags = take((size_t)nranks * sizeof(unsigned long long));
    if (!slots || !flags) { err = "peer arena too small"; return 1; }
    red.nranks = nranks; red.rank = rank;
    for (int r = 0; r < nranks; ++r) {
      red.slot[r] = reinterpret_cast<double*>(peer_arena[r] + (slots - arena));
      red.flag[r] = reinterpret_cast<unsigned long long*>(peer_arena[r] + (flags - arena));
    }
    red.seq = dseq;
    red.ctr = reinterpret_cast<unsigned*>(dseq + 2);
    red.err = derr;
    return 0;
  }

  int allreduce(double* dbuf, size_t count, hipStream_t s) override {
    if (count > (size_t)kPeerMaxCount || !usable(s)) {
      ++boot_calls;
      return forward(boot->allreduce(dbuf, count, s));
    }
    // (in place: a sum is written only after the flags of ALL ranks arrived,
    // this rank's own among them - and that one is set after its last
    // workgroup pushed its share of the operand)
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((count * nranks + kBlock - 1) / kBlock,
                                                               (size_t)kPeerXchgGrid));
    ++peer_calls;
    hipLaunchKernelGGL(k_peer_allreduce, dim3(grid), dim3(kBlock), 0, s, red, dbuf, (int)count, spin_limit);
    return fail(hipGetLastError(), "k_peer_allreduce");
  }
  int exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs, hipStream_t s) override {
    ++boot_calls;
    return forward(boot->exchange(sends, recvs, s));
  }
  int forward(int rc) { if (rc) err = boot->err; return rc; }

  // Collective registration of one halo: this rank allocates a landing buffer
  // and a flag per receive peer and tells each where they are; what it is
  // told by its send peers completes the channel.  False (channel not ready,
  // exchanges of this operator keep to the bootstrap path) when an arena is
  // full - decided by every rank alike through the all-reduce of a flag.
  int register_halo(const HaloPlan& pl, const int* d_send_idx, double* ghost, PeerHalo& ph,
                    hipStream_t s) {
    ph = PeerHalo();
    // The double-buffered landing areas are safe because a sender's exchange
    // n + 2 follows ITS wait for the receiver's exchange n + 1, which the
    // receiver issues after its consumers of n - true only if every receiver
    // of a halo also sends to that peer on the same channel.  One-directional
    // halos (restrictions, algebraic prolongators) get the missing direction
    // as an EMPTY message: a flag and no data.  The rule is its own mirror
    // image (q in recv \ send here <=> this rank in send \ recv there), so
    // the ranks agree without talking.
    HaloPlan sym = pl;
    for (int q : pl.peers_recv)
      if (std::find(pl.peers_send.begin(), pl.peers_send.end(), q) == pl.peers_send.end()) {
        sym.peers_send.push_back(q); sym.send_off.push_back(sym.send_off.back());
      }
    for (int q : pl.peers_send)
      if (std::find(pl.peers_recv.begin(), pl.peers_recv.end(), q) == pl.peers_recv.end()) {
        sym.peers_recv.push_back(q); sym.recv_off.push_back(sym.recv_off.back());
      }
    return register_halo_sym(sym, d_send_idx, ghost, ph, s);
  }
  int register_halo_sym(const HaloPlan& pl, const int* d_send_idx, double* ghost, PeerHalo& ph,
                        hipStream_t s) {
    const int nsp = (int)pl.peers_send.size(), nrp = (int)pl.peers_recv.size();
    bool fits = nsp <= kPeerMaxPeers && nrp <= kPeerMaxPeers;
    std::vector<size_t> land_off(nrp, 0), flag_off(nrp, 0);
    taking.clear();
    for (int j = 0; j < nrp && fits; ++j) {
      const size_t cnt = (size_t)(pl.recv_off[j + 1] - pl.recv_off[j]);
      char* l = take(2 * cnt * sizeof(double));
      char* f = take(sizeof(unsigned long long));
      if (!l || !f) { fits = false; break; }
      land_off[j] = (size_t)(l - arena); flag_off[j] = (size_t)(f - arena);
    }
    // counters of the channel (ordinary device memory)
    unsigned long long* seq = nullptr;
    if (fits && hipMalloc((void**)&seq, 64) != hipSuccess) fits = false;
    if (fits && hipMemset(seq, 0, 64) != hipSuccess) fits = false;
    // all ranks agree (and the offsets travel) through the bootstrap backend
    double ok = fits ? 0.0 : 1.0;
    double* dok = nullptr;
    if (fail(hipMalloc((void**)&dok, sizeof(double)), "hipMalloc")) return 1;
    bool bad = fail(hipMemcpyAsync(dok, &ok, sizeof ok, hipMemcpyHostToDevice, s), "memcpy");
    if (!bad && boot->allreduce(dok, 1, s)) { err = boot->err; bad = true; }
    if (!bad) bad = fail(hipMemcpyAsync(&ok, dok, sizeof ok, hipMemcpyDeviceToHost, s), "memcpy");
    if (!bad) bad = fail(hipStreamSynchronize(s), "sync");
    (void)hipFree(dok);
    if (bad) return 1;
    if (ok != 0.0) {                                   // nobody registers
      for (auto& t : taking) freed.push_back(t);
      taking.clear();
      if (seq) (void)hipFree(seq);
      ++declined;
      if (getenv("PCD_COMM_VERBOSE"))
        fprintf(stderr, "[pcd comm] rank %d: a halo channel does not fit the peer arena (%zu of %zu bytes "
                        "in use, PCD_PEER_ARENA_MB): its exchanges take the bootstrap path\n",
                rank, used, cap);
      return 0;
    }
    ph.owner = this;
    ph.takes.swap(taking);
    std::vector<double> out(2 * (size_t)std::max(nrp, 1)), in(2 * (size_t)std::max(nsp, 1));
    for (int j = 0; j < nrp; ++j) { out[2 * j] = (double)land_off[j]; out[2 * j + 1] = (double)flag_off[j]; }
    double *dout = nullptr, *din = nullptr;
    if (fail(hipMalloc((void**)&dout, out.size() * sizeof(double)), "hipMalloc")) return 1;
    if (fail(hipMalloc((void**)&din, in.size() * sizeof(double)), "hipMalloc")) { (void)hipFree(dout); return 1; }
    bad = fail(hipMemcpyAsync(dout, out.data(), out.size() * sizeof(double), hipMemcpyHostToDevice, s), "memcpy");
    std::vector<Msg> sends, recvs;
    for (int j = 0; j < nrp; ++j) sends.push_back(Msg{pl.peers_recv[j], dout + 2 * j, 2});
    for (int i = 0; i < nsp; ++i) recvs.push_back(Msg{pl.peers_send[i], din + 2 * i, 2});
    if (!bad && boot->exchange(sends, recvs, s)) { err = boot->err; bad = true; }
    if (!bad && nsp) bad = fail(hipMemcpyAsync(in.data(), din, 2 * (size_t)nsp * sizeof(double), hipMemcpyDeviceToHost, s), "memcpy");
    if (!bad) bad = fail(hipStreamSynchronize(s), "sync");
    (void)hipFree(dout); (void)hipFree(din);
    if (bad) return 1;
    PeerHaloDev& d = ph.dev;
    d.nsend = (int)pl.send_idx.size(); d.nsp = nsp; d.nrecv = pl.nghost; d.nrp = nrp;
    d.send_idx = d_send_idx;
    for (int i = 0; i <= nsp; ++i) d.send_off[i] = pl.send_off[i];
    for (int j = 0; j <= nrp; ++j) d.recv_off[j] = pl.recv_off[j];
    for (int i = 0; i < nsp; ++i) {
      char* base = peer_arena[pl.peers_send[i]];
      d.dst[i] = reinterpret_cast<double*>(base + (size_t)in[2 * i]);
      d.rflag[i] = reinterpret_cast<unsigned long long*>(base + (size_t)in[2 * i + 1]);
    }
    for (int j = 0; j < nrp; ++j) {
      d.land[j] = reinterpret_cast<double*>(arena + land_off[j]);
      d.lflag[j] = reinterpret_cast<unsigned long long*>(arena + flag_off[j]);
    }
    d.ghost = ghost;
    d.seq = seq;
    d.ctr = reinterpret_cast<unsigned*>(seq + 2);
    d.err = derr;
    const int work = std::max(d.nsend, d.nrecv);
    ph.grid = std::max(1, std::min(kPeerXchgGrid, (work + kBlock - 1) / kBlock));
    ph.ready = true;
    return 0;
  }

  // thread ranks on the legacy stream share ONE in-order queue: a kernel that
  // waits for another rank's kernel would sit in front of it.  They keep to
  // the bootstrap protocol unless every rank was given a stream of its own.
  bool usable(hipStream_t s) const { return process_mode || s != nullptr; }
  int halo(const PeerHalo& ph, const double* x, hipStream_t s) {
    ++peer_calls;
    hipLaunchKernelGGL(k_halo_xchg, dim3(ph.grid), dim3(kBlock), 0, s, ph.dev, x, spin_limit);
    return fail(hipGetLastError(), "k_halo_xchg");
  }
  int halo_send(const PeerHalo& ph, const double* x, hipStream_t s) {
    ++peer_calls;
    hipLaunchKernelGGL(k_halo_send, dim3(ph.grid), dim3(kBlock), 0, s, ph.dev, x);
    return fail(hipGetLastError(), "k_halo_send");
  }
  int halo_wait(const PeerHalo& ph, hipStream_t s) {
    hipLaunchKernelGGL(k_halo_wait, dim3(ph.grid), dim3(kBlock), 0, s, ph.dev, spin_limit);
    return fail(hipGetLastError(), "k_halo_wait");
  }
  // a wait gave up since the last call?  (host synchronisation points)
  int take_error(hipStream_t s) {
    int e = 0;
    if (hipMemcpyAsync(&e, derr, sizeof e, hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    // (not reset: the ranks' sequence numbers are out of step from here on -
    // the communicator stays failed, every later call reports it at once)
    if (e) err = "peer exchange: a neighbour's data did not arrive (rank gone or out of step)";
    return e;
  }
};

// Hand a released operator's channel back to its backend's free list.  The
// drain runs on the OWNER's device (a pcd_destroy / GC thread, or thread ranks
// on several GPUs, may call with another one current) and OUTSIDE the
// process-wide lock: a device-wide wait under it would serialise all thread
// ranks, and one waiting on a kernel that spins for a rank waiting on the lock
// could only leave through the spin time-out.
inline void peer_give_back(PeerHalo& ph) {
  if (!ph.owner) return;
  int dev = -1;
  {
    std::lock_guard<std::mutex> lk(peer_live_mu());
    if (peer_live().count(ph.owner)) dev = ph.owner->device;
  }
  if (dev >= 0) {
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    (void)hipDeviceSynchronize();
    if (cur >= 0 && cur != dev) (void)hipSetDevice(cur);
    std::lock_guard<std::mutex> lk(peer_live_mu());
    if (peer_live().count(ph.owner)) ph.owner->give_back(ph);
  }
  ph.owner = nullptr;
}

}  // namespace pcd
