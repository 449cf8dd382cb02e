// kyber_hip.hip -- C-ABI of libkyber_hip.so (include/kyber_batch.h,
// include/kyber_api.h): workspace management and the kernel sequences for
// batched keypair / encaps / decaps.  There is no CPU fallback anywhere in
// this file: every result is produced by the gfx950 kernels in kem_kernels.h,
// kem_kernels_90s.h and arith_kernels.h (the latter through arith_launch.h).
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <thread>
#include <algorithm>
#include <vector>
#include <map>
#include <string>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>
#include <strings.h>
#include <sys/random.h>
#include <unistd.h>
#include <pthread.h>

#include "../../include/kyber_batch.h"
#include "../../include/kyber_api.h"
#include "../../include/mlkem_batch.h"
#include "kem_kernels_90s.h"
#include "keccak_split.h"
#include "arith_launch.h"

using namespace ky;

namespace {

thread_local char g_err[256] = "";
static const bool g_trace = getenv("KYBER_HIP_TRACE") != nullptr;
#define TR(...) do { if (g_trace) { fprintf(stderr, "[ky %lx] ", (unsigned long)pthread_self()); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

int hip_fail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  return KYBER_HIP_EHIP;
}
#define HIP_TRY(call)                                    \
  do {                                                   \
    hipError_t e_ = (call);                              \
    if (e_ != hipSuccess) return hip_fail(e_, #call);    \
  } while (0)

// ---- per-device workspace --------------------------------------------------
constexpr int MAX_DEV = 64;
constexpr size_t DEFAULT_CHUNK = 1u << 19;  // 3 GB of workspace for Kyber768; kernel-boundary tails amortised (measured: 2^17 48.7, 2^18 51.0, 2^19 51.5 M/s)
struct Workspace {
  void* base = nullptr;
  size_t bytes = 0;
  hipEvent_t idle = nullptr;        // recorded after the last kernel that uses the workspace
  hipStream_t last = nullptr;       // ... on this stream
  bool used = false;
  uint64_t stamp = 0;               // order of last use among the device's workspaces
};
// Workspaces per device.  One by default: the GPU work of any two pipeline calls on a device is then ordered
// (WsUse).  kyber_hip_set_workspace_slots(s > 1) lets batches up to POOL_MAX_N instances that arrive on DIFFERENT
// streams take different workspaces, so their kernels overlap on the device -- small batches leave most of it
// idle ("small batches") and a server with several request threads wants them side by side.
// Every workspace on its own keeps the one-workspace protocol, so the choice of slot is a matter of speed only.
// Larger batches fill the chip alone and always take slot 0; a call under stream capture takes the workspace
// a call outside capture laid out for its size (WsUse).
constexpr int MAX_SLOTS = 8;
constexpr size_t POOL_MAX_N = 65536;
Workspace g_ws[MAX_DEV][MAX_SLOTS];
uint64_t g_ws_stamp[MAX_DEV];
std::atomic<int> g_slots{1};
std::atomic<size_t> g_chunk{DEFAULT_CHUNK};
std::atomic<size_t> g_tail_cap_max{(size_t)-1};
// One lock per device (workspace + launch order), so that host threads driving
// different GPUs do not serialise each other.  Indexed by the calling thread's
// current device.
std::mutex g_dev_mu[MAX_DEV];
std::mutex g_prof_mu;
inline std::mutex& dev_mutex(std::mutex (&arr)[MAX_DEV]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) dev = 0;
  return arr[dev];
}

int ws_reserve(int slot, size_t bytes, uint8_t** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEV || slot < 0 || slot >= MAX_SLOTS) return KYBER_HIP_EINVAL;
  Workspace& w = g_ws[dev][slot];
  if (w.bytes < bytes) {
    TR("ws_reserve slot %d grow %zu -> %zu", slot, w.bytes, bytes);
    if (w.base) {
      HIP_TRY(hipDeviceSynchronize());
      TR("ws_reserve synced");
      HIP_TRY(hipFree(w.base));
      w.base = nullptr;
      w.bytes = 0;
    }
    HIP_TRY(hipMalloc(&w.base, bytes));
    w.bytes = bytes;
    TR("ws_reserve done");
  }
  *out = static_cast<uint8_t*>(w.base);
  return 0;
}

// The slot for a batch of up to POOL_MAX_N instances outside capture, by speed: the one this stream used last
// (ordered behind it anyway), else one never used, else one whose last call has finished, else the one used
// longest ago (the call then waits for it).
int pick_slot(int dev, int nslot, hipStream_t st) {
  Workspace* ws = g_ws[dev];
  int best = -1;
  for (int s = 0; s < nslot; s++)
    if (ws[s].used && ws[s].last == st && (best < 0 || ws[s].stamp > ws[best].stamp)) best = s;
  if (best >= 0) return best;
  for (int s = 0; s < nslot; s++)
    if (!ws[s].used) return s;
  for (int s = 0; s < nslot; s++) {
    if (hipEventQuery(ws[s].idle) == hipSuccess) return s;
    (void)hipGetLastError();  // "not ready" is an answer, not a failure of the launch that follows
  }
  best = 0;
  for (int s = 1; s < nslot; s++)
    if (ws[s].stamp < ws[best].stamp) best = s;
  return best;
}
bool stream_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return true;  // unknown: the conservative answer
  }
  return cs != hipStreamCaptureStatusNone;
}

// The GPU work of two pipeline calls that share a workspace must not overlap even when
// they are queued on different streams (or from different
// host threads): a call first makes its stream wait for the event the previous
// call recorded behind its last kernel.  Calls on one stream cost nothing extra.
struct WsUse {
  hipStream_t st;
  Workspace* w = nullptr;
  int slot = 0;
  int dev = 0;
  int rc = 0;  // checked by the caller before it launches anything
  bool passive = false;  // a captured call among several workspaces: takes one, leaves no trace in its state
  // need: the bytes the call will ask ws_reserve for
  WsUse(hipStream_t s, size_t n, size_t need) : st(s) {
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { rc = hip_fail(e, "hipGetDevice"); return; }
    if (dev < 0 || dev >= MAX_DEV) { rc = KYBER_HIP_EINVAL; return; }
    const int nslot = g_slots.load();
    const bool capturing = nslot > 1 && stream_capturing(st);
    const bool pool = nslot > 1 && !capturing;
    if (pool && n <= POOL_MAX_N) slot = pick_slot(dev, nslot, st);
    if (capturing) {
      // A captured call cannot allocate, so it takes the workspace that is large enough and was used last (the one
      // a call of this size outside capture laid out), else slot 0 as with one workspace.  What it records runs at
      // replay time only: there is nothing to wait for now and nothing for later calls to wait for, and the caller
      // orders replays against other calls, as the header says.
      int best = -1;
      for (int i = 0; i < nslot; i++)
        if (g_ws[dev][i].bytes >= need && (best < 0 || g_ws[dev][i].stamp > g_ws[dev][best].stamp)) best = i;
      if (best >= 0) slot = best;
      TR("WsUse n %zu st %p slot %d (captured)", n, (void*)st, slot);
      w = &g_ws[dev][slot];
      passive = true;
      return;
    }
    Workspace* ws = &g_ws[dev][slot];
    TR("WsUse n %zu st %p slot %d used %d last %p", n, (void*)st, slot, (int)ws->used, (void*)ws->last);
    if (!ws->idle && (e = hipEventCreateWithFlags(&ws->idle, hipEventDisableTiming)) != hipSuccess) {
      ws->idle = nullptr;
      rc = hip_fail(e, "hipEventCreateWithFlags");
      return;
    }
    // an unordered second user of the one workspace would corrupt both calls: fail instead.
    // With several workspaces the HOST waits for the workspace's last call (normally finished long ago: pick_slot
    // prefers such a slot) and no stream ever waits on the device for another: a stream that does holds up its
    // hardware queue, which other callers' streams share.  (The standstill met while this mode was built -- 8
    // streams on 4 workspaces, two of them never drained -- went away with the second stream, see side_for; the
    // host-side wait is the cautious half of the same rule.)
    if (ws->used && ws->last != st) {
      if (pool) {
        TR("WsUse host wait");
        if ((e = hipEventSynchronize(ws->idle)) != hipSuccess) {
          rc = hip_fail(e, "hipEventSynchronize");
          return;
        }
      } else if ((e = hipStreamWaitEvent(st, ws->idle, 0)) != hipSuccess) {
        rc = hip_fail(e, "hipStreamWaitEvent");
        return;
      }
    }
    w = ws;
  }
  ~WsUse() {
    TR("~WsUse slot %d", slot);
    if (!w || passive) return;
    if (hipEventRecord(w->idle, st) != hipSuccess) {
      // the next call cannot be ordered behind this one on the device: order it on the host
      (void)hipStreamSynchronize(st);
      w->used = false;
      return;
    }
    w->last = st;
    w->used = true;
    w->stamp = ++g_ws_stamp[dev];
  }
};

template <class P>
struct Layout {
  static constexpr size_t MAT = (size_t)P::NMAT * 512;
  static constexpr size_t PRF = (size_t)(2 * P::K + 1) * P::PRF_STRIDE;
  static constexpr size_t PER = MAT + PRF + SMALL;
  int16_t* mat;
  uint8_t* prf;
  uint8_t* small;
  uint32_t* tailq;  // k_sample_matrix's queue of unfinished entries
  uint32_t tail_cap;
  int16_t* smat;    // shared-key calls: the one expanded matrix
  uint8_t* hpk;     // shared-key calls: H(pk)
  static constexpr size_t tq_bytes(size_t chunk) { return (tailq_bytes(chunk * P::NMAT) + 255) & ~(size_t)255; }
  static constexpr size_t need(size_t chunk) { return chunk * PER + tq_bytes(chunk) + MAT + 256; }
  void place(uint8_t* b, size_t chunk) {
    const size_t tq = tq_bytes(chunk);
    mat = reinterpret_cast<int16_t*>(b);
    prf = b + chunk * MAT;
    small = prf + chunk * PRF;
    tailq = reinterpret_cast<uint32_t*>(small + chunk * SMALL);
    smat = reinterpret_cast<int16_t*>(small + chunk * SMALL + tq);
    hpk = reinterpret_cast<uint8_t*>(smat) + MAT;
    const size_t cap = tailq_capacity(chunk * P::NMAT);
    const size_t cap_max = g_tail_cap_max.load();
    tail_cap = (uint32_t)(cap < cap_max ? cap : cap_max);
  }
  int carve(size_t chunk, const WsUse& use) {
    uint8_t* b;
    int rc = ws_reserve(use.slot, need(chunk), &b);
    if (rc) return rc;
    place(b, chunk);
    return 0;
  }
};

// SHAKE128 matrix expansion: the main pass gives every entry three blocks, the
// tail pass finishes the ~1 % that need more.
// CUs of the current device (the AES kernels choose their workgroup size by it)
int device_cus() {
  static std::atomic<int> cus[MAX_DEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 256;
  int c = cus[dev].load(std::memory_order_relaxed);
  if (c == 0) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    cus[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}
// the AES kernels: workgroup size by the amount of work (aes_wg)
#define AES_LAUNCH_WG(name, st, kern, W, jobs_, cus_, args) \
  RUN(name, st, kern<P, W><<<blocks(jobs_, W), W, 0, st>>> args)
#define AES_LAUNCH(name, st, kern, jobs, args)                                                      \
  do {                                                                                              \
    const size_t jobs_ = (jobs);                                                                    \
    const int cus_ = device_cus(), wg_ = aes_wg((long)jobs_, cus_);                                 \
    if (wg_ == AES_WG) AES_LAUNCH_WG(name, st, kern, AES_WG, jobs_, cus_, args);                    \
    else if (wg_ == AES_WG_MID) AES_LAUNCH_WG(name, st, kern, AES_WG_MID, jobs_, cus_, args);       \
    else AES_LAUNCH_WG(name, st, kern, AES_WG_SMALL, jobs_, cus_, args);                            \
  } while (0)

template <class P>
int sample_matrix(int entries, const uint8_t* rho, size_t rho_stride, int transposed, int16_t* mat, uint32_t* tailq,
                  uint32_t tail_cap, hipStream_t st);

inline unsigned blocks(size_t items, unsigned per) { return (unsigned)((items + per - 1) / per); }

#define LAUNCH_CHECK(name)                                      \
  do {                                                          \
    hipError_t e_ = hipGetLastError();                          \
    if (e_ != hipSuccess) return hip_fail(e_, "launch " name);  \
  } while (0)

// ---- optional per-kernel timing (HIP events on the launch stream) -----------------
// Off by default; bench.py switches it on for a separate, untimed pass to get
// each kernel's average launch duration for the roofline line.
struct ProfRec {
  const char* name;
  hipEvent_t a, b;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;

inline void prof_begin(hipStream_t st, hipEvent_t* a) {
  if (!g_prof_on) return;
  (void)hipEventCreate(a);
  (void)hipEventRecord(*a, st);
}
inline void prof_end(const char* name, hipStream_t st, hipEvent_t a) {
  if (!g_prof_on) return;
  hipEvent_t b;
  (void)hipEventCreate(&b);
  (void)hipEventRecord(b, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(ProfRec{name, a, b});
}
#define RUN(name, st, ...)        \
  do {                            \
    TR("run %s", name);           \
    hipEvent_t ev_a_{};           \
    prof_begin(st, &ev_a_);       \
    __VA_ARGS__;                  \
    LAUNCH_CHECK(name);           \
    prof_end(name, st, ev_a_);    \
  } while (0)

__global__ void __launch_bounds__(64) k_zero32(uint32_t* p) {
  if (threadIdx.x == 0) *p = 0u;
}

// An FO-hash kernel of kem_kernels.h, or for small batches its stand-in on work-item pairs (keccak_split.h)
#define RUN_HASH(name, st, m_, kern, kern2, ...)                                                                                  \
  do {                                                                                                                            \
    if ((size_t)(m_) <= SPLIT_MAX_N) RUN(name, st, kern2<<<blocks(2 * (size_t)(m_), SPLIT_WG), SPLIT_WG, 0, st>>>(__VA_ARGS__));     \
    else RUN(name, st, kern<<<blocks((m_), sponge_wg(m_)), sponge_wg(m_), 0, st>>>(__VA_ARGS__));                                   \
  } while (0)

template <class P>
int sample_matrix(int entries, const uint8_t* rho, size_t rho_stride, int transposed, int16_t* mat, uint32_t* tailq,
                  uint32_t tail_cap, hipStream_t st) {
  // the queue's counter starts at zero.  A kernel, not hipMemsetAsync: as a memset NODE of a captured graph
  // (kyber_batch_dec_dev_noalloc under stream capture) the clearing took effect on the first launch of the
  // graph only -- every replay after it gave a wrong secret for n not a multiple of 8 (tools/graph_replay_check.py).
  k_zero32<<<1, 64, 0, st>>>(tailq);
  LAUNCH_CHECK("k_zero32");
  if (sponge_wg(entries) == SPONGE_WG)
    RUN("k_sample_matrix", st,
        (k_sample_matrix<P, SPONGE_WG><<<blocks((size_t)entries, SPONGE_WG), SPONGE_WG, 0, st>>>(entries, rho, rho_stride, transposed, mat, tailq, tail_cap)));
  else
    RUN("k_sample_matrix", st,
        (k_sample_matrix<P, SPONGE_WG_SMALL><<<blocks((size_t)entries, SPONGE_WG_SMALL), SPONGE_WG_SMALL, 0, st>>>(entries, rho, rho_stride, transposed, mat, tailq, tail_cap)));
  if (tail_cap) RUN("k_sample_matrix_tail", st, k_sample_matrix_tail<<<blocks(tail_cap, 64), 64, 0, st>>>(tailq, tail_cap, mat));
  return 0;
}
// small batches (keccak_split.h, "front kernels"): the matrix expansion as some of the workgroups of one launch
// that also runs the hashes in front of an encapsulation / the PRF of a key generation
template <class P>
int enc_front(int m, const uint8_t* coins, const uint8_t* pk, Layout<P>& L, hipStream_t st) {
  k_zero32<<<1, 64, 0, st>>>(L.tailq);
  LAUNCH_CHECK("k_zero32");
  const unsigned nb_pre = blocks(2 * (size_t)m, SPLIT_WG), nb_mat = blocks((size_t)m * P::NMAT, SPONGE_WG_SMALL);
  RUN("k_enc_front", st, k_enc_front<P><<<nb_pre + nb_mat, SPLIT_WG, 0, st>>>(m, (int)nb_pre, coins, pk, P::PKB, L.small, L.mat, L.tailq, L.tail_cap));
  if (L.tail_cap) RUN("k_sample_matrix_tail", st, k_sample_matrix_tail<<<blocks(L.tail_cap, 64), 64, 0, st>>>(L.tailq, L.tail_cap, L.mat));
  return 0;
}
template <class P>
int kp_front(int m, Layout<P>& L, hipStream_t st) {
  k_zero32<<<1, 64, 0, st>>>(L.tailq);
  LAUNCH_CHECK("k_zero32");
  const unsigned nb_mat = blocks((size_t)m * P::NMAT, SPONGE_WG_SMALL), nb_prf = blocks((size_t)m * 2 * P::K, SPLIT_WG);
  RUN("k_kp_front", st, k_kp_front<P><<<nb_mat + nb_prf, SPLIT_WG, 0, st>>>(m, (int)nb_mat, L.small, L.mat, L.tailq, L.tail_cap, L.prf));
  if (L.tail_cap) RUN("k_sample_matrix_tail", st, k_sample_matrix_tail<<<blocks(L.tail_cap, 64), 64, 0, st>>>(L.tailq, L.tail_cap, L.mat));
  return 0;
}

// Matrix expansion for `insts` instances (SHAKE128 or AES-256-CTR XOF), Aᵀ when transposed.
template <class P, bool N90>
int expand_matrix(int insts, const uint8_t* rho, size_t rho_stride, int transposed, int16_t* mat, Layout<P>& L, hipStream_t st) {
  const int entries = insts * P::NMAT;
  if constexpr (N90) {
    AES_LAUNCH("k90_sample_matrix", st, k90_sample_matrix, (size_t)entries, (entries, rho, rho_stride, transposed, mat));
    return 0;
  } else {
    return sample_matrix<P>(entries, rho, rho_stride, transposed, mat, L.tailq, L.tail_cap, st);
  }
}

// ---- small batches: the matrix expansion beside the hashes --------------------------------
// Up to a few ten thousand instances no kernel fills the chip and a call's time is its chain of dependent
// launches, each as long as its longest lane (a permutation on one lane takes ~10 us).  The matrix
// expansion (three dependent permutations) needs only rho, so for such batches it runs on a second
// stream beside the hashes in front of it (encapsulation: H(m), H(pk), G) and the caller's stream
// picks it up again before the PRF.  Not under stream
// capture, and not in the no-allocation entry point (one stream there, by contract).
// The two cross-stream hops cost ~25 us themselves (profiles/r03_ab_latency_side_stream.txt,
// r03_ab_latency_side_stream_dev.txt): an encapsulation hides the expansion behind eleven permutations
// and gains at every size (device pointers, n = 1: 233 -> 215 us, 256: 233 -> 200, 4096: 244 -> 221); a
// decapsulation has only the decryption and one permutation in front and measured -4 % to +3 %
// depending on size and on the caller's stream, so it stays on one stream.  Threshold: n = 16384 330 -> 303 us,
// 32768 437 -> 412, 65536 680 -> 690 (profiles/r03_ab_midsize_thresholds.txt).
#ifndef KY_SIDE_MAX_N
#define KY_SIDE_MAX_N 32768  // tools/build_variant.sh noside -DKY_SIDE_MAX_N=0: the A/B of this path
#endif
constexpr size_t SIDE_MAX_N = KY_SIDE_MAX_N;
struct Side {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
Side g_side[MAX_DEV][MAX_SLOTS];  // one per workspace, under g_dev_mu like it

Side* side_for(size_t n, hipStream_t st, int slot) {
  if (n > SIDE_MAX_N) return nullptr;
  // Several workspaces: calls overlap with each other, which fills the device better than a call's own second
  // stream -- and the fork/join of that stream between several streams that are busy at once stalled two of them
  // for good (tools/slots_stress.py 4 25 0 1 2 7 4 with the second stream; without it every mix runs).  One stream
  // per call there, as in the no-allocation entry point.
  if (g_slots.load() > 1) return nullptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  Side& sd = g_side[dev][slot];
  if (!sd.s) {
    Side t;
    // Highest priority: the runtime keeps the hardware queues of a priority level apart from those of the
    // other levels, so this stream never shares a queue with a caller's (normal-priority) stream --
    // on a shared queue the fork is a detour, not an overlap (profiles/r03_ab_latency_side_stream_mixed.txt).
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_greatest = 0;
    if (hipStreamCreateWithPriority(&t.s, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
        hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&t.join, hipEventDisableTiming) != hipSuccess) {
      if (t.s) (void)hipStreamDestroy(t.s);
      if (t.fork) (void)hipEventDestroy(t.fork);
      (void)hipGetLastError();
      return nullptr;  // the call then runs on one stream, as large batches do
    }
    sd = t;
  }
  return &sd;
}
// everything queued on st so far (the previous user of the workspace included) comes first
int side_begin(Side* sd, hipStream_t st) {
  TR("side_begin");
  HIP_TRY(hipEventRecord(sd->fork, st));
  HIP_TRY(hipStreamWaitEvent(sd->s, sd->fork, 0));
  return 0;
}
int side_end(Side* sd) {
  HIP_TRY(hipEventRecord(sd->join, sd->s));
  return 0;
}
int side_join(Side* sd, hipStream_t st) {
  TR("side_join");
  HIP_TRY(hipStreamWaitEvent(st, sd->join, 0));
  return 0;
}
// expand_matrix on the side stream when there is one, else on st; *pending tells the caller to join
template <class P, bool N90>
int expand_matrix_beside(Side* sd, bool* pending, int insts, const uint8_t* rho, size_t rho_stride, int transposed, int16_t* mat,
                         Layout<P>& L, hipStream_t st) {
  if (!sd) return expand_matrix<P, N90>(insts, rho, rho_stride, transposed, mat, L, st);
  if (int rc = side_begin(sd, st)) return rc;
  if (int rc = expand_matrix<P, N90>(insts, rho, rho_stride, transposed, mat, L, sd->s)) return rc;
  if (int rc = side_end(sd)) return rc;
  *pending = true;
  return 0;
}
#define SIDE_JOIN(sd, pending, st)                  \
  do {                                              \
    if (pending) {                                  \
      if (int rc_ = side_join(sd, st)) return rc_;  \
      pending = false;                              \
    }                                               \
  } while (0)

// ---- the three pipelines (device pointers) -----------------------------------
template <class P, bool N90>
int keypair_dev(size_t n, uint8_t* pk, uint8_t* sk, const uint8_t* coins, hipStream_t st) {
  std::lock_guard<std::mutex> lk(dev_mutex(g_dev_mu));
  const size_t gc = g_chunk.load();
  const size_t chunk = n < gc ? n : gc;
  WsUse use(st, n, Layout<P>::need(chunk));
  if (use.rc) return use.rc;
  Layout<P> L;
  if (int rc = L.carve(chunk, use)) return rc;
  for (size_t o = 0; o < n; o += chunk) {
    const int m = (int)(n - o < chunk ? n - o : chunk);
    const uint8_t* c = coins + o * 64;
    uint8_t *p = pk + o * P::PKB, *s = sk + o * P::SKB;
    if constexpr (N90) RUN("k90_kp_pre", st, k90_kp_pre<<<blocks(m, 256), 256, 0, st>>>(m, c, L.small));
    else RUN_HASH("k_kp_pre", st, m, k_kp_pre, k2_kp_pre, m, c, L.small);
    if constexpr (N90) {
      if (int rc = expand_matrix<P, N90>(m, L.small, SMALL, 0, L.mat, L, st)) return rc;
      AES_LAUNCH("k90_prf", st, k90_prf, (size_t)m * 2 * P::K, (m, 2 * P::K, 2 * P::K, L.small, 32, L.prf));
    } else if ((size_t)m <= FRONT_MAX_N) {
      if (int rc = kp_front<P>(m, L, st)) return rc;
    } else {
      if (int rc = expand_matrix<P, N90>(m, L.small, SMALL, 0, L.mat, L, st)) return rc;
      RUN("k_prf", st, k_prf<P><<<blocks((size_t)m * 2 * P::K, sponge_wg((size_t)m * 2 * P::K)), sponge_wg((size_t)m * 2 * P::K), 0, st>>>(m, 2 * P::K, 2 * P::K, L.small, 32, L.prf));
    }
    RUN("k_kp_arith", st, launch_kp_arith<P>(st, m, L.small, L.mat, L.prf, p, s));
    if constexpr (N90) {
      RUN("k90_kp_post", st, k90_kp_post<P><<<blocks(m, 256), 256, 0, st>>>(m, c, p, s));
    } else {
      RUN_HASH("k_kp_post", st, m, k_kp_post<P>, k2_kp_post<P>, m, c, p, s);
    }
  }
  return 0;
}

// Encapsulation.  pk_stride == PKB: n keys (ref/kem.c:51-76 n times);
// pk_stride == 0: n encapsulations to ONE key -- H(pk) and the matrix are
// computed once and shared; the bytes written are the same as
// with n copies of the key.
// pmat / phpk: the key's expanded matrix and H(pk) from a prepared-key handle (shared calls only);
// without them a shared call computes both into the workspace first.
template <class P, bool N90>
int enc_dev(size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* pk, size_t pk_stride, const uint8_t* coins, hipStream_t st,
            const int16_t* pmat = nullptr, const uint8_t* phpk = nullptr) {
  std::lock_guard<std::mutex> lk(dev_mutex(g_dev_mu));
  const size_t gc = g_chunk.load();
  const size_t chunk = n < gc ? n : gc;
  WsUse use(st, n, Layout<P>::need(chunk));
  if (use.rc) return use.rc;
  const bool shared = pk_stride == 0;
  Layout<P> L;
  if (int rc = L.carve(chunk, use)) return rc;
  const bool front = !N90 && !shared && n <= FRONT_MAX_N;  // one launch for the hashes in front and the matrix expansion
  Side* sd = front ? nullptr : side_for(n, st, use.slot);
  bool pending = false;
  if (shared && !pmat) {
    if (int rc = expand_matrix_beside<P, N90>(sd, &pending, 1, pk + P::PVB, 0, 1, L.smat, L, st)) return rc;
    if constexpr (N90) RUN("k90_hash_pk", st, k90_hash_pk<P><<<1, 64, 0, st>>>(pk, L.hpk));
    else RUN("k_hash_pk", st, k2_hash_pk<P><<<1, SPLIT_WG, 0, st>>>(pk, L.hpk));
  }
  const uint8_t* hpk = shared ? (pmat ? phpk : L.hpk) : nullptr;
  const int16_t* mat = shared ? (pmat ? pmat : L.smat) : L.mat;
  const size_t mat_stride = shared ? 0 : (size_t)P::NMAT * 256;
  for (size_t o = 0; o < n; o += chunk) {
    const int m = (int)(n - o < chunk ? n - o : chunk);
    const uint8_t* p = pk + o * pk_stride;
    uint8_t* c = ct + o * P::CTB;
    if (!shared && !front)
      if (int rc = expand_matrix_beside<P, N90>(sd, &pending, m, p + P::PVB, P::PKB, 1, L.mat, L, st)) return rc;
    if constexpr (N90) {
      RUN("k90_enc_pre", st, k90_enc_pre<P><<<blocks(m, 256), 256, 0, st>>>(m, coins + o * 32, p, hpk, L.small));
    } else if (front) {
      if (int rc = enc_front<P>(m, coins + o * 32, p, L, st)) return rc;
    } else {
      RUN_HASH("k_enc_pre", st, m, k_enc_pre<P>, k2_enc_pre<P>, m, coins + o * 32, p, hpk, L.small);
    }
    SIDE_JOIN(sd, pending, st);
    if constexpr (N90) {
      AES_LAUNCH("k90_prf", st, k90_prf, (size_t)m * (2 * P::K + 1), (m, 2 * P::K + 1, P::K, L.small, 64, L.prf));
    } else {
      RUN("k_prf", st, k_prf<P><<<blocks((size_t)m * (2 * P::K + 1), sponge_wg((size_t)m * (2 * P::K + 1))), sponge_wg((size_t)m * (2 * P::K + 1)), 0, st>>>(m, 2 * P::K + 1, P::K, L.small, 64, L.prf));
    }
    RUN("k_enc_arith", st, launch_enc_arith<P, false>(st, m, p, pk_stride, L.small, mat, mat_stride, L.prf, c));
    if constexpr (N90) {
      RUN("k90_post", st, k90_post<P, false><<<blocks(m, 256), 256, 0, st>>>(m, c, L.small, nullptr, 0, ss + o * 32));
    } else {
      RUN_HASH("k_post", st, m, (k_post<P, false>), (k2_post<P, false>), m, c, L.small, nullptr, 0, ss + o * 32);
    }
  }
  return 0;
}

// Decapsulation.  sk_stride == SKB: n keys (ref/kem.c:94-127 n times);
// sk_stride == 0: n ciphertexts for ONE secret key, matrix expanded once.
// the kernel sequence of a decapsulation over a workspace that is already laid out
template <class P, bool N90>
int dec_run(size_t n, size_t chunk, Layout<P>& L, uint8_t* ss, const uint8_t* ct, const uint8_t* sk, size_t sk_stride, hipStream_t st,
            const int16_t* pmat = nullptr, Side* sd = nullptr) {
  const bool shared = sk_stride == 0;
  bool pending = false;
  if (shared && !pmat)
    if (int rc = expand_matrix<P, N90>(1, sk + 2 * P::PVB, 0, 1, L.smat, L, st)) return rc;
  const int16_t* mat = shared ? (pmat ? pmat : L.smat) : L.mat;
  const size_t mat_stride = shared ? 0 : (size_t)P::NMAT * 256;
  for (size_t o = 0; o < n; o += chunk) {
    const int m = (int)(n - o < chunk ? n - o : chunk);
    const uint8_t* c = ct + o * P::CTB;
    const uint8_t* s = sk + o * sk_stride;
    // Batches up to FRONT_MAX_N (SHAKE sets, a key per instance): the three jobs between the decryption and the PRF --
    // G, the matrix expansion and H(ct), which leaves the end of the chain for it -- are one launch (keccak_split.h:
    // k_dec_middle), and the last kernel is the one permutation of the KDF.  Same-box A/B, device pointers
    // (profiles/r03_ab_latency_front_kernels.txt): n = 1 162 -> 135 us, 4096 183 -> 150, 16384 262 -> 232, 65536 a tie.
    const bool middle = !N90 && !shared && (size_t)m <= FRONT_MAX_N;
    // One key for the whole batch (shared-key and prepared calls): no matrix to expand, so nothing long enough to hide
    // H(ct) behind in one launch -- its nine permutations are longer than the rest of the chain together.  They run on
    // the second stream from the start of the call and the caller's stream picks them up in front of the KDF.
    const bool side_h = !N90 && shared && sd != nullptr && (size_t)m <= FRONT_MAX_N;
    if (side_h) {
      if constexpr (!N90) {
        if (int rc = side_begin(sd, st)) return rc;
        RUN("k_hash_ct", sd->s, k2_hash_ct<P><<<blocks(2 * (size_t)m, SPLIT_WG), SPLIT_WG, 0, sd->s>>>(m, c, ss + o * 32));
        if (int rc = side_end(sd)) return rc;
        pending = true;
      }
    }
    RUN("k_dec_arith", st, launch_dec_arith<P>(st, m, c, s, sk_stride, L.small));
    if (middle) {
      if constexpr (!N90) {
        k_zero32<<<1, 64, 0, st>>>(L.tailq);
        LAUNCH_CHECK("k_zero32");
        const unsigned nb_mat = blocks((size_t)m * P::NMAT, SPONGE_WG_SMALL), nb_h = blocks(2 * (size_t)m, SPLIT_WG);
        RUN("k_dec_middle", st, k_dec_middle<P><<<nb_mat + 2 * nb_h, SPLIT_WG, 0, st>>>(m, (int)nb_mat, (int)nb_h, s, L.mat, L.tailq, L.tail_cap, c, ss + o * 32, L.small));
        if (L.tail_cap) RUN("k_sample_matrix_tail", st, k_sample_matrix_tail<<<blocks(L.tail_cap, 64), 64, 0, st>>>(L.tailq, L.tail_cap, L.mat));
      }
    } else {
      if constexpr (N90) {
        RUN("k90_dec_mid", st, k90_dec_mid<P><<<blocks(m, 256), 256, 0, st>>>(m, s, sk_stride, L.small));
      } else {
        RUN_HASH("k_dec_mid", st, m, k_dec_mid<P>, k2_dec_mid<P>, m, s, sk_stride, L.small);
      }
      if (!shared)
        if (int rc = expand_matrix<P, N90>(m, s + 2 * P::PVB, P::SKB, 1, L.mat, L, st)) return rc;
    }
    if constexpr (N90) {
      AES_LAUNCH("k90_prf", st, k90_prf, (size_t)m * (2 * P::K + 1), (m, 2 * P::K + 1, P::K, L.small, 64, L.prf));
    } else {
      RUN("k_prf", st, k_prf<P><<<blocks((size_t)m * (2 * P::K + 1), sponge_wg((size_t)m * (2 * P::K + 1))), sponge_wg((size_t)m * (2 * P::K + 1)), 0, st>>>(m, 2 * P::K + 1, P::K, L.small, 64, L.prf));
    }
    // re-encrypt with pk = sk + PVB and compare against ct (ct is only read)
    RUN("k_enc_arith_cmp", st,
        launch_enc_arith<P, true>(st, m, s + P::PVB, sk_stride, L.small, mat, mat_stride, L.prf, const_cast<uint8_t*>(c)));
    if constexpr (N90) {
      RUN("k90_post", st, k90_post<P, true><<<blocks(m, 256), 256, 0, st>>>(m, c, L.small, s, sk_stride, ss + o * 32));
    } else {
      SIDE_JOIN(sd, pending, st);
      if (middle || side_h) RUN("k_kdf_dec", st, k2_kdf_dec<P><<<blocks(2 * (size_t)m, SPLIT_WG), SPLIT_WG, 0, st>>>(m, L.small, s, sk_stride, ss + o * 32));
      else RUN_HASH("k_post", st, m, (k_post<P, true>), (k2_post<P, true>), m, c, L.small, s, sk_stride, ss + o * 32);
    }
  }
  return 0;
}

template <class P, bool N90>
int dec_dev(size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* sk, size_t sk_stride, hipStream_t st, const int16_t* pmat = nullptr) {
  std::lock_guard<std::mutex> lk(dev_mutex(g_dev_mu));
  const size_t gc = g_chunk.load();
  const size_t chunk = n < gc ? n : gc;
  WsUse use(st, n, Layout<P>::need(chunk));
  if (use.rc) return use.rc;
  Layout<P> L;
  if (int rc = L.carve(chunk, use)) return rc;
  return dec_run<P, N90>(n, chunk, L, ss, ct, sk, sk_stride, st, pmat, sk_stride == 0 ? side_for(n, st, use.slot) : nullptr);
}

// ---- ML-KEM (FIPS 203) pipelines: the Kyber kernel sequences with the standard's FO glue ----
template <class P>
int ml_keypair_dev(size_t n, uint8_t* ek, uint8_t* dk, const uint8_t* coins, hipStream_t st) {
  std::lock_guard<std::mutex> lk(dev_mutex(g_dev_mu));
  const size_t gc = g_chunk.load();
  const size_t chunk = n < gc ? n : gc;
  WsUse use(st, n, Layout<P>::need(chunk));
  if (use.rc) return use.rc;
  Layout<P> L;
  if (int rc = L.carve(chunk, use)) return rc;
  for (size_t o = 0; o < n; o += chunk) {
    const int m = (int)(n - o < chunk ? n - o : chunk);
    const uint8_t* c = coins + o * 64;
    uint8_t *p = ek + o * P::PKB, *s = dk + o * P::SKB;
    RUN_HASH("k_ml_kp_pre", st, m, k_ml_kp_pre, k2_ml_kp_pre, m, P::K, c, L.small);
    if ((size_t)m <= FRONT_MAX_N) {  // small batches: matrix expansion and PRF as one launch, as in keypair_dev
      if (int rc = kp_front<P>(m, L, st)) return rc;
    } else {
      if (int rc = expand_matrix<P, false>(m, L.small, SMALL, 0, L.mat, L, st)) return rc;
      RUN("k_prf", st, k_prf<P><<<blocks((size_t)m * 2 * P::K, sponge_wg((size_t)m * 2 * P::K)), sponge_wg((size_t)m * 2 * P::K), 0, st>>>(m, 2 * P::K, 2 * P::K, L.small, 32, L.prf));
    }
    RUN("k_kp_arith", st, launch_kp_arith<P>(st, m, L.small, L.mat, L.prf, p, s));
    RUN_HASH("k_kp_post", st, m, k_kp_post<P>, k2_kp_post<P>, m, c, p, s);  // dk tail = H(ek) || z, as in round 3
  }
  return 0;
}

template <class P>
int ml_enc_dev(size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* ek, const uint8_t* coins, hipStream_t st) {
  std::lock_guard<std::mutex> lk(dev_mutex(g_dev_mu));
  const size_t gc = g_chunk.load();
  const size_t chunk = n < gc ? n : gc;
  WsUse use(st, n, Layout<P>::need(chunk));
  if (use.rc) return use.rc;
  Layout<P> L;
  if (int rc = L.carve(chunk, use)) return rc;
  for (size_t o = 0; o < n; o += chunk) {
    const int m = (int)(n - o < chunk ? n - o : chunk);
    const uint8_t* p = ek + o * P::PKB;
    uint8_t* c = ct + o * P::CTB;
    if ((size_t)m <= FRONT_MAX_N) {  // small batches: H(ek), G and the matrix expansion as one launch (keccak_split.h)
      k_zero32<<<1, 64, 0, st>>>(L.tailq);
      LAUNCH_CHECK("k_zero32");
      const unsigned nb_pre = blocks(2 * (size_t)m, SPLIT_WG), nb_mat = blocks((size_t)m * P::NMAT, SPONGE_WG_SMALL);
      RUN("k_ml_enc_front", st, k_ml_enc_front<P><<<nb_pre + nb_mat, SPLIT_WG, 0, st>>>(m, (int)nb_pre, coins + o * 32, p, P::PKB, L.small, ss + o * 32, L.mat, L.tailq, L.tail_cap));
      if (L.tail_cap) RUN("k_sample_matrix_tail", st, k_sample_matrix_tail<<<blocks(L.tail_cap, 64), 64, 0, st>>>(L.tailq, L.tail_cap, L.mat));
    } else {
      RUN("k_ml_enc_pre", st, k_ml_enc_pre<P><<<blocks(m, sponge_wg(m)), sponge_wg(m), 0, st>>>(m, coins + o * 32, p, L.small, ss + o * 32));
      if (int rc = expand_matrix<P, false>(m, p + P::PVB, P::PKB, 1, L.mat, L, st)) return rc;
    }
    RUN("k_prf", st, k_prf<P><<<blocks((size_t)m * (2 * P::K + 1), sponge_wg((size_t)m * (2 * P::K + 1))), sponge_wg((size_t)m * (2 * P::K + 1)), 0, st>>>(m, 2 * P::K + 1, P::K, L.small, 64, L.prf));
    RUN("k_enc_arith", st, launch_enc_arith<P, false>(st, m, p, P::PKB, L.small, L.mat, (size_t)P::NMAT * 256, L.prf, c));
  }
  return 0;
}

template <class P>
int ml_dec_dev(size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* dk, hipStream_t st) {
  std::lock_guard<std::mutex> lk(dev_mutex(g_dev_mu));
  const size_t gc = g_chunk.load();
  const size_t chunk = n < gc ? n : gc;
  WsUse use(st, n, Layout<P>::need(chunk));
  if (use.rc) return use.rc;
  Layout<P> L;
  if (int rc = L.carve(chunk, use)) return rc;
  for (size_t o = 0; o < n; o += chunk) {
    const int m = (int)(n - o < chunk ? n - o : chunk);
    const uint8_t* c = ct + o * P::CTB;
    const uint8_t* s = dk + o * P::SKB;
    // small batches: G, the matrix expansion and the rejection key J(z || c) -- which waits in the caller's ss
    // buffer -- as one launch behind the decryption; behind the re-encryption only a select is left (keccak_split.h)
    const bool middle = (size_t)m <= FRONT_MAX_N;
    RUN("k_dec_arith", st, launch_dec_arith<P>(st, m, c, s, P::SKB, L.small));
    if (middle) {
      k_zero32<<<1, 64, 0, st>>>(L.tailq);
      LAUNCH_CHECK("k_zero32");
      const unsigned nb_mat = blocks((size_t)m * P::NMAT, SPONGE_WG_SMALL), nb_h = blocks(2 * (size_t)m, SPLIT_WG);
      RUN("k_ml_dec_middle", st, k_ml_dec_middle<P><<<nb_mat + 2 * nb_h, SPLIT_WG, 0, st>>>(m, (int)nb_mat, (int)nb_h, s, L.mat, L.tailq, L.tail_cap, c, ss + o * 32, L.small));
      if (L.tail_cap) RUN("k_sample_matrix_tail", st, k_sample_matrix_tail<<<blocks(L.tail_cap, 64), 64, 0, st>>>(L.tailq, L.tail_cap, L.mat));
    } else {
      RUN("k_dec_mid", st, k_dec_mid<P><<<blocks(m, sponge_wg(m)), sponge_wg(m), 0, st>>>(m, s, P::SKB, L.small));
      if (int rc = expand_matrix<P, false>(m, s + 2 * P::PVB, P::SKB, 1, L.mat, L, st)) return rc;
    }
    RUN("k_prf", st, k_prf<P><<<blocks((size_t)m * (2 * P::K + 1), sponge_wg((size_t)m * (2 * P::K + 1))), sponge_wg((size_t)m * (2 * P::K + 1)), 0, st>>>(m, 2 * P::K + 1, P::K, L.small, 64, L.prf));
    RUN("k_enc_arith_cmp", st,
        launch_enc_arith<P, true>(st, m, s + P::PVB, P::SKB, L.small, L.mat, (size_t)P::NMAT * 256, L.prf, const_cast<uint8_t*>(c)));
    if (middle) RUN("k_ml_select", st, k_ml_select<<<blocks(8 * (size_t)m, 256), 256, 0, st>>>(m, L.small, ss + o * 32));
    else RUN("k_ml_post_dec", st, k_ml_post_dec<P><<<blocks(m, sponge_wg(m)), sponge_wg(m), 0, st>>>(m, c, L.small, s, P::SKB, ss + o * 32));
  }
  return 0;
}

template <class P>
int ml_check_dev(size_t n, const uint8_t* key, int is_dk, uint32_t* flags, hipStream_t st) {
  if (is_dk) RUN("k_ml_check", st, k_ml_check<P, true><<<blocks(n, sponge_wg(n)), sponge_wg(n), 0, st>>>((int)n, key, flags));
  else RUN("k_ml_check", st, k_ml_check<P, false><<<blocks(n, sponge_wg(n)), sponge_wg(n), 0, st>>>((int)n, key, flags));
  return 0;
}

// ---- decapsulation that allocates nothing (include/kyber_batch.h: kyber_batch_dec_dev_noalloc) ----
// The workspace of kyber_batch_dec_dev (expanded matrices, PRF bytes, the
// 128-byte records, the sampler's tail queue) is carved out of this file-scope
// device array instead of a hipMalloc'ed block.  Every GPU the library runs
// on holds its own copy until the process ends.  In host code the array's name
// is only a handle: its device address is asked for once per device with
// hipGetSymbolAddress and only that pointer is ever handed to kernels.
constexpr size_t NOALLOC_MAX_N = KYBER_HIP_NOALLOC_MAX_N;
constexpr size_t max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
constexpr size_t NOALLOC_SCRATCH_BYTES =
    max3(Layout<Par<2>>::need(NOALLOC_MAX_N), Layout<Par<3>>::need(NOALLOC_MAX_N), Layout<Par<4>>::need(NOALLOC_MAX_N));
static __device__ __align__(256) uint8_t scratch[NOALLOC_SCRATCH_BYTES];
uint8_t* g_scratch_ptr[MAX_DEV] = {};  // device address of `scratch` on each device, guarded by g_dev_mu[dev]
// Calls outside stream capture are ordered on the device like the workspace calls (WsUse): each
// first waits for the event recorded behind the previous one.  Under capture nothing is waited for
// or recorded (an event of the eager world means nothing inside a graph): captured calls and
// replays that overlap on one GPU race on the array, as the header says.
struct ScratchOrder {
  hipEvent_t idle = nullptr;
  hipStream_t last = nullptr;
  bool used = false;
};
ScratchOrder g_scratch_order[MAX_DEV];

template <class P, bool N90>
int dec_noalloc(size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* sk, hipStream_t st) {
  static_assert(Layout<P>::need(NOALLOC_MAX_N) <= NOALLOC_SCRATCH_BYTES, "scratch too small");
  if (n > NOALLOC_MAX_N) {
    snprintf(g_err, sizeof g_err, "kyber_batch_dec_dev_noalloc: n = %zu exceeds KYBER_HIP_NOALLOC_MAX_N = %zu", n, NOALLOC_MAX_N);
    return KYBER_HIP_EINVAL;
  }
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEV) return KYBER_HIP_EINVAL;
  std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
  ScratchOrder& ord = g_scratch_order[dev];
  if (!g_scratch_ptr[dev]) {  // set-up, once per device: the array's address and the ordering event
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));  // this device's copy (dev is current)
    if (!p) return KYBER_HIP_EHIP;
    if (!ord.idle) HIP_TRY(hipEventCreateWithFlags(&ord.idle, hipEventDisableTiming));
    (void)device_cus();  // cached now, so that the AES launches ask the runtime nothing later
    g_scratch_ptr[dev] = static_cast<uint8_t*>(p);
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  HIP_TRY(hipStreamIsCapturing(st, &cap));
  const bool eager = cap == hipStreamCaptureStatusNone;
  if (eager && ord.used && ord.last != st) HIP_TRY(hipStreamWaitEvent(st, ord.idle, 0));
  Layout<P> L;
  L.place(g_scratch_ptr[dev], n);
  // what must start at zero (the tail queue's counter) is cleared by sample_matrix
  // on `st`; every other word of the workspace is written before it is read
  const int rc = dec_run<P, N90>(n, n, L, ss, ct, sk, P::SKB, st);
  if (eager) {
    if (hipEventRecord(ord.idle, st) == hipSuccess) {
      ord.last = st;
      ord.used = true;
    } else {
      (void)hipStreamSynchronize(st);  // cannot order the next call on the device: order it on the host
      ord.used = false;
    }
  }
  return rc;
}

template <class P, bool N90>
int gen_matrix_dev(size_t n, int16_t* mat, const uint8_t* seeds, int transposed, hipStream_t st) {
  if (n * P::NMAT > 0x7fffffffu) return KYBER_HIP_EINVAL;
  if constexpr (N90) {
    AES_LAUNCH("k90_sample_matrix", st, k90_sample_matrix, n * P::NMAT, ((int)(n * P::NMAT), seeds, 32, transposed, mat));
  } else {
    // no workspace here: a null queue keeps every entry in the main kernel until it is complete
    const size_t entries = n * P::NMAT;
    if (sponge_wg(entries) == SPONGE_WG)
      RUN("k_sample_matrix", st,
          (k_sample_matrix<P, SPONGE_WG><<<blocks(entries, SPONGE_WG), SPONGE_WG, 0, st>>>((int)entries, seeds, 32, transposed, mat, nullptr, 0u)));
    else
      RUN("k_sample_matrix", st,
          (k_sample_matrix<P, SPONGE_WG_SMALL><<<blocks(entries, SPONGE_WG_SMALL), SPONGE_WG_SMALL, 0, st>>>((int)entries, seeds, 32, transposed, mat, nullptr, 0u)));
  }
  return 0;
}

// a device allocation owned by a scope
struct DevBuf {
  uint8_t* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), bytes ? bytes : 16));
    return 0;
  }
};

// ---- prepared keys (include/kyber_batch.h: kyber_hip_prepare_pk / _sk) ------------------------
// What a shared-key call derives from its one key before it touches an instance -- the expanded
// matrix A^T (ref/indcpa.c:270) and H(pk) (ref/kem.c:64) -- computed once and kept on the device
// next to a copy of the key.  Immutable once kyber_hip_prepare_* has returned (it synchronises),
// so any number of calls on any stream of that device may use one handle.
constexpr uint32_t PREP_MAGIC = 0x4B595052u;  // "KYPR"
}  // namespace
struct kyber_hip_prepared {
  uint32_t magic;
  int k, n90, dev, is_sk;
  uint8_t* base;        // one allocation: key bytes | matrix | H(pk)
  size_t bytes;
  const uint8_t* key;   // pk (PKB) or sk (SKB)
  const int16_t* mat;   // A^T, K*K polynomials
  const uint8_t* hpk;   // 32 bytes (pk handles)
};
namespace {
template <class P, bool N90>
int prepare_key(const uint8_t* key, int is_sk, int key_on_device, kyber_hip_prepared** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  const size_t kb = is_sk ? P::SKB : P::PKB, koff = (kb + 255) & ~(size_t)255, moff = koff + (size_t)P::NMAT * 512;
  DevBuf buf;
  if (int rc = buf.alloc(moff + 256)) return rc;
  HIP_TRY(hipMemcpy(buf.p, key, kb, key_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  hipStream_t st = nullptr;
  const uint8_t* pk = buf.p + (is_sk ? P::PVB : 0);  // the public key inside a secret key, ref/kem.c:28-29
  // no workspace here: a null queue keeps every entry in the main kernel until it is complete
  if (int rc = gen_matrix_dev<P, N90>(1, reinterpret_cast<int16_t*>(buf.p + koff), pk + P::PVB, 1, st)) return rc;
  if (is_sk) {
    HIP_TRY(hipMemcpyAsync(buf.p + moff, buf.p + 2 * P::PVB + 32, 32, hipMemcpyDeviceToDevice, st));  // sk carries H(pk), ref/kem.c:30
  } else {
    if constexpr (N90) RUN("k90_hash_pk", st, k90_hash_pk<P><<<1, 64, 0, st>>>(pk, buf.p + moff));
    else RUN("k_hash_pk", st, k_hash_pk<P><<<1, 64, 0, st>>>(pk, buf.p + moff));
  }
  HIP_TRY(hipStreamSynchronize(st));
  kyber_hip_prepared* h = new kyber_hip_prepared{PREP_MAGIC, P::K, N90 ? 1 : 0, dev, is_sk, buf.p, moff + 256,
                                                 buf.p, reinterpret_cast<const int16_t*>(buf.p + koff), buf.p + moff};
  buf.p = nullptr;  // owned by the handle now
  *out = h;
  return 0;
}

// a handle is used on the device it was prepared on, for the operation its key allows
int prepared_check(const kyber_hip_prepared* h, int want_sk) {
  if (!h || h->magic != PREP_MAGIC || h->is_sk != want_sk) return KYBER_HIP_EINVAL;
  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  if (dev != h->dev) {
    snprintf(g_err, sizeof g_err, "prepared key lives on device %d, current device is %d", h->dev, dev);
    return KYBER_HIP_EINVAL;
  }
  return 0;
}
template <class P, bool N90>
int enc_prepared_dev(const kyber_hip_prepared* h, size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* coins, hipStream_t st) {
  return enc_dev<P, N90>(n, ct, ss, h->key, 0, coins, st, h->mat, h->hpk);
}
template <class P, bool N90>
int dec_prepared_dev(const kyber_hip_prepared* h, size_t n, uint8_t* ss, const uint8_t* ct, hipStream_t st) {
  return dec_dev<P, N90>(n, ss, ct, h->key, 0, st, h->mat);
}

// ---- key exchange (ref/kex.c:6-54) as batched compositions ----------------------
// The protocol messages interleave a key and a ciphertext per instance
// (ref/kex.h:7-12) while the KEM pipelines work on dense arrays, so messages
// are split/joined with strided device copies into stream-ordered temporaries.
struct AsyncBuf {
  uint8_t* p = nullptr;
  size_t bytes = 0;
  hipStream_t st = nullptr;
  ~AsyncBuf() {
    if (!p) return;
    (void)hipMemsetAsync(p, 0, bytes, st);  // shared secrets and ephemeral coins pass through these temporaries
    (void)hipFreeAsync(p, st);
  }
  int alloc(size_t n, hipStream_t s) {
    st = s;
    bytes = n ? n : 16;
    HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&p), bytes, s));
    return 0;
  }
};

// dst[i][0..w) = src[i][0..w) for n rows with the given row pitches
inline int copy_rows(uint8_t* dst, size_t dpitch, const uint8_t* src, size_t spitch, size_t w, size_t n, hipStream_t st) {
  HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, w, n, hipMemcpyDeviceToDevice, st));
  return 0;
}

template <class P, bool N90>
int hash_h_dev(size_t n, int of_ct, uint8_t* out, const uint8_t* in, hipStream_t st) {
  if (n > 0x7fffffffu) return KYBER_HIP_EINVAL;
  if (of_ct) RUN("k_st_hash_h", st, (k_st_hash_h<P, N90, true><<<blocks(n, sponge_wg(n)), sponge_wg(n), 0, st>>>((int)n, in, out)));
  else RUN("k_st_hash_h", st, (k_st_hash_h<P, N90, false><<<blocks(n, sponge_wg(n)), sponge_wg(n), 0, st>>>((int)n, in, out)));
  return 0;
}

// Noise polynomials for nonces 0..nnonce-1 of every seed: the PRF kernel of the KEM pipelines
// followed by the centred binomial map (ref/poly.c:208-232), r[i][nonce][256].
template <class P, bool N90>
int getnoise_dev(size_t n, int nnonce, int n_eta1, int16_t* r, const uint8_t* seeds, hipStream_t st) {
  if (nnonce < 1 || nnonce > 256 || n_eta1 < 0 || n * (size_t)nnonce > 0x7fffffffu) return KYBER_HIP_EINVAL;
  const size_t jobs = n * (size_t)nnonce;
  AsyncBuf small, prf;
  if (int rc = small.alloc(n * SMALL, st)) return rc;
  if (int rc = prf.alloc(jobs * P::PRF_STRIDE, st)) return rc;
  if (int rc = copy_rows(small.p, SMALL, seeds, 32, 32, n, st)) return rc;
  if constexpr (N90)
    AES_LAUNCH("k90_prf", st, k90_prf, jobs, ((int)n, nnonce, n_eta1, small.p, 0, prf.p));
  else
    RUN("k_prf", st, k_prf<P><<<blocks(jobs, sponge_wg(jobs)), sponge_wg(jobs), 0, st>>>((int)n, nnonce, n_eta1, small.p, 0, prf.p));
  RUN("k_st_noise", st, launch_st_noise<P>(st, jobs, nnonce, n_eta1, r, prf.p));
  return 0;
}

template <int NP, bool N90>
int kdf_parts(size_t n, const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, uint8_t* out, hipStream_t st) {
  RUN("k_kdf_parts", st, k_kdf_parts<NP, N90><<<blocks(n, sponge_wg(n)), sponge_wg(n), 0, st>>>((int)n, p0, p1, p2, out));
  return 0;
}

#define KX(call)              \
  do {                        \
    if (int rc_ = (call)) return rc_; \
  } while (0)

// kex_uake_initA == kex_ake_initA (ref/kex.c:6-10, :30-34): ephemeral keypair, encapsulate to B's static key
template <class P, bool N90>
int kex_initA_dev(size_t n, uint8_t* send, uint8_t* tk, uint8_t* sk, const uint8_t* pkb, const uint8_t* coins, int shared_b,
                  hipStream_t st) {
  AsyncBuf pk, ct, kc, ec;
  KX(pk.alloc(n * P::PKB, st)); KX(ct.alloc(n * P::CTB, st)); KX(kc.alloc(n * 64, st)); KX(ec.alloc(n * 32, st));
  KX(copy_rows(kc.p, 64, coins, 96, 64, n, st));
  KX(copy_rows(ec.p, 32, coins + 64, 96, 32, n, st));
  KX((keypair_dev<P, N90>(n, pk.p, sk, kc.p, st)));
  KX((enc_dev<P, N90>(n, ct.p, tk, pkb, shared_b ? 0 : P::PKB, ec.p, st)));
  KX(copy_rows(send, P::PKB + P::CTB, pk.p, P::PKB, P::PKB, n, st));
  KX(copy_rows(send + P::PKB, P::PKB + P::CTB, ct.p, P::CTB, P::CTB, n, st));
  return 0;
}

// ref/kex.c:12-18
template <class P, bool N90>
int kex_uake_sharedB_dev(size_t n, uint8_t* send, uint8_t* key, const uint8_t* recv, const uint8_t* skb, const uint8_t* coins,
                         int shared_b, hipStream_t st) {
  AsyncBuf pk, ct, s0, s1;
  KX(pk.alloc(n * P::PKB, st)); KX(ct.alloc(n * P::CTB, st)); KX(s0.alloc(n * 32, st)); KX(s1.alloc(n * 32, st));
  KX(copy_rows(pk.p, P::PKB, recv, P::PKB + P::CTB, P::PKB, n, st));
  KX(copy_rows(ct.p, P::CTB, recv + P::PKB, P::PKB + P::CTB, P::CTB, n, st));
  KX((enc_dev<P, N90>(n, send, s0.p, pk.p, P::PKB, coins, st)));
  KX((dec_dev<P, N90>(n, s1.p, ct.p, skb, shared_b ? 0 : P::SKB, st)));
  return kdf_parts<2, N90>(n, s0.p, s1.p, nullptr, key, st);
}

// ref/kex.c:20-27
template <class P, bool N90>
int kex_uake_sharedA_dev(size_t n, uint8_t* key, const uint8_t* recv, const uint8_t* tk, const uint8_t* sk, hipStream_t st) {
  AsyncBuf s0;
  KX(s0.alloc(n * 32, st));
  KX((dec_dev<P, N90>(n, s0.p, recv, sk, P::SKB, st)));
  return kdf_parts<2, N90>(n, s0.p, tk, nullptr, key, st);
}

// ref/kex.c:36-43
template <class P, bool N90>
int kex_ake_sharedB_dev(size_t n, uint8_t* send, uint8_t* key, const uint8_t* recv, const uint8_t* skb, const uint8_t* pka,
                        const uint8_t* coins, int shared_b, hipStream_t st) {
  AsyncBuf pk, ct, ct0, ct1, s0, s1, s2, c0, c1;
  KX(pk.alloc(n * P::PKB, st)); KX(ct.alloc(n * P::CTB, st)); KX(ct0.alloc(n * P::CTB, st)); KX(ct1.alloc(n * P::CTB, st));
  KX(s0.alloc(n * 32, st)); KX(s1.alloc(n * 32, st)); KX(s2.alloc(n * 32, st)); KX(c0.alloc(n * 32, st)); KX(c1.alloc(n * 32, st));
  KX(copy_rows(pk.p, P::PKB, recv, P::PKB + P::CTB, P::PKB, n, st));
  KX(copy_rows(ct.p, P::CTB, recv + P::PKB, P::PKB + P::CTB, P::CTB, n, st));
  KX(copy_rows(c0.p, 32, coins, 64, 32, n, st));
  KX(copy_rows(c1.p, 32, coins + 32, 64, 32, n, st));
  KX((enc_dev<P, N90>(n, ct0.p, s0.p, pk.p, P::PKB, c0.p, st)));
  KX((enc_dev<P, N90>(n, ct1.p, s1.p, pka, P::PKB, c1.p, st)));
  KX((dec_dev<P, N90>(n, s2.p, ct.p, skb, shared_b ? 0 : P::SKB, st)));
  KX(copy_rows(send, 2 * P::CTB, ct0.p, P::CTB, P::CTB, n, st));
  KX(copy_rows(send + P::CTB, 2 * P::CTB, ct1.p, P::CTB, P::CTB, n, st));
  return kdf_parts<3, N90>(n, s0.p, s1.p, s2.p, key, st);
}

// ref/kex.c:45-54
template <class P, bool N90>
int kex_ake_sharedA_dev(size_t n, uint8_t* key, const uint8_t* recv, const uint8_t* tk, const uint8_t* sk, const uint8_t* ska,
                        hipStream_t st) {
  AsyncBuf ct0, ct1, s0, s1;
  KX(ct0.alloc(n * P::CTB, st)); KX(ct1.alloc(n * P::CTB, st)); KX(s0.alloc(n * 32, st)); KX(s1.alloc(n * 32, st));
  KX(copy_rows(ct0.p, P::CTB, recv, 2 * P::CTB, P::CTB, n, st));
  KX(copy_rows(ct1.p, P::CTB, recv + P::CTB, 2 * P::CTB, P::CTB, n, st));
  KX((dec_dev<P, N90>(n, s0.p, ct0.p, sk, P::SKB, st)));
  KX((dec_dev<P, N90>(n, s1.p, ct1.p, ska, P::SKB, st)));
  return kdf_parts<3, N90>(n, s0.p, s1.p, tk, key, st);
}

// ---- host-pointer pipelines ------------------------------------------------------
// A host batch is cut into passes of HOST_PASS instances that flow through three
// streams -- copy in, compute, copy out -- with two device staging slots, so the
// PCIe transfers of neighbouring passes overlap the kernels of the current one.
// With pinned host memory (kyber_hip_host_alloc, hipHostMalloc, hipHostRegister)
// the copies are true DMA; with pageable memory the runtime stages them and the
// calls are merely correct.
constexpr size_t HOST_PASS = 1u << 15;

struct HostIn {
  const uint8_t* p;
  size_t per;   // bytes per instance
  bool once;    // one record for the whole batch (shared key), copied before the first pass
};
struct HostOut {
  uint8_t* p;
  size_t per;
};

struct Streams {
  hipStream_t in = nullptr, comp = nullptr, out = nullptr;
  hipEvent_t in_done[2] = {}, comp_done[2] = {}, out_done[2] = {};
  bool ok = false;
  int init() {
    HIP_TRY(hipStreamCreateWithFlags(&in, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&comp, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&out, hipStreamNonBlocking));
    for (int s = 0; s < 2; s++) {
      HIP_TRY(hipEventCreateWithFlags(&in_done[s], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&comp_done[s], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&out_done[s], hipEventDisableTiming));
    }
    ok = true;
    return 0;
  }
  ~Streams() {
    if (in) (void)hipStreamDestroy(in);
    if (comp) (void)hipStreamDestroy(comp);
    if (out) (void)hipStreamDestroy(out);
    for (int s = 0; s < 2; s++) {
      if (in_done[s]) (void)hipEventDestroy(in_done[s]);
      if (comp_done[s]) (void)hipEventDestroy(comp_done[s]);
      if (out_done[s]) (void)hipEventDestroy(out_done[s]);
    }
  }
};

// Streams, events and the staging allocation are kept per device between calls
// (small batches would otherwise spend most of their time creating them).  One context per
// device by default, whose lock serialises the host calls on that device; with
// kyber_hip_set_workspace_slots(s > 1) there are s, and a host call of up to POOL_MAX_N instances
// takes one that is free (HostLock), so calls from several threads overlap: each context has its own
// streams, and a stream keeps to its workspace (pick_slot).
struct HostCtx {
  Streams S;
  uint8_t* staging = nullptr;
  size_t cap = 0;
  uint8_t* mirror = nullptr;  // pinned host copy of the staging area's layout, for small batches (host_pipeline)
};
constexpr size_t HOST_MIRROR_BYTES = 1u << 20;  // Kyber768: up to n = 256 (the CPU moves these bytes, as the runtime would for pageable memory)
HostCtx* g_host[MAX_DEV][MAX_SLOTS] = {};
std::mutex g_host_mu[MAX_DEV][MAX_SLOTS];
std::atomic<unsigned> g_host_rr[MAX_DEV];

// holds one of the current device's host contexts for the length of a host-pointer call
struct HostLock {
  int dev = 0, idx = 0;
  std::unique_lock<std::mutex> lk;
  explicit HostLock(size_t n) {
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) dev = 0;
    const int nslot = g_slots.load();
    if (nslot > 1 && n <= POOL_MAX_N) {
      for (int i = 0; i < nslot; i++) {
        std::unique_lock<std::mutex> t(g_host_mu[dev][i], std::try_to_lock);
        if (t.owns_lock()) {
          idx = i;
          lk = std::move(t);
          return;
        }
      }
      idx = (int)(g_host_rr[dev]++ % (unsigned)nslot);  // all busy: queue behind one of them
    }
    lk = std::unique_lock<std::mutex>(g_host_mu[dev][idx]);
  }
};

int host_ctx(const HostLock& hl, size_t bytes, HostCtx** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEV) return KYBER_HIP_EINVAL;
  if (dev != hl.dev) return KYBER_HIP_EINVAL;
  if (!g_host[dev][hl.idx]) {
    HostCtx* c = new HostCtx();
    if (int rc = c->S.init()) {
      delete c;
      return rc;
    }
    g_host[dev][hl.idx] = c;
  }
  HostCtx* c = g_host[dev][hl.idx];
  if (c->cap < bytes) {
    if (c->staging) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(c->staging));
      c->staging = nullptr;
      c->cap = 0;
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->staging), bytes));
    c->cap = bytes;
  }
  *out = c;
  return 0;
}

// f(m, in[], out[], stream): the device-pointer routine for one pass of m instances
template <class F>
int host_pipeline(size_t n, std::initializer_list<HostIn> ins_, std::initializer_list<HostOut> outs_, F&& f) {
  HostLock hlk(n);
  const std::vector<HostIn> ins(ins_);
  const std::vector<HostOut> outs(outs_);
  const size_t pass = n < HOST_PASS ? n : HOST_PASS;
  const int slots = n > pass ? 2 : 1;
  // one allocation holds every staging buffer of every slot (256-byte aligned pieces)
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t total = 0;
  for (auto& a : ins) total += a.once ? pad(a.per) : slots * pad(a.per * pass);
  for (auto& a : outs) total += slots * pad(a.per * pass);
  HostCtx* ctx = nullptr;
  if (int rc = host_ctx(hlk, total, &ctx)) return rc;
  Streams& S = ctx->S;
  // A batch that fits one pass has nothing to overlap: copies and kernels go down ONE stream in order,
  // which saves the two cross-stream event hops and two stream synchronisations of the general path
  // (tens of microseconds of a 0.2 ms call, DESIGN 5 "small batches").
  const bool single = slots == 1;
  const hipStream_t s_in = single ? S.comp : S.in, s_out = single ? S.comp : S.out;
  // Every early return below leaves copies into or out of caller memory (and out
  // of Coins' buffer) in flight: drain the three streams before the caller may
  // free anything.  Disarmed after the final synchronisation of the normal path.
  struct Drain {
    Streams& S;
    bool armed = true;
    ~Drain() {
      if (!armed) return;
      (void)hipStreamSynchronize(S.in);
      (void)hipStreamSynchronize(S.comp);
      (void)hipStreamSynchronize(S.out);
    }
  } drain{S};
  std::vector<uint8_t*> din[2], dout[2];
  uint8_t* cur = ctx->staging;
  for (int s = 0; s < 2; s++) {
    din[s].resize(ins.size());
    dout[s].resize(outs.size());
  }
  // Small single-pass batches go through a pinned mirror of the staging area: inputs are gathered into
  // it by the CPU and reach the device as ONE copy, outputs come back as ONE copy and are scattered by
  // the CPU -- two DMA commands and one wait per call instead of a command per buffer, each of which,
  // from or to pageable memory, is staged by the runtime and (device to host) waited for on its own.
  if (single && total <= HOST_MIRROR_BYTES) {
    if (!ctx->mirror) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->mirror), HOST_MIRROR_BYTES, hipHostMallocDefault));
    uint8_t* mir = ctx->mirror;
    size_t off = 0;
    for (size_t i = 0; i < ins.size(); i++) {
      const size_t b = ins[i].once ? ins[i].per : ins[i].per * n;
      din[0][i] = cur + off;
      std::memcpy(mir + off, ins[i].p, b);
      off += pad(b);
    }
    const size_t in_bytes = off;
    for (size_t i = 0; i < outs.size(); i++) {
      dout[0][i] = cur + off;
      off += pad(outs[i].per * n);
    }
    struct WipeMirror {  // coins, secret keys and shared secrets passed through it
      uint8_t* p;
      size_t b;
      hipStream_t st;
      bool idle = false;  // set once the stream has been waited for: nothing can land in the mirror after the wipe
      ~WipeMirror() {
        if (!idle) (void)hipStreamSynchronize(st);
        explicit_bzero(p, b);
      }
    } wipe{mir, off, S.comp};
    if (in_bytes) HIP_TRY(hipMemcpyAsync(cur, mir, in_bytes, hipMemcpyHostToDevice, S.comp));
    if (int rc = f(n, din[0].data(), dout[0].data(), S.comp)) return rc;
    if (off > in_bytes) HIP_TRY(hipMemcpyAsync(mir + in_bytes, cur + in_bytes, off - in_bytes, hipMemcpyDeviceToHost, S.comp));
    HIP_TRY(hipStreamSynchronize(S.comp));
    wipe.idle = true;
    drain.armed = false;
    for (size_t i = 0; i < outs.size(); i++) std::memcpy(outs[i].p, mir + (dout[0][i] - cur), outs[i].per * n);
    return 0;
  }
  for (size_t i = 0; i < ins.size(); i++) {
    if (ins[i].once) {
      din[0][i] = din[1][i] = cur;
      cur += pad(ins[i].per);
      HIP_TRY(hipMemcpyAsync(din[0][i], ins[i].p, ins[i].per, hipMemcpyHostToDevice, s_in));
    } else {
      for (int s = 0; s < slots; s++) {
        din[s][i] = cur;
        cur += pad(ins[i].per * pass);
      }
    }
  }
  for (size_t i = 0; i < outs.size(); i++)
    for (int s = 0; s < slots; s++) {
      dout[s][i] = cur;
      cur += pad(outs[i].per * pass);
    }
  size_t c = 0;
  for (size_t o = 0; o < n; o += pass, c++) {
    const size_t m = n - o < pass ? n - o : pass;
    const int s = (int)(c & 1) % slots;
    if (c >= 2) HIP_TRY(hipStreamWaitEvent(S.in, S.comp_done[s], 0));  // slot's inputs consumed
    for (size_t i = 0; i < ins.size(); i++)
      if (!ins[i].once)
        HIP_TRY(hipMemcpyAsync(din[s][i], ins[i].p + o * ins[i].per, m * ins[i].per, hipMemcpyHostToDevice, s_in));
    if (!single) {
      HIP_TRY(hipEventRecord(S.in_done[s], S.in));
      HIP_TRY(hipStreamWaitEvent(S.comp, S.in_done[s], 0));
    }
    if (c >= 2) HIP_TRY(hipStreamWaitEvent(S.comp, S.out_done[s], 0));  // slot's outputs drained
    if (int rc = f(m, din[s].data(), dout[s].data(), S.comp)) return rc;
    if (!single) {
      HIP_TRY(hipEventRecord(S.comp_done[s], S.comp));
      HIP_TRY(hipStreamWaitEvent(S.out, S.comp_done[s], 0));
    }
    for (size_t i = 0; i < outs.size(); i++)
      HIP_TRY(hipMemcpyAsync(outs[i].p + o * outs[i].per, dout[s][i], m * outs[i].per, hipMemcpyDeviceToHost, s_out));
    if (!single) HIP_TRY(hipEventRecord(S.out_done[s], S.out));
  }
  if (!single) HIP_TRY(hipStreamSynchronize(S.out));
  HIP_TRY(hipStreamSynchronize(S.comp));
  if (!single) HIP_TRY(hipStreamSynchronize(S.in));
  drain.armed = false;
  return 0;
}

// coins == NULL: draw them from randombytes (zeroised afterwards)
struct Coins {
  std::vector<uint8_t> own;
  const uint8_t* p;
  Coins(const uint8_t* given, size_t bytes) : p(given) {
    if (!given) {
      own.resize(bytes ? bytes : 1);
      randombytes(own.data(), bytes);
      p = own.data();
    }
  }
  ~Coins() {
    if (!own.empty()) explicit_bzero(own.data(), own.size());  // not a dead store the compiler may drop
  }
};

template <class P, bool N90>
int keypair_host(size_t n, uint8_t* pk, uint8_t* sk, const uint8_t* coins) {
  Coins c(coins, n * 64);
  return host_pipeline(n, {{c.p, 64, false}}, {{pk, P::PKB}, {sk, P::SKB}}, [](size_t m, uint8_t** in, uint8_t** out, hipStream_t st) {
    return keypair_dev<P, N90>(m, out[0], out[1], in[0], st);
  });
}

template <class P, bool N90>
int enc_each_dev(size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* pk, const uint8_t* coins, hipStream_t st) {
  return enc_dev<P, N90>(n, ct, ss, pk, P::PKB, coins, st);
}
template <class P, bool N90>
int enc_shared_dev(size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* pk, const uint8_t* coins, hipStream_t st) {
  return enc_dev<P, N90>(n, ct, ss, pk, 0, coins, st);
}
template <class P, bool N90>
int dec_each_dev(size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* sk, hipStream_t st) {
  return dec_dev<P, N90>(n, ss, ct, sk, P::SKB, st);
}
template <class P, bool N90>
int dec_shared_dev(size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* sk, hipStream_t st) {
  return dec_dev<P, N90>(n, ss, ct, sk, 0, st);
}

template <class P, bool N90>
int enc_host(size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* pk, const uint8_t* coins, bool shared) {
  Coins c(coins, n * 32);
  return host_pipeline(n, {{pk, P::PKB, shared}, {c.p, 32, false}}, {{ct, P::CTB}, {ss, 32}},
                       [shared](size_t m, uint8_t** in, uint8_t** out, hipStream_t st) {
                         return enc_dev<P, N90>(m, out[0], out[1], in[0], shared ? 0 : P::PKB, in[1], st);
                       });
}

template <class P, bool N90>
int dec_host(size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* sk, bool shared) {
  return host_pipeline(n, {{ct, P::CTB, false}, {sk, P::SKB, shared}}, {{ss, 32}},
                       [shared](size_t m, uint8_t** in, uint8_t** out, hipStream_t st) {
                         return dec_dev<P, N90>(m, out[0], in[0], in[1], shared ? 0 : P::SKB, st);
                       });
}

template <class P, bool N90>
int enc_prepared_host(const kyber_hip_prepared* h, size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* coins) {
  Coins c(coins, n * 32);
  return host_pipeline(n, {{c.p, 32, false}}, {{ct, P::CTB}, {ss, 32}}, [h](size_t m, uint8_t** in, uint8_t** out, hipStream_t st) {
    return enc_prepared_dev<P, N90>(h, m, out[0], out[1], in[0], st);
  });
}
template <class P, bool N90>
int dec_prepared_host(const kyber_hip_prepared* h, size_t n, uint8_t* ss, const uint8_t* ct) {
  return host_pipeline(n, {{ct, P::CTB, false}}, {{ss, 32}}, [h](size_t m, uint8_t** in, uint8_t** out, hipStream_t st) {
    return dec_prepared_dev<P, N90>(h, m, out[0], in[0], st);
  });
}

// Host-pointer form of a device-pointer routine: stage every argument in device
// memory, run `f` on the null stream, copy the outputs back.
struct HostArg {
  const void* in;  // copied to the device before the call (may be null)
  void* out;       // copied back after the call (may be null)
  size_t bytes;
};
template <class F>
int with_device(std::initializer_list<HostArg> args, F&& f) {
  std::vector<DevBuf> bufs(args.size());
  std::vector<uint8_t*> ptr(args.size());
  size_t i = 0;
  for (const HostArg& a : args) {
    if (int rc = bufs[i].alloc(a.bytes)) return rc;
    if (a.in) HIP_TRY(hipMemcpy(bufs[i].p, a.in, a.bytes, hipMemcpyHostToDevice));
    ptr[i] = bufs[i].p;
    i++;
  }
  if (int rc = f(ptr.data())) return rc;
  i = 0;
  for (const HostArg& a : args) {
    if (a.out) HIP_TRY(hipMemcpy(a.out, bufs[i].p, a.bytes, hipMemcpyDeviceToHost));
    i++;
  }
  return 0;
}

template <class P, bool N90>
int kex_initA_host(size_t n, uint8_t* send, uint8_t* tk, uint8_t* sk, const uint8_t* pkb, const uint8_t* coins, int shared_b) {
  Coins c(coins, n * 96);
  return with_device({{nullptr, send, n * (P::PKB + P::CTB)}, {nullptr, tk, n * 32}, {nullptr, sk, n * P::SKB},
                      {pkb, nullptr, (shared_b ? 1 : n) * P::PKB}, {c.p, nullptr, n * 96}},
                     [&](uint8_t** d) { return kex_initA_dev<P, N90>(n, d[0], d[1], d[2], d[3], d[4], shared_b, nullptr); });
}
template <class P, bool N90>
int kex_uake_sharedB_host(size_t n, uint8_t* send, uint8_t* key, const uint8_t* recv, const uint8_t* skb, const uint8_t* coins,
                          int shared_b) {
  Coins c(coins, n * 32);
  return with_device({{nullptr, send, n * P::CTB}, {nullptr, key, n * 32}, {recv, nullptr, n * (P::PKB + P::CTB)},
                      {skb, nullptr, (shared_b ? 1 : n) * P::SKB}, {c.p, nullptr, n * 32}},
                     [&](uint8_t** d) { return kex_uake_sharedB_dev<P, N90>(n, d[0], d[1], d[2], d[3], d[4], shared_b, nullptr); });
}
template <class P, bool N90>
int kex_uake_sharedA_host(size_t n, uint8_t* key, const uint8_t* recv, const uint8_t* tk, const uint8_t* sk) {
  return with_device({{nullptr, key, n * 32}, {recv, nullptr, n * P::CTB}, {tk, nullptr, n * 32}, {sk, nullptr, n * P::SKB}},
                     [&](uint8_t** d) { return kex_uake_sharedA_dev<P, N90>(n, d[0], d[1], d[2], d[3], nullptr); });
}
template <class P, bool N90>
int kex_ake_sharedB_host(size_t n, uint8_t* send, uint8_t* key, const uint8_t* recv, const uint8_t* skb, const uint8_t* pka,
                         const uint8_t* coins, int shared_b) {
  Coins c(coins, n * 64);
  return with_device({{nullptr, send, n * 2 * P::CTB}, {nullptr, key, n * 32}, {recv, nullptr, n * (P::PKB + P::CTB)},
                      {skb, nullptr, (shared_b ? 1 : n) * P::SKB}, {pka, nullptr, n * P::PKB}, {c.p, nullptr, n * 64}},
                     [&](uint8_t** d) { return kex_ake_sharedB_dev<P, N90>(n, d[0], d[1], d[2], d[3], d[4], d[5], shared_b, nullptr); });
}
template <class P, bool N90>
int kex_ake_sharedA_host(size_t n, uint8_t* key, const uint8_t* recv, const uint8_t* tk, const uint8_t* sk, const uint8_t* ska) {
  return with_device({{nullptr, key, n * 32}, {recv, nullptr, n * 2 * P::CTB}, {tk, nullptr, n * 32}, {sk, nullptr, n * P::SKB},
                      {ska, nullptr, n * P::SKB}},
                     [&](uint8_t** d) { return kex_ake_sharedA_dev<P, N90>(n, d[0], d[1], d[2], d[3], d[4], nullptr); });
}

template <class P>
int ml_keypair_host(size_t n, uint8_t* ek, uint8_t* dk, const uint8_t* coins) {
  Coins c(coins, n * 64);
  return host_pipeline(n, {{c.p, 64, false}}, {{ek, P::PKB}, {dk, P::SKB}}, [](size_t m, uint8_t** in, uint8_t** out, hipStream_t st) {
    return ml_keypair_dev<P>(m, out[0], out[1], in[0], st);
  });
}
template <class P>
int ml_enc_host(size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* ek, const uint8_t* coins) {
  Coins c(coins, n * 32);
  return host_pipeline(n, {{ek, P::PKB, false}, {c.p, 32, false}}, {{ct, P::CTB}, {ss, 32}},
                       [](size_t m, uint8_t** in, uint8_t** out, hipStream_t st) { return ml_enc_dev<P>(m, out[0], out[1], in[0], in[1], st); });
}
template <class P>
int ml_dec_host(size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* dk) {
  return host_pipeline(n, {{ct, P::CTB, false}, {dk, P::SKB, false}}, {{ss, 32}},
                       [](size_t m, uint8_t** in, uint8_t** out, hipStream_t st) { return ml_dec_dev<P>(m, out[0], in[0], in[1], st); });
}
template <class P>
int ml_check_host(size_t n, const uint8_t* key, int is_dk, uint32_t* flags) {
  const size_t per = is_dk ? P::SKB : P::PKB;
  return host_pipeline(n, {{key, per, false}}, {{reinterpret_cast<uint8_t*>(flags), 4}},
                       [is_dk](size_t m, uint8_t** in, uint8_t** out, hipStream_t st) {
                         return ml_check_dev<P>(m, in[0], is_dk, reinterpret_cast<uint32_t*>(out[0]), st);
                       });
}

#define DISPATCH_ML(K, FN, ...)                          \
  do {                                                   \
    switch (K) {                                         \
      case 2: return FN<Par<2>>(__VA_ARGS__);            \
      case 3: return FN<Par<3>>(__VA_ARGS__);            \
      case 4: return FN<Par<4>>(__VA_ARGS__);            \
      default: return KYBER_HIP_EINVAL;                  \
    }                                                    \
  } while (0)

// dispatch on the run-time parameter set
#define DISPATCH(K, NINETIES, FN, ...)                                   \
  do {                                                                    \
    switch ((K) * 2 + ((NINETIES) ? 1 : 0)) {                             \
      case 4: return FN<Par<2>, false>(__VA_ARGS__);                      \
      case 5: return FN<Par<2>, true>(__VA_ARGS__);                       \
      case 6: return FN<Par<3>, false>(__VA_ARGS__);                      \
      case 7: return FN<Par<3>, true>(__VA_ARGS__);                       \
      case 8: return FN<Par<4>, false>(__VA_ARGS__);                      \
      case 9: return FN<Par<4>, true>(__VA_ARGS__);                       \
      default: return KYBER_HIP_EINVAL;                                   \
    }                                                                     \
  } while (0)

}  // namespace

extern "C" {

int kyber_hip_supported(int k, int nineties) { return (k >= 2 && k <= 4 && (nineties == 0 || nineties == 1)) ? 1 : 0; }

size_t kyber_hip_pk_bytes(int k) { return (k < 2 || k > 4) ? 0 : (size_t)384 * k + 32; }
size_t kyber_hip_sk_bytes(int k) { return (k < 2 || k > 4) ? 0 : (size_t)768 * k + 96; }
size_t kyber_hip_ct_bytes(int k) { return (k < 2 || k > 4) ? 0 : (size_t)32 * ((k == 4 ? 11 : 10) * k + (k == 4 ? 5 : 4)); }

// device pointers of the KEM objects: 16-byte aligned (include/kyber_batch.h); the kernels move them as 16-byte pieces
static bool dev_aligned(std::initializer_list<const void*> ps) {
  for (const void* p : ps)
    if (reinterpret_cast<uintptr_t>(p) & 15u) {
      snprintf(g_err, sizeof g_err, "device pointer %p is not 16-byte aligned", p);
      return false;
    }
  return true;
}
int kyber_batch_keypair_dev(int k, int nineties, size_t n, uint8_t* pk, uint8_t* sk, const uint8_t* coins, void* stream) {
  if (!pk || !sk || !coins) return KYBER_HIP_EINVAL;
  if (!dev_aligned({pk, sk, coins})) return KYBER_HIP_EINVAL;
  if (n == 0) return (k < 2 || k > 4) ? KYBER_HIP_EINVAL : kyber_hip_supported(k, nineties) ? 0 : KYBER_HIP_ENOSYS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, keypair_dev, n, pk, sk, coins, st);
}
int kyber_batch_enc_dev(int k, int nineties, size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* pk, const uint8_t* coins,
                        void* stream) {
  if (!ct || !ss || !pk || !coins) return KYBER_HIP_EINVAL;
  if (!dev_aligned({ct, ss, pk, coins})) return KYBER_HIP_EINVAL;
  if (n == 0) return (k < 2 || k > 4) ? KYBER_HIP_EINVAL : kyber_hip_supported(k, nineties) ? 0 : KYBER_HIP_ENOSYS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, enc_each_dev, n, ct, ss, pk, coins, st);
}
int kyber_batch_enc_shared_pk_dev(int k, int nineties, size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* pk, const uint8_t* coins,
                                  void* stream) {
  if (!ct || !ss || !pk || !coins) return KYBER_HIP_EINVAL;
  if (!dev_aligned({ct, ss, pk, coins})) return KYBER_HIP_EINVAL;
  if (n == 0) return (k < 2 || k > 4) ? KYBER_HIP_EINVAL : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, enc_shared_dev, n, ct, ss, pk, coins, st);
}
int kyber_batch_dec_dev(int k, int nineties, size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* sk, void* stream) {
  if (!ss || !ct || !sk) return KYBER_HIP_EINVAL;
  if (!dev_aligned({ss, ct, sk})) return KYBER_HIP_EINVAL;
  if (n == 0) return (k < 2 || k > 4) ? KYBER_HIP_EINVAL : kyber_hip_supported(k, nineties) ? 0 : KYBER_HIP_ENOSYS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, dec_each_dev, n, ss, ct, sk, st);
}
int kyber_batch_dec_dev_noalloc(int k, int nineties, size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* sk, void* stream) {
  if (!ss || !ct || !sk) return KYBER_HIP_EINVAL;
  if (!dev_aligned({ss, ct, sk})) return KYBER_HIP_EINVAL;
  if (k < 2 || k > 4 || (nineties != 0 && nineties != 1)) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, dec_noalloc, n, ss, ct, sk, st);
}
size_t kyber_hip_noalloc_max_n(void) { return NOALLOC_MAX_N; }
int kyber_batch_dec_shared_sk_dev(int k, int nineties, size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* sk, void* stream) {
  if (!ss || !ct || !sk) return KYBER_HIP_EINVAL;
  if (!dev_aligned({ss, ct, sk})) return KYBER_HIP_EINVAL;
  if (n == 0) return (k < 2 || k > 4) ? KYBER_HIP_EINVAL : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, dec_shared_dev, n, ss, ct, sk, st);
}

int kyber_batch_keypair(int k, int nineties, size_t n, uint8_t* pk, uint8_t* sk, const uint8_t* coins) {
  if (!pk || !sk) return KYBER_HIP_EINVAL;
  if (n == 0) return (k < 2 || k > 4) ? KYBER_HIP_EINVAL : kyber_hip_supported(k, nineties) ? 0 : KYBER_HIP_ENOSYS;
  DISPATCH(k, nineties, keypair_host, n, pk, sk, coins);
}
int kyber_batch_enc(int k, int nineties, size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* pk, const uint8_t* coins) {
  if (!ct || !ss || !pk) return KYBER_HIP_EINVAL;
  if (n == 0) return (k < 2 || k > 4) ? KYBER_HIP_EINVAL : kyber_hip_supported(k, nineties) ? 0 : KYBER_HIP_ENOSYS;
  DISPATCH(k, nineties, enc_host, n, ct, ss, pk, coins, false);
}
int kyber_batch_enc_shared_pk(int k, int nineties, size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* pk, const uint8_t* coins) {
  if (!ct || !ss || !pk) return KYBER_HIP_EINVAL;
  if (n == 0) return (k < 2 || k > 4) ? KYBER_HIP_EINVAL : 0;
  DISPATCH(k, nineties, enc_host, n, ct, ss, pk, coins, true);
}
int kyber_batch_dec(int k, int nineties, size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* sk) {
  if (!ss || !ct || !sk) return KYBER_HIP_EINVAL;
  if (n == 0) return (k < 2 || k > 4) ? KYBER_HIP_EINVAL : kyber_hip_supported(k, nineties) ? 0 : KYBER_HIP_ENOSYS;
  DISPATCH(k, nineties, dec_host, n, ss, ct, sk, false);
}
int kyber_batch_dec_shared_sk(int k, int nineties, size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* sk) {
  if (!ss || !ct || !sk) return KYBER_HIP_EINVAL;
  if (n == 0) return (k < 2 || k > 4) ? KYBER_HIP_EINVAL : 0;
  DISPATCH(k, nineties, dec_host, n, ss, ct, sk, true);
}

// ---- prepared keys ----
static int prepare_any(int k, int nineties, const uint8_t* key, int is_sk, int on_device, kyber_hip_prepared** out) {
  if (!key || !out) return KYBER_HIP_EINVAL;
  *out = nullptr;
  DISPATCH(k, nineties, prepare_key, key, is_sk, on_device, out);
}
int kyber_hip_prepare_pk(int k, int nineties, const uint8_t* pk, kyber_hip_prepared** out) { return prepare_any(k, nineties, pk, 0, 0, out); }
int kyber_hip_prepare_sk(int k, int nineties, const uint8_t* sk, kyber_hip_prepared** out) { return prepare_any(k, nineties, sk, 1, 0, out); }
int kyber_hip_prepare_pk_dev(int k, int nineties, const uint8_t* pk, kyber_hip_prepared** out) { return prepare_any(k, nineties, pk, 0, 1, out); }
int kyber_hip_prepare_sk_dev(int k, int nineties, const uint8_t* sk, kyber_hip_prepared** out) { return prepare_any(k, nineties, sk, 1, 1, out); }
int kyber_hip_prepared_free(kyber_hip_prepared* h) {
  if (!h) return 0;
  if (h->magic != PREP_MAGIC) return KYBER_HIP_EINVAL;
  int cur = 0, rc = 0;
  HIP_TRY(hipGetDevice(&cur));
  if (cur != h->dev) HIP_TRY(hipSetDevice(h->dev));
  hipError_t e = hipDeviceSynchronize();  // no kernel may still read the key
  if (e == hipSuccess && h->is_sk) e = hipMemset(h->base, 0, h->bytes);  // a secret key was kept here
  if (e == hipSuccess) e = hipFree(h->base);
  if (e != hipSuccess) rc = hip_fail(e, "kyber_hip_prepared_free");
  if (cur != h->dev) (void)hipSetDevice(cur);
  h->magic = 0;
  delete h;
  return rc;
}
int kyber_batch_enc_prepared_dev(const kyber_hip_prepared* h, size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* coins, void* stream) {
  if (!ct || !ss || !coins) return KYBER_HIP_EINVAL;
  if (!dev_aligned({ct, ss, coins})) return KYBER_HIP_EINVAL;
  if (int rc = prepared_check(h, 0)) return rc;
  if (n == 0) return 0;
  DISPATCH(h->k, h->n90, enc_prepared_dev, h, n, ct, ss, coins, static_cast<hipStream_t>(stream));
}
int kyber_batch_dec_prepared_dev(const kyber_hip_prepared* h, size_t n, uint8_t* ss, const uint8_t* ct, void* stream) {
  if (!ss || !ct) return KYBER_HIP_EINVAL;
  if (!dev_aligned({ss, ct})) return KYBER_HIP_EINVAL;
  if (int rc = prepared_check(h, 1)) return rc;
  if (n == 0) return 0;
  DISPATCH(h->k, h->n90, dec_prepared_dev, h, n, ss, ct, static_cast<hipStream_t>(stream));
}
int kyber_batch_enc_prepared(const kyber_hip_prepared* h, size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* coins) {
  if (!ct || !ss) return KYBER_HIP_EINVAL;
  if (int rc = prepared_check(h, 0)) return rc;
  if (n == 0) return 0;
  DISPATCH(h->k, h->n90, enc_prepared_host, h, n, ct, ss, coins);
}
int kyber_batch_dec_prepared(const kyber_hip_prepared* h, size_t n, uint8_t* ss, const uint8_t* ct) {
  if (!ss || !ct) return KYBER_HIP_EINVAL;
  if (int rc = prepared_check(h, 1)) return rc;
  if (n == 0) return 0;
  DISPATCH(h->k, h->n90, dec_prepared_host, h, n, ss, ct);
}

// ---- key exchange ----
#define KEX_PRECHECK(cond)                                       \
  do {                                                           \
    if (!(cond) || (flags & ~KYBER_KEX_SHARED_B)) return KYBER_HIP_EINVAL; \
    if (k < 2 || k > 4) return KYBER_HIP_EINVAL;                 \
    if (n == 0) return 0;                                        \
  } while (0)

int kyber_batch_kex_uake_initA(int k, int nineties, size_t n, uint8_t* send, uint8_t* tk, uint8_t* sk, const uint8_t* pkb,
                               const uint8_t* coins, int flags) {
  KEX_PRECHECK(send && tk && sk && pkb);
  DISPATCH(k, nineties, kex_initA_host, n, send, tk, sk, pkb, coins, flags & KYBER_KEX_SHARED_B);
}
int kyber_batch_kex_ake_initA(int k, int nineties, size_t n, uint8_t* send, uint8_t* tk, uint8_t* sk, const uint8_t* pkb,
                              const uint8_t* coins, int flags) {
  return kyber_batch_kex_uake_initA(k, nineties, n, send, tk, sk, pkb, coins, flags);  // same body, ref/kex.c:30-34
}
int kyber_batch_kex_uake_sharedB(int k, int nineties, size_t n, uint8_t* send, uint8_t* key, const uint8_t* recv,
                                 const uint8_t* skb, const uint8_t* coins, int flags) {
  KEX_PRECHECK(send && key && recv && skb);
  DISPATCH(k, nineties, kex_uake_sharedB_host, n, send, key, recv, skb, coins, flags & KYBER_KEX_SHARED_B);
}
int kyber_batch_kex_uake_sharedA(int k, int nineties, size_t n, uint8_t* key, const uint8_t* recv, const uint8_t* tk,
                                 const uint8_t* sk) {
  const int flags = 0;
  KEX_PRECHECK(key && recv && tk && sk);
  DISPATCH(k, nineties, kex_uake_sharedA_host, n, key, recv, tk, sk);
}
int kyber_batch_kex_ake_sharedB(int k, int nineties, size_t n, uint8_t* send, uint8_t* key, const uint8_t* recv,
                                const uint8_t* skb, const uint8_t* pka, const uint8_t* coins, int flags) {
  KEX_PRECHECK(send && key && recv && skb && pka);
  DISPATCH(k, nineties, kex_ake_sharedB_host, n, send, key, recv, skb, pka, coins, flags & KYBER_KEX_SHARED_B);
}
int kyber_batch_kex_ake_sharedA(int k, int nineties, size_t n, uint8_t* key, const uint8_t* recv, const uint8_t* tk,
                                const uint8_t* sk, const uint8_t* ska) {
  const int flags = 0;
  KEX_PRECHECK(key && recv && tk && sk && ska);
  DISPATCH(k, nineties, kex_ake_sharedA_host, n, key, recv, tk, sk, ska);
}

int kyber_batch_kex_uake_initA_dev(int k, int nineties, size_t n, uint8_t* send, uint8_t* tk, uint8_t* sk, const uint8_t* pkb,
                                   const uint8_t* coins, int flags, void* stream) {
  KEX_PRECHECK(send && tk && sk && pkb && coins);
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, kex_initA_dev, n, send, tk, sk, pkb, coins, flags & KYBER_KEX_SHARED_B, st);
}
int kyber_batch_kex_ake_initA_dev(int k, int nineties, size_t n, uint8_t* send, uint8_t* tk, uint8_t* sk, const uint8_t* pkb,
                                  const uint8_t* coins, int flags, void* stream) {
  return kyber_batch_kex_uake_initA_dev(k, nineties, n, send, tk, sk, pkb, coins, flags, stream);
}
int kyber_batch_kex_uake_sharedB_dev(int k, int nineties, size_t n, uint8_t* send, uint8_t* key, const uint8_t* recv,
                                     const uint8_t* skb, const uint8_t* coins, int flags, void* stream) {
  KEX_PRECHECK(send && key && recv && skb && coins);
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, kex_uake_sharedB_dev, n, send, key, recv, skb, coins, flags & KYBER_KEX_SHARED_B, st);
}
int kyber_batch_kex_uake_sharedA_dev(int k, int nineties, size_t n, uint8_t* key, const uint8_t* recv, const uint8_t* tk,
                                     const uint8_t* sk, void* stream) {
  const int flags = 0;
  KEX_PRECHECK(key && recv && tk && sk);
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, kex_uake_sharedA_dev, n, key, recv, tk, sk, st);
}
int kyber_batch_kex_ake_sharedB_dev(int k, int nineties, size_t n, uint8_t* send, uint8_t* key, const uint8_t* recv,
                                    const uint8_t* skb, const uint8_t* pka, const uint8_t* coins, int flags, void* stream) {
  KEX_PRECHECK(send && key && recv && skb && pka && coins);
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, kex_ake_sharedB_dev, n, send, key, recv, skb, pka, coins, flags & KYBER_KEX_SHARED_B, st);
}
int kyber_batch_kex_ake_sharedA_dev(int k, int nineties, size_t n, uint8_t* key, const uint8_t* recv, const uint8_t* tk,
                                    const uint8_t* sk, const uint8_t* ska, void* stream) {
  const int flags = 0;
  KEX_PRECHECK(key && recv && tk && sk && ska);
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, kex_ake_sharedA_dev, n, key, recv, tk, sk, ska, st);
}

// ---- ML-KEM (FIPS 203), include/mlkem_batch.h ----
int mlkem_batch_keypair(int k, size_t n, uint8_t* ek, uint8_t* dk, const uint8_t* coins) {
  if (!ek || !dk || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  DISPATCH_ML(k, ml_keypair_host, n, ek, dk, coins);
}
int mlkem_batch_encaps(int k, size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* ek, const uint8_t* coins) {
  if (!ct || !ss || !ek || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  DISPATCH_ML(k, ml_enc_host, n, ct, ss, ek, coins);
}
int mlkem_batch_decaps(int k, size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* dk) {
  if (!ss || !ct || !dk || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  DISPATCH_ML(k, ml_dec_host, n, ss, ct, dk);
}
int mlkem_batch_keypair_dev(int k, size_t n, uint8_t* ek, uint8_t* dk, const uint8_t* coins, void* stream) {
  if (!ek || !dk || !coins || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  if (!dev_aligned({ek, dk, coins})) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  DISPATCH_ML(k, ml_keypair_dev, n, ek, dk, coins, static_cast<hipStream_t>(stream));
}
int mlkem_batch_encaps_dev(int k, size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* ek, const uint8_t* coins, void* stream) {
  if (!ct || !ss || !ek || !coins || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  if (!dev_aligned({ct, ss, ek, coins})) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  DISPATCH_ML(k, ml_enc_dev, n, ct, ss, ek, coins, static_cast<hipStream_t>(stream));
}
int mlkem_batch_decaps_dev(int k, size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* dk, void* stream) {
  if (!ss || !ct || !dk || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  if (!dev_aligned({ss, ct, dk})) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  DISPATCH_ML(k, ml_dec_dev, n, ss, ct, dk, static_cast<hipStream_t>(stream));
}
int mlkem_batch_check_ek(int k, size_t n, const uint8_t* ek, uint32_t* flags) {
  if (!ek || !flags || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  DISPATCH_ML(k, ml_check_host, n, ek, 0, flags);
}
int mlkem_batch_check_dk(int k, size_t n, const uint8_t* dk, uint32_t* flags) {
  if (!dk || !flags || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  DISPATCH_ML(k, ml_check_host, n, dk, 1, flags);
}
int mlkem_batch_check_dev(int k, size_t n, const uint8_t* key, int is_dk, uint32_t* flags, void* stream) {
  if (!key || !flags || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  DISPATCH_ML(k, ml_check_dev, n, key, is_dk, flags, static_cast<hipStream_t>(stream));
}

int kyber_batch_gen_matrix_dev(int k, int nineties, size_t n, int16_t* mat, const uint8_t* seeds, int transposed, void* stream) {
  if (!mat || !seeds) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, gen_matrix_dev, n, mat, seeds, transposed, st);
}

int kyber_batch_keccak_f1600_dev(size_t n, uint64_t* states, void* stream) {
  if (!states || n > 0x7fffffffu) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  RUN("k_st_keccak_f1600", st, k_st_keccak_f1600<<<blocks(n, sponge_wg(n)), sponge_wg(n), 0, st>>>((int)n, states));
  return 0;
}
int kyber_batch_hash_h_dev(int k, int nineties, size_t n, int of_ct, uint8_t* out, const uint8_t* in, void* stream) {
  if (!out || !in) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, hash_h_dev, n, of_ct, out, in, st);
}
int kyber_batch_hash_g_dev(int nineties, size_t n, size_t inlen, uint8_t* out, const uint8_t* in, void* stream) {
  if (!out || !in || (inlen != 32 && inlen != 64) || n > 0x7fffffffu || (nineties != 0 && nineties != 1)) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned wg = sponge_wg(n), nb = blocks(n, wg);
  if (inlen == 32 && nineties) RUN("k_st_hash_g", st, (k_st_hash_g<1, true><<<nb, wg, 0, st>>>((int)n, in, out)));
  else if (inlen == 32) RUN("k_st_hash_g", st, (k_st_hash_g<1, false><<<nb, wg, 0, st>>>((int)n, in, out)));
  else if (nineties) RUN("k_st_hash_g", st, (k_st_hash_g<2, true><<<nb, wg, 0, st>>>((int)n, in, out)));
  else RUN("k_st_hash_g", st, (k_st_hash_g<2, false><<<nb, wg, 0, st>>>((int)n, in, out)));
  return 0;
}
int kyber_batch_aes256ctr_dev(size_t n, uint8_t* out, size_t nblocks, const uint8_t* keys, const uint16_t* nonces, uint32_t first_block,
                              void* stream) {
  if (!out || !keys || !nonces || nblocks > (1u << 20) || n > (size_t)INT32_MAX) return KYBER_HIP_EINVAL;
  if (n == 0 || nblocks == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  RUN("k_st_aes256ctr", st, k_st_aes256ctr<AES_WG_SMALL><<<blocks(n, AES_WG_SMALL), AES_WG_SMALL, 0, st>>>((int)n, keys, nonces, first_block, (int)nblocks, out));
  return 0;
}
int kyber_batch_getnoise_dev(int k, int nineties, size_t n, int nnonce, int n_eta1, int16_t* r, const uint8_t* seeds, void* stream) {
  if (!r || !seeds) return KYBER_HIP_EINVAL;
  if (n == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH(k, nineties, getnoise_dev, n, nnonce, n_eta1, r, seeds, st);
}
int kyber_batch_poly_cbd_dev(int eta, size_t npoly, int16_t* r, const uint8_t* buf, void* stream) {
  if (!r || !buf || (eta != 2 && eta != 3)) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_st_cbd", static_cast<hipStream_t>(stream), (void)launch_st_cbd(static_cast<hipStream_t>(stream), eta, npoly, r, buf));
  return 0;
}
int kyber_batch_poly_tobytes_dev(size_t npoly, uint8_t* r, const int16_t* a, void* stream) {
  if (!r || !a) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_st_tobytes", static_cast<hipStream_t>(stream), launch_st_tobytes(static_cast<hipStream_t>(stream), npoly, r, a));
  return 0;
}
int kyber_batch_poly_frombytes_dev(size_t npoly, int16_t* r, const uint8_t* a, void* stream) {
  if (!r || !a) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_st_frombytes", static_cast<hipStream_t>(stream), launch_st_frombytes(static_cast<hipStream_t>(stream), npoly, r, a));
  return 0;
}
static bool codec_bits_ok(int d) { return d == 1 || d == 4 || d == 5 || d == 10 || d == 11; }
int kyber_batch_poly_compress_dev(int d, size_t npoly, uint8_t* r, const int16_t* a, void* stream) {
  if (!r || !a || !codec_bits_ok(d)) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_st_compress", static_cast<hipStream_t>(stream), (void)launch_st_compress(static_cast<hipStream_t>(stream), d, npoly, r, a));
  return 0;
}
int kyber_batch_poly_decompress_dev(int d, size_t npoly, int16_t* r, const uint8_t* a, void* stream) {
  if (!r || !a || !codec_bits_ok(d)) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_st_decompress", static_cast<hipStream_t>(stream), (void)launch_st_decompress(static_cast<hipStream_t>(stream), d, npoly, r, a));
  return 0;
}
int kyber_batch_polyvec_basemul_acc_dev(int k, size_t nvec, int16_t* r, const int16_t* a, const int16_t* b, void* stream) {
  if (!r || !a || !b || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  if (!nvec) return 0;
  RUN("k_st_basemul_acc", static_cast<hipStream_t>(stream), (void)launch_st_basemul_acc(static_cast<hipStream_t>(stream), k, nvec, r, a, b));
  return 0;
}

int kyber_batch_poly_ntt_dev(size_t npoly, int16_t* r, void* stream) {
  if (!r) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_poly_ntt", static_cast<hipStream_t>(stream), launch_poly_ntt(static_cast<hipStream_t>(stream), npoly, r));
  return 0;
}
int kyber_batch_poly_invntt_dev(size_t npoly, int16_t* r, void* stream) {
  if (!r) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_poly_invntt", static_cast<hipStream_t>(stream), launch_poly_invntt(static_cast<hipStream_t>(stream), npoly, r));
  return 0;
}
int kyber_batch_poly_basemul_dev(size_t npoly, int16_t* r, const int16_t* a, const int16_t* b, void* stream) {
  if (!r || !a || !b) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_poly_basemul", static_cast<hipStream_t>(stream), launch_poly_basemul(static_cast<hipStream_t>(stream), npoly, r, a, b));
  return 0;
}
int kyber_batch_poly_mul_fused_dev(size_t npoly, int16_t* r, const int16_t* a, const int16_t* b, void* stream) {
  if (!r || !a || !b) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_poly_mul_fused", static_cast<hipStream_t>(stream), launch_poly_mul_fused(static_cast<hipStream_t>(stream), npoly, r, a, b));
  return 0;
}
int kyber_batch_poly_ntt_canon_dev(size_t npoly, int16_t* r, void* stream) {
  if (!r) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_wpoly_ntt_canon", static_cast<hipStream_t>(stream), launch_poly_ntt_canon(static_cast<hipStream_t>(stream), npoly, r));
  return 0;
}
int kyber_batch_poly_invntt_canon_dev(size_t npoly, int16_t* r, void* stream) {
  if (!r) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_wpoly_invntt_canon", static_cast<hipStream_t>(stream), launch_poly_invntt_canon(static_cast<hipStream_t>(stream), npoly, r));
  return 0;
}
int kyber_batch_poly_basemul_canon_dev(size_t npoly, int16_t* r, const int16_t* a, const int16_t* b, void* stream) {
  if (!r || !a || !b) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_wpoly_basemul_canon", static_cast<hipStream_t>(stream), launch_poly_basemul_canon(static_cast<hipStream_t>(stream), npoly, r, a, b));
  return 0;
}
int kyber_batch_poly_mul_fused_canon_dev(size_t npoly, int16_t* r, const int16_t* a, const int16_t* b, void* stream) {
  if (!r || !a || !b) return KYBER_HIP_EINVAL;
  if (!npoly) return 0;
  RUN("k_wpoly_mul_fused_canon", static_cast<hipStream_t>(stream), launch_poly_mul_fused_canon(static_cast<hipStream_t>(stream), npoly, r, a, b));
  return 0;
}

// ---- one node, several GPUs: contiguous shards, one host thread per device ----
// Instances share nothing (ref/kem.c:23-127 keep no state), so shard g of G is instances
// [g*n/G, (g+1)*n/G) of every array, run through the ordinary host-pointer entry point on
// device devices[g] by a thread of its own; no device talks to another.
extern "C++" {
namespace {
template <class F>
int run_sharded(size_t n, int ndev, const int* devices, F&& shard_call) {
  if (ndev < 1 || ndev > MAX_DEV) return KYBER_HIP_EINVAL;
  int count = 0;
  HIP_TRY(hipGetDeviceCount(&count));
  for (int g = 0; g < ndev; g++) {
    const int d = devices ? devices[g] : g;
    if (d < 0 || d >= count) {
      snprintf(g_err, sizeof g_err, "device %d not present (%d visible)", d, count);
      return KYBER_HIP_EINVAL;
    }
  }
  std::vector<int> rc(ndev, 0);
  std::vector<std::string> err(ndev);
  std::vector<std::thread> th;
  for (int g = 0; g < ndev; g++) {
    const size_t lo = n / ndev * g + std::min<size_t>(g, n % ndev);
    const size_t cnt = n / ndev + ((size_t)g < n % ndev ? 1 : 0);
    const int d = devices ? devices[g] : g;
    th.emplace_back([&, g, lo, cnt, d] {
      if (cnt == 0) return;
      hipError_t e = hipSetDevice(d);  // the current device is per host thread
      rc[g] = e == hipSuccess ? shard_call(lo, cnt) : hip_fail(e, "hipSetDevice");
      if (rc[g]) err[g] = g_err;  // g_err is thread-local: hand the text to the caller's thread
    });
  }
  for (auto& t : th) t.join();
  for (int g = 0; g < ndev; g++)
    if (rc[g]) {
      snprintf(g_err, sizeof g_err, "shard %d: %s", g, err[g].c_str());
      return rc[g];
    }
  return 0;
}
}  // namespace
}  // extern "C++"

// coins == NULL: every draw from randombytes() happens here, on the calling thread and in instance
// order, before the shards start -- an entropy hook with static state (the reference's own test
// doubles: ref/test_vectors.c:41, ref/rng.c:155) must not be entered by ndev threads at once.
int kyber_batch_keypair_mgpu(int k, int nineties, size_t n, uint8_t* pk, uint8_t* sk, const uint8_t* coins, int ndev,
                             const int* devices) {
  if (!pk || !sk || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  const size_t pkb = kyber_hip_pk_bytes(k), skb = kyber_hip_sk_bytes(k);
  Coins c(coins, n * 64);
  return run_sharded(n, ndev, devices, [&](size_t lo, size_t cnt) {
    return kyber_batch_keypair(k, nineties, cnt, pk + lo * pkb, sk + lo * skb, c.p + lo * 64);
  });
}
int kyber_batch_enc_mgpu(int k, int nineties, size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* pk, const uint8_t* coins, int ndev,
                         const int* devices) {
  if (!ct || !ss || !pk || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  const size_t pkb = kyber_hip_pk_bytes(k), ctb = kyber_hip_ct_bytes(k);
  Coins c(coins, n * 32);
  return run_sharded(n, ndev, devices, [&](size_t lo, size_t cnt) {
    return kyber_batch_enc(k, nineties, cnt, ct + lo * ctb, ss + lo * 32, pk + lo * pkb, c.p + lo * 32);
  });
}
int kyber_batch_dec_mgpu(int k, int nineties, size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* sk, int ndev,
                         const int* devices) {
  if (!ss || !ct || !sk || k < 2 || k > 4) return KYBER_HIP_EINVAL;
  const size_t skb = kyber_hip_sk_bytes(k), ctb = kyber_hip_ct_bytes(k);
  return run_sharded(n, ndev, devices, [&](size_t lo, size_t cnt) {
    return kyber_batch_dec(k, nineties, cnt, ss + lo * 32, ct + lo * ctb, sk + lo * skb);
  });
}

// Device-resident shards: shard g lives on HIP device devices[g] (its own pk/sk/ct/ss/coins arrays,
// n[g] instances) and its kernels are queued on streams[g].  Nothing is copied and nothing is waited
// for: the calling thread visits the devices in turn and enqueues, the devices then run side by side.
extern "C++" {
namespace {
template <class F>
int each_device(int ndev, const int* devices, F&& call) {
  if (ndev < 1 || ndev > MAX_DEV || !devices) return KYBER_HIP_EINVAL;
  int count = 0, home = 0, rc = 0;
  HIP_TRY(hipGetDeviceCount(&count));
  HIP_TRY(hipGetDevice(&home));
  for (int g = 0; g < ndev; g++)
    if (devices[g] < 0 || devices[g] >= count) {
      snprintf(g_err, sizeof g_err, "device %d not present (%d visible)", devices[g], count);
      return KYBER_HIP_EINVAL;
    }
  for (int g = 0; g < ndev && rc == 0; g++) {
    hipError_t e = hipSetDevice(devices[g]);
    rc = e == hipSuccess ? call(g) : hip_fail(e, "hipSetDevice");
    if (rc) {
      char msg[200];
      snprintf(msg, sizeof msg, "shard %d: %s", g, g_err);
      snprintf(g_err, sizeof g_err, "%s", msg);
    }
  }
  (void)hipSetDevice(home);
  return rc;
}
}  // namespace
}  // extern "C++"

int kyber_batch_keypair_mgpu_dev(int k, int nineties, int ndev, const int* devices, const size_t* n, uint8_t* const* pk,
                                 uint8_t* const* sk, const uint8_t* const* coins, void* const* streams) {
  if (!n || !pk || !sk || !coins) return KYBER_HIP_EINVAL;
  return each_device(ndev, devices, [&](int g) {
    return kyber_batch_keypair_dev(k, nineties, n[g], pk[g], sk[g], coins[g], streams ? streams[g] : nullptr);
  });
}
int kyber_batch_enc_mgpu_dev(int k, int nineties, int ndev, const int* devices, const size_t* n, uint8_t* const* ct, uint8_t* const* ss,
                             const uint8_t* const* pk, const uint8_t* const* coins, void* const* streams) {
  if (!n || !ct || !ss || !pk || !coins) return KYBER_HIP_EINVAL;
  return each_device(ndev, devices, [&](int g) {
    return kyber_batch_enc_dev(k, nineties, n[g], ct[g], ss[g], pk[g], coins[g], streams ? streams[g] : nullptr);
  });
}
int kyber_batch_dec_mgpu_dev(int k, int nineties, int ndev, const int* devices, const size_t* n, uint8_t* const* ss,
                             const uint8_t* const* ct, const uint8_t* const* sk, void* const* streams) {
  if (!n || !ss || !ct || !sk) return KYBER_HIP_EINVAL;
  return each_device(ndev, devices, [&](int g) {
    return kyber_batch_dec_dev(k, nineties, n[g], ss[g], ct[g], sk[g], streams ? streams[g] : nullptr);
  });
}

void* kyber_hip_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void kyber_hip_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int kyber_hip_set_chunk(size_t instances) {
  if (instances > ((size_t)1 << 24)) return KYBER_HIP_EINVAL;  // chunk * K^2 matrix entries are indexed with int
  g_chunk = instances ? instances : DEFAULT_CHUNK;
  return 0;
}

int kyber_hip_set_workspace_slots(int slots) {
  if (slots < 1 || slots > MAX_SLOTS) return KYBER_HIP_EINVAL;
  g_slots = slots;
  return 0;
}

int kyber_hip_set_tail_capacity(size_t records) {
  g_tail_cap_max = records;
  return 0;
}

// every device the library holds something on: workspace, host-path staging and streams, and the
// contents of the no-allocation scratch array (the array itself is part of the code object)
static int release_device(int dev) {
  std::unique_lock<std::mutex> hlk[MAX_SLOTS];  // same order as the host-pointer calls: host lock(s), then workspace lock
  for (int s = 0; s < MAX_SLOTS; s++) hlk[s] = std::unique_lock<std::mutex>(g_host_mu[dev][s]);
  std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
  bool any = g_scratch_ptr[dev] != nullptr;
  for (int s = 0; s < MAX_SLOTS; s++) any = any || g_host[dev][s] || g_ws[dev][s].base || g_ws[dev][s].idle || g_side[dev][s].s;
  if (!any) return 0;
  HIP_TRY(hipSetDevice(dev));
  HIP_TRY(hipDeviceSynchronize());
  for (int s = 0; s < MAX_SLOTS; s++) {
    Side& sd = g_side[dev][s];
    if (sd.s) {
      (void)hipStreamDestroy(sd.s);
      (void)hipEventDestroy(sd.fork);
      (void)hipEventDestroy(sd.join);
      sd = Side{};
    }
    Workspace& w = g_ws[dev][s];
    if (w.base) {
      HIP_TRY(hipMemset(w.base, 0, w.bytes));  // sigma, r, m, Kbar, PRF bytes of the last call
      HIP_TRY(hipFree(w.base));
    }
    if (w.idle) (void)hipEventDestroy(w.idle);
    w = Workspace{};
  }
  for (int s = 0; s < MAX_SLOTS; s++) {
    HostCtx* c = g_host[dev][s];
    if (!c) continue;
    if (c->staging) {
      HIP_TRY(hipMemset(c->staging, 0, c->cap));  // secret keys and shared secrets were staged here
      HIP_TRY(hipFree(c->staging));
      c->staging = nullptr;
      c->cap = 0;
    }
    if (c->mirror) {  // wiped after every call (host_pipeline)
      HIP_TRY(hipHostFree(c->mirror));
      c->mirror = nullptr;
    }
    delete c;  // streams and events
    g_host[dev][s] = nullptr;
  }
  if (g_scratch_ptr[dev]) {
    HIP_TRY(hipMemset(g_scratch_ptr[dev], 0, NOALLOC_SCRATCH_BYTES));  // m', r', Kbar', PRF bytes of the last no-allocation call
    g_scratch_order[dev].used = false;
  }
  return 0;
}
int kyber_hip_release(void) {
  int home = 0, count = 0, rc = 0;
  HIP_TRY(hipGetDevice(&home));
  HIP_TRY(hipGetDeviceCount(&count));
  for (int dev = 0; dev < count && dev < MAX_DEV; dev++)
    if (int r = release_device(dev)) rc = rc ? rc : r;
  (void)hipSetDevice(home);
  return rc;
}

const char* kyber_hip_last_error(void) { return g_err; }

int kyber_hip_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
  g_prof_on = on != 0;
  return 0;
}

int kyber_hip_profile_read(char* buf, size_t len) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  HIP_TRY(hipDeviceSynchronize());
  std::map<std::string, std::pair<double, long>> agg;
  for (auto& r : g_prof) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
    auto& e = agg[r.name];
    e.first += ms;
    e.second += 1;
  }
  size_t o = 0;
  if (len) buf[0] = 0;
  for (auto& kv : agg) {
    int w = snprintf(buf + o, o < len ? len - o : 0, "%s %ld %.6f\n", kv.first.c_str(), kv.second.second, kv.second.first);
    if (w < 0 || o + (size_t)w >= len) break;
    o += (size_t)w;
  }
  return 0;
}

// weak default entropy source; retries on EINTR and aborts on any other OS failure like the reference's (ref/randombytes.c:46-50)
__attribute__((weak)) void randombytes(uint8_t* out, size_t outlen) {
  while (outlen) {
    ssize_t r = getrandom(out, outlen > 256 ? 256 : outlen, 0);
    if (r < 0 && errno == EINTR) continue;  // ref/randombytes.c:46-50
    if (r < 0) abort();
    out += r;
    outlen -= (size_t)r;
  }
}

// ---- per-set batch spellings + the namespaced single-call API ---------------------
#define DEFINE_SET(NAME, KK, N90)                                                                              \
  int NAME##_batch_keypair(size_t n, uint8_t* pk, uint8_t* sk, const uint8_t* coins) {                         \
    return kyber_batch_keypair(KK, N90, n, pk, sk, coins);                                                     \
  }                                                                                                            \
  int NAME##_batch_enc(size_t n, uint8_t* ct, uint8_t* ss, const uint8_t* pk, const uint8_t* coins) {          \
    return kyber_batch_enc(KK, N90, n, ct, ss, pk, coins);                                                     \
  }                                                                                                            \
  int NAME##_batch_dec(size_t n, uint8_t* ss, const uint8_t* ct, const uint8_t* sk) {                          \
    return kyber_batch_dec(KK, N90, n, ss, ct, sk);                                                            \
  }                                                                                                            \
  int pqcrystals_##NAME##_hip_keypair(uint8_t* pk, uint8_t* sk) {                                              \
    uint8_t coins[64];                                                                                         \
    randombytes(coins, 32);      /* d,  ref/indcpa.c:215 */                                                    \
    randombytes(coins + 32, 32); /* z,  ref/kem.c:32     */                                                    \
    const int rc = kyber_batch_keypair(KK, N90, 1, pk, sk, coins);                                             \
    explicit_bzero(coins, sizeof coins); /* d and z do not outlive the call on this stack */                   \
    return rc;                                                                                                 \
  }                                                                                                            \
  int pqcrystals_##NAME##_hip_enc(uint8_t* ct, uint8_t* ss, const uint8_t* pk) {                               \
    uint8_t coins[32];                                                                                         \
    randombytes(coins, 32); /* m0, ref/kem.c:59 */                                                             \
    const int rc = kyber_batch_enc(KK, N90, 1, ct, ss, pk, coins);                                             \
    explicit_bzero(coins, sizeof coins);                                                                       \
    return rc;                                                                                                 \
  }                                                                                                            \
  int pqcrystals_##NAME##_hip_dec(uint8_t* ss, const uint8_t* ct, const uint8_t* sk) {                         \
    return kyber_batch_dec(KK, N90, 1, ss, ct, sk);                                                            \
  }

DEFINE_SET(kyber512, 2, 0)
DEFINE_SET(kyber768, 3, 0)
DEFINE_SET(kyber1024, 4, 0)
DEFINE_SET(kyber512_90s, 2, 1)
DEFINE_SET(kyber768_90s, 3, 1)
DEFINE_SET(kyber1024_90s, 4, 1)

}  // extern "C"
