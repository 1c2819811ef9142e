// C ABI of the MI355X back-end (include/summa_gpu.h).  Thin: argument checks, staging of
// host buffers, per-device context (stream, workspaces, twiddle / SRS caches), dispatch to
// the NTT and MSM engines.  No CPU implementation of the path exists in this library: if
// HIP is unusable every entry point fails with SG_ERR_NO_DEVICE / SG_ERR_HIP.
#include "../../include/summa_gpu.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "host_curve.h"
#include "host_pairing.h"
#include "../../include/summa_prover.hpp"
#include "../../include/summa_circuit.hpp"
#include "msm.h"
#include "ntt.h"
#include "quotient.h"
#include "gates.h"
#include "numerator.h"
#include "poly.h"
#include "witness.h"
#include "side_prio.cuh"
#include "host_wait.h"

using namespace sg;
namespace sg {
SG_DEFINE_SIDE_PRIO_SETTER(abi_set_side_prio)
}

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* what, hipError_t e = hipSuccess) {
  if (e != hipSuccess) std::snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  else std::snprintf(g_err, sizeof g_err, "%s", what);
  return code;
}
int hip_fail(const char* what, hipError_t e) {
  return fail(e == hipErrorOutOfMemory ? SG_ERR_NOMEM : SG_ERR_HIP, what, e);
}
#define CHECK_HIP(call, what)                  \
  do {                                         \
    hipError_t _e = (call);                    \
    if (_e != hipSuccess) return hip_fail(what, _e); \
  } while (0)

// BN254 Fr constants as Montgomery-2^256 words
__device__ const uint32_t ROOT_OF_UNITY_M[8] = {0xb639feb8u, 0x9632c7c5u, 0x0d0ff299u, 0x985ce340u,
                                                0x01b0ecd8u, 0xb2dd8800u, 0x6d98ce29u, 0x1d69070du};  // order 2^28
__device__ const uint32_t ZETA_M[8] = {0x55fcd653u, 0x0363f299u, 0x5fc1e200u, 0x73e7950bu,
                                       0x576d9d24u, 0xc5fce83eu, 0xa1c3a4d4u, 0x059c805du};  // Fr::ZETA

struct DomainConsts {  // all Montgomery-2^256 words
  words8 omega, omega_inv, n_inv, zeta, zeta2, ninv_zeta2, ninv_zeta, one;
};
__device__ void put_words(words8* dst, const f29& v_r261) {
  f29_to_words(f29_reduce_with<Fr29>(v_r261, Fr29::r256), dst->l);
}
// EvaluationDomain::new constants for 2^k (computed in the 2^261 domain, exported as words)
__global__ void domain_kernel(uint32_t k, DomainConsts* out) {
  side_kernel_prio();
  typedef Fr29 P;
  uint32_t rw[8], zw[8];
  for (int i = 0; i < 8; i++) { rw[i] = ROOT_OF_UNITY_M[i]; zw[i] = ZETA_M[i]; }
  f29 w = f29_words_to_r261<P>(rw), z = f29_words_to_r261<P>(zw);
  for (uint32_t i = k; i < 28; i++) w = f29_sqr<P>(w);
  // 2^k in the 2^261 domain: canonical integer times 2^522 * 2^-261; build it by doubling 1^
  f29 n = f29_one<P>();
  for (uint32_t i = 0; i < k; i++) n = f29_cond_sub_p<P>(f29_normalize(f29_dbl(n)));
  f29 ninv = f29_inv<P>(n);
  f29 z2 = f29_sqr<P>(z);
  put_words(&out->omega, w);
  put_words(&out->omega_inv, f29_inv<P>(w));
  put_words(&out->n_inv, ninv);
  put_words(&out->zeta, z);
  put_words(&out->zeta2, z2);
  put_words(&out->ninv_zeta2, f29_mul<P>(ninv, z2));
  put_words(&out->ninv_zeta, f29_mul<P>(ninv, z));
  put_words(&out->one, f29_one<P>());
}
// t_evaluations[i] = 1 / ((zeta * omega_ext^i)^(2^k) - 1), i < 2^(ext_k - k)
__global__ void t_eval_kernel(uint32_t k, uint32_t ext_k, words8 omega_ext, fp_words* out) {
  side_kernel_prio();
  typedef Fr29 P;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> (ext_k - k)) return;
  uint32_t zw[8];
  for (int j = 0; j < 8; j++) zw[j] = ZETA_M[j];
  f29 x = f29_mul<P>(f29_words_to_r261<P>(zw), f29_pow_u64<P>(f29_words_to_r261<P>(omega_ext.l), i));
  for (uint32_t s = 0; s < k; s++) x = f29_sqr<P>(x);
  x = f29_sub<P, 0>(x, f29_one<P>());
  uint32_t o[8];
  f29_to_words(f29_reduce_with<P>(f29_inv<P>(f29_mul<P>(x, f29_one<P>())), P::r256), o);
  fp_words_store(out + i, o);
}

// ParamsKZG::setup scalars: pw[i] = tau^i, lg[i] = L_i(tau) = omega^i (tau^n - 1) / (n (tau - omega^i))
// (Montgomery-2^256 words); the group part is g1_fixed_base_mul over them.
__global__ void kzg_setup_scalars(uint32_t k, words8 tau_w, fp_words* pw, fp_words* lg) {
  side_kernel_prio();
  typedef Fr29 P;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> k) return;
  uint32_t rw[8];
  for (int q = 0; q < 8; q++) rw[q] = ROOT_OF_UNITY_M[q];
  f29 omega = f29_words_to_r261<P>(rw);
  for (uint32_t q = k; q < 28; q++) omega = f29_sqr<P>(omega);
  const f29 tau = f29_words_to_r261<P>(tau_w.l);
  uint32_t o[8];
  f29_to_words(f29_reduce_with<P>(f29_pow_u64<P>(tau, i), P::r256), o);
  fp_words_store(pw + i, o);
  f29 tn = tau;
  for (uint32_t q = 0; q < k; q++) tn = f29_sqr<P>(tn);
  f29 num = f29_sub<P, 0>(tn, f29_one<P>());                   // tau^n - 1
  f29 n = f29_one<P>();
  for (uint32_t q = 0; q < k; q++) n = f29_cond_sub_p<P>(f29_normalize(f29_dbl(n)));
  const f29 wi = f29_pow_u64<P>(omega, i);
  f29 den = f29_mul<P>(n, f29_sub<P, 0>(tau, wi));             // n (tau - omega^i)
  f29 l = f29_mul<P>(f29_mul<P>(wi, num), f29_inv<P>(den));
  f29_to_words(f29_reduce_with<P>(l, P::r256), o);
  fp_words_store(lg + i, o);
}

struct Srs {
  uint32_t k;
  g1_affine_mem* g;
  g1_affine_mem* g_lagrange;
  FixedTable tab[3];  // optional precomputed window tables (sg_srs_precompute): [0] g, [1] g_lagrange, [2] the prefix sums
                      // of g_lagrange (difference-form commitments of Lagrange columns)
  g1_affine_mem* lagrange_prefix = nullptr;   // made with tab[2]
};

struct Context {
  int device = -1;
  hipStream_t stream = nullptr;
  NttEngine ntt;
  MsmEngine msm, msm_b;          // two engines: batches ping-pong between them
  WitnessEngine witness;
  hipStream_t bstream[2] = {nullptr, nullptr};
  hipStream_t tstream[2] = {nullptr, nullptr};  // high-priority tails
  hipEvent_t ev_in = nullptr;
  DevBuf<uint8_t> stage_a, stage_b, scratch;
  std::vector<uint8_t> gate_blob_host;
  std::map<uint64_t, GateProgram> gate_cache;  // lowered gate programs by structure hash
  uint64_t gate_recent[4] = {0, 0, 0, 0};      // keys of the programs used last (tried first, by comparison)
  uint32_t gate_recent_next = 0;
  // in-place multi-pass transforms need a scratch vector; one per caller stream, so that transforms
  // enqueued on a side stream never share it with work in flight on another stream
  std::map<hipStream_t, DevBuf<uint8_t>> ntt_scratch;
  // work space of the scan-type helpers (prefix / grand products, Kate division): per (stream, slot), so
  // that the calls are asynchronous -- work on one stream is ordered, other streams own other buffers
  std::map<std::pair<hipStream_t, int>, DevBuf<uint8_t>> stream_scratch;
  DomainConsts* d_consts = nullptr;
  struct CosetTables {       // sg_coeff_to_cosets / sg_cosets_to_pieces: per (k, ext_k, cosets)
    fp_words* fwd = nullptr;  // [nc][n] c_b^i   (2^261-domain words)
    fp_words* inv = nullptr;  // [nc][n] c_b^-i
    words8 shift[MAX_COSETS]; // c_b = zeta omega_ext^b
    uint32_t m[MAX_COSETS * MAX_COSETS][8];   // V^-1 diag(1 / (c_b^n - 1)), row-major [t][b]
  };
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, CosetTables> coset_tables;
  struct BlobSlot {   // sg_quotient_gates: program blobs in flight
    uint8_t* host = nullptr;
    uint8_t* dev = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
  };
  static constexpr uint32_t BLOB_RING = 16;
  BlobSlot blob_ring[BLOB_RING];
  uint32_t blob_next = 0;
  // sg_fr_kate_division_batch: the divisions' power tables (host-computed, 684 B each) on their way to the device -- a ring like
  // the one above, so that the call returns without waiting for the stream
  static constexpr uint32_t KATE_RING = 4;
  BlobSlot kate_ring[KATE_RING];
  uint32_t kate_next = 0;
  std::map<uint32_t, DomainConsts> consts;
  std::map<uint64_t, fp_words*> t_evals;  // key = k << 32 | ext_k
  // page-locked host memory mapped into the device: small results the host waits for anyway (evaluations, a remainder, a
  // verdict) are written there by the kernel that produces them -- no copy kernel, no second wait
  static constexpr size_t MAIL_BYTES = 4096;
  uint8_t* h_mail = nullptr;
  uint8_t* d_mail = nullptr;
  // sg_lookup_permute_small_async_dev: two work spaces per caller stream; the write pass of one call zeroes the other's
  // histograms for the next call (no memset launches)
  struct LookupWork {
    DevBuf<uint32_t> buf;
    uint32_t next = 0;
  };
  std::map<hipStream_t, LookupWork> lookup_work;
};

// What every lane shares: the device index, the SRS cache (read-only after upload / precompute) and the runtime
// parameters.  Guarded by its own short mutex (never held across device work).
struct Shared {
  std::mutex mu;
  int device = -1;
  std::map<uint64_t, Srs> srs;
  uint64_t next_handle = 1;
  std::vector<std::pair<std::string, int>> params;  // sg_set_param history, replayed on every new lane
};

// Concurrency model.  The library keeps g_lane_count (default 4, at most kLanes) independent contexts ("lanes"), each with its own streams, MSM
// engines, NTT plans, staging and scratch buffers.  A call takes ONE lane for its whole duration (lane 0 when it is
// free, so a single-threaded caller always works in the same warm work space) and touches nothing of the others:
// calls from different host threads -- halo2 reaches best_multiexp / best_fft from rayon iterators; the batch driver
// keeps several proofs in flight -- run side by side on the device, host tails included.  Asynchronous `_dev` calls
// leave work behind only in buffers keyed by the caller's stream, so a lane can be handed to the next caller as soon
// as the call returns.  The lock is per lane and re-entrant for the owning thread (entry points that stage host
// buffers and then call their `_dev` form keep the lane in between).
Shared g_sh;
static constexpr int kLanes = 8;          // upper bound; g_lane_count of them are handed out (sg_set_param "lanes")
std::atomic<int> g_lane_count{4};
struct Lane {
  std::mutex mu;
  Context* ctx = nullptr;
};
Lane g_lanes[kLanes];
std::atomic<unsigned> g_rr{0};
thread_local Context* g_ctx = nullptr;   // the lane this thread holds (valid inside LOCKED_CTX scopes only)
thread_local Lane* g_held = nullptr;
thread_local int g_depth = 0;

int apply_param(Context& c, const std::string& s, int value);

// The main streams of the lanes in use are created together, before any of the library's other streams, so that they land
// on different hardware queues: two streams on ONE queue run their kernels one after the other whatever the priorities,
// and the next MSM's sort then sits behind the current accumulation instead of under it
// (profiles/r03_sweeps/persistent_accumulate.txt).  No more of them than lanes: every stream is a queue the firmware has
// to schedule, and idle ones cost too (a k = 17 proof: 5.83 ms with four, 6.27 ms with eight; raising HIP's number of
// hardware queues, GPU_MAX_HW_QUEUES = 8 / 16, takes a batch of sixteen proofs in flight from 237 to 140 / 52 proofs/s).
hipStream_t g_lane_main[kLanes] = {};   // guarded by g_sh.mu
hipError_t make_lane_streams(int count) {
  for (int i = 0; i < count && i < kLanes; i++) {
    if (g_lane_main[i]) continue;
    hipError_t e = hipStreamCreateWithFlags(&g_lane_main[i], hipStreamNonBlocking);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
int make_context(int device, int lane_index, Context** out) {
  Context* c = new Context();
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lk(g_sh.mu);
    e = make_lane_streams(std::max(g_lane_count.load(), lane_index + 1));
    if (e == hipSuccess) c->stream = g_lane_main[lane_index];
  }
  if (e == hipSuccess) e = c->ntt.init();
  if (e == hipSuccess) e = c->msm.init();
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->bstream[0], hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->bstream[1], hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc(&c->d_consts, sizeof(DomainConsts));
  if (e != hipSuccess) {
    delete c;
    return hip_fail("sg_init", e);
  }
  c->device = device;
  *out = c;
  return SG_OK;
}

void destroy_context(Context* c) {
  for (auto& kv : c->t_evals) (void)hipFree(kv.second);
  for (auto& kv : c->coset_tables) {
    (void)hipFree(kv.second.fwd);
    (void)hipFree(kv.second.inv);
  }
  for (auto& slot : c->blob_ring) {
    if (slot.ev) (void)hipEventDestroy(slot.ev);
    if (slot.host) (void)hipHostFree(slot.host);
    if (slot.dev) (void)hipFree(slot.dev);
  }
  c->ntt.clear();
  c->msm.release();
  c->msm_b.release();
  c->witness.release();
  for (auto& bs : c->bstream) {
    if (bs) (void)hipStreamDestroy(bs);
  }
  for (auto& ts : c->tstream) {
    if (ts) (void)hipStreamDestroy(ts);
  }
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  c->stage_a.release();
  c->stage_b.release();
  c->scratch.release();
  for (auto& kv : c->ntt_scratch) kv.second.release();
  for (auto& kv : c->stream_scratch) kv.second.release();
  for (auto& kv : c->lookup_work) kv.second.buf.release();
  if (c->h_mail) (void)hipHostFree(c->h_mail);
  for (auto& sl : c->kate_ring) {
    if (sl.ev) (void)hipEventDestroy(sl.ev);
    if (sl.host) (void)hipHostFree(sl.host);
    if (sl.dev) (void)hipFree(sl.dev);
  }
  if (c->d_consts) (void)hipFree(c->d_consts);
  c->stream = nullptr;   // one of g_lane_main: destroyed with the others at sg_shutdown
  delete c;
}

// take a lane for the calling thread (g_ctx / g_held): lane 0 if free, else the first free one, else wait
int acquire_lane() {
  int device;
  std::vector<std::pair<std::string, int>> params;
  {
    std::lock_guard<std::mutex> lk(g_sh.mu);
    if (g_sh.device < 0) {
      // lazy default initialisation on device 0 keeps the seam a pure function call, like best_multiexp / best_fft
      int n = 0;
      if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(SG_ERR_NO_DEVICE, "no HIP device visible");
      g_sh.device = 0;
    }
    device = g_sh.device;
    params = g_sh.params;
  }
  Lane* lane = nullptr;
  const int lanes = g_lane_count.load();
  for (int i = 0; i < lanes && !lane; i++)
    if (g_lanes[i].mu.try_lock()) lane = &g_lanes[i];
  if (!lane) {   // more concurrent callers than lanes: wait for one (round-robin), for the whole of its current call
    lane = &g_lanes[g_rr.fetch_add(1) % (unsigned)lanes];
    lane->mu.lock();
  }
  if (!lane->ctx) {
    int rc = make_context(device, (int)(lane - g_lanes), &lane->ctx);
    if (rc != SG_OK) {
      lane->mu.unlock();
      return rc;
    }
    for (auto& kv : params) (void)apply_param(*lane->ctx, kv.first, kv.second);
  }
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) {
    lane->mu.unlock();
    return hip_fail("hipSetDevice", e);
  }
  g_held = lane;
  g_ctx = lane->ctx;
  return SG_OK;
}
struct LaneHold {
  int rc = SG_OK;
  LaneHold() {
    if (g_depth > 0) {
      g_depth++;
      return;
    }
    rc = acquire_lane();
    if (rc == SG_OK) g_depth = 1;
  }
  ~LaneHold() {
    if (rc != SG_OK) return;
    if (--g_depth == 0) {
      Lane* l = g_held;
      g_held = nullptr;
      g_ctx = nullptr;
      l->mu.unlock();
    }
  }
  LaneHold(const LaneHold&) = delete;
  LaneHold& operator=(const LaneHold&) = delete;
};

// the SRS behind a handle: a COPY of the entry (device pointers, window tables' descriptors), taken under the lock, so
// that a concurrent sg_srs_precompute / sg_srs_free on another lane never changes it under a reader.  The device
// memory it names stays valid while the handle does (retired tables are kept until sg_collect_retired / sg_shutdown).
bool find_srs(uint64_t handle, Srs* out) {
  std::lock_guard<std::mutex> lk(g_sh.mu);
  auto it = g_sh.srs.find(handle);
  if (it == g_sh.srs.end()) return false;
  *out = it->second;
  return true;
}

int get_consts(uint32_t k, const DomainConsts** out) {
  Context& c = *g_ctx;
  auto it = c.consts.find(k);
  if (it == c.consts.end()) {
    domain_kernel<<<1, 1, 0, c.stream>>>(k, c.d_consts);
    DomainConsts h;
    CHECK_HIP(hipMemcpyAsync(&h, c.d_consts, sizeof h, hipMemcpyDeviceToHost, c.stream), "domain constants");
    CHECK_HIP(host_wait_stream(c.stream), "domain constants");
    it = c.consts.emplace(k, h).first;
  }
  *out = &it->second;
  return SG_OK;
}

// _dev entry points run on exactly the stream they are given (NULL = HIP's default stream,
// which is what torch.cuda.current_stream() is unless the caller switched streams)
hipStream_t pick_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// order the library's own stream work (plan/twiddle generation) before a caller stream
int sync_own_stream_into(hipStream_t s) {
  if (s == g_ctx->stream) return SG_OK;
  CHECK_HIP(host_wait_stream(g_ctx->stream), "stream sync");
  return SG_OK;
}

int ntt_dev(const fp_words* in, size_t in_len, fp_words* out, uint32_t log_n, const words8& omega,
            const words8* scale, const words8* pre3, const words8* post3, hipStream_t s) {
  Context& c = *g_ctx;
  if (log_n > 28) return fail(SG_ERR_INVALID, "log_n exceeds the 2-adicity (28) of BN254 Fr");
  fp_words* scratch = nullptr;
  if (in == out && log_n > c.ntt.config().max_single_log) {
    DevBuf<uint8_t>& buf = c.ntt_scratch[s];
    hipError_t e = buf.reserve((size_t)32 << log_n);
    if (e != hipSuccess) return hip_fail("ntt scratch", e);
    scratch = reinterpret_cast<fp_words*>(buf.p);
  }
  // plans are generated on the same stream the transform runs on
  hipError_t e = c.ntt.transform(in, in_len, out, scratch, log_n, omega, scale, pre3, post3, s);
  if (e != hipSuccess) return hip_fail("ntt", e);
  return SG_OK;
}

hipError_t mailbox(uint8_t** host, uint8_t** dev) {
  Context& c = *g_ctx;
  if (!c.h_mail) {
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c.h_mail), Context::MAIL_BYTES, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) return e;
    e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c.d_mail), c.h_mail, 0);
    if (e != hipSuccess) return e;
  }
  *host = c.h_mail;
  *dev = c.d_mail;
  return hipSuccess;
}

hipError_t scratch_for(hipStream_t s, int slot, size_t bytes, uint8_t** out) {
  DevBuf<uint8_t>& buf = g_ctx->stream_scratch[std::make_pair(s, slot)];
  hipError_t e = buf.reserve(bytes);  // growing frees the old buffer, which waits for the device: safe
  *out = buf.p;
  return e;
}

int upload(DevBuf<uint8_t>& buf, const uint8_t* host, size_t bytes, hipStream_t s) {
  hipError_t e = buf.reserve(bytes ? bytes : 1);
  if (e != hipSuccess) return hip_fail("staging buffer", e);
  if (bytes) CHECK_HIP(hipMemcpyAsync(buf.p, host, bytes, hipMemcpyHostToDevice, s), "H2D copy");
  return SG_OK;
}
int download(uint8_t* host, const void* dev, size_t bytes, hipStream_t s) {
  CHECK_HIP(host_copy_d2h(host, dev, bytes, s), "D2H copy");
  return SG_OK;
}

#define LOCKED_CTX()     \
  LaneHold _hold;        \
  if (_hold.rc != SG_OK) return _hold.rc;
#define TRY(x)                  \
  do {                          \
    int _rc = (x);              \
    if (_rc != SG_OK) return _rc; \
  } while (0)

int apply_param(Context& c, const std::string& s, int value) {
  if (s == "msm.window_bits") c.msm.config().window_bits = c.msm_b.config().window_bits = (uint32_t)value;
  else if (s == "msm.log_seg") c.msm.config().log_seg = c.msm_b.config().log_seg = (uint32_t)std::min(12, value);
  else if (s == "msm.log_fuse_entries") {   // one value for generic and fixed-base jobs; 0: the built-in defaults (what sg_get_param reports for a parameter never set)
    const uint32_t v = (uint32_t)std::max(16, std::min(30, value));
    c.msm.config().log_fuse_entries = c.msm_b.config().log_fuse_entries = value == 0 ? MSM_LOG_FUSE_ENTRIES_GENERIC : v;
    c.msm.config().log_fuse_entries_fixed = c.msm_b.config().log_fuse_entries_fixed = value == 0 ? MSM_LOG_FUSE_ENTRIES_FIXED : v;
  }
  else if (s == "msm.red_threads") { uint32_t v = value <= 64 ? 64 : value <= 128 ? 128 : 256; c.msm.config().red_threads = c.msm_b.config().red_threads = v; }
  else if (s == "msm.log_scatter_rounds") c.msm.config().log_scatter_rounds = c.msm_b.config().log_scatter_rounds = (uint32_t)std::min(6, std::max(0, value));
  else if (s == "msm.two_pass") c.msm.config().two_pass = c.msm_b.config().two_pass = (uint32_t)std::min(2, std::max(0, value));
  else if (s == "side_prio") {   // device-wide, not per lane: wave priority 3 for every kernel but msm_accumulate (side_prio.cuh)
    const uint32_t on = value ? 1u : 0u;
    hipError_t e = msm_set_side_prio(on);
    if (e == hipSuccess) e = ntt_set_side_prio(on);
    if (e == hipSuccess) e = poly_set_side_prio(on);
    if (e == hipSuccess) e = quotient_set_side_prio(on);
    if (e == hipSuccess) e = gates_set_side_prio(on);
    if (e == hipSuccess) e = numerator_set_side_prio(on);
    if (e == hipSuccess) e = witness_set_side_prio(on);
    if (e == hipSuccess) e = abi_set_side_prio(on);
    if (e != hipSuccess) return hip_fail("side_prio", e);
  }
  else if (s == "msm.fused_frontend") c.msm.config().fused_frontend = c.msm_b.config().fused_frontend = (uint32_t)std::min(2, std::max(0, value));
  else if (s == "msm.acc_trace") c.msm.config().acc_trace = c.msm_b.config().acc_trace = value ? 1u : 0u;
  else if (s == "msm.acc_chain") c.msm.config().acc_chain = c.msm_b.config().acc_chain = value ? 1u : 0u;
  else if (s == "msm.red_lean") c.msm.config().red_lean = c.msm_b.config().red_lean = (uint32_t)std::max(0, std::min(2, value));
  else if (s == "msm.acc_waves_fixed") c.msm.config().acc_waves_fixed = c.msm_b.config().acc_waves_fixed = (uint32_t)std::max(0, std::min(8, value));
  else if (s == "msm.acc_waves") c.msm.config().acc_waves = c.msm_b.config().acc_waves = (uint32_t)std::max(0, std::min(8, value));
  else if (s == "msm.acc_threads") c.msm.config().acc_threads = c.msm_b.config().acc_threads = (value == 64 || value == 128 || value == 256) ? (uint32_t)value : 0u;
  else if (s == "msm.merge_quad_tasks") c.msm.config().merge_quad_tasks = c.msm_b.config().merge_quad_tasks = (uint32_t)std::max(0, value);
  else if (s == "msm.red2d_max_sets") c.msm.config().red2d_max_sets = c.msm_b.config().red2d_max_sets = (uint32_t)std::min(32, std::max(0, value));
  else if (s == "msm.red2d_fold") c.msm.config().red2d_fold = c.msm_b.config().red2d_fold = (uint32_t)std::min(256, std::max(1, value));
  else if (s == "msm.red2d_prefold") c.msm.config().red2d_prefold = c.msm_b.config().red2d_prefold = value ? 1u : 0u;
  else if (s == "msm.prefold_quad_buckets") c.msm.config().prefold_quad_buckets = c.msm_b.config().prefold_quad_buckets = (uint32_t)std::max(0, value);
  else if (s == "msm.red2d") c.msm.config().red2d = c.msm_b.config().red2d = (uint32_t)std::min(2, std::max(0, value));
  else if (s == "msm.quad") c.msm.config().quad = c.msm_b.config().quad = (uint32_t)std::min(2, std::max(0, value));
  else if (s == "msm.log_red_chunk") c.msm.config().log_red_chunk = c.msm_b.config().log_red_chunk = (uint32_t)std::min(8, value);
  else if (s == "ntt.tile_log") c.ntt.config().tile_log = (uint32_t)std::max(6, std::min(12, value));
  else if (s == "ntt.threads") c.ntt.config().threads = (uint32_t)std::max(64, std::min(1024, value));
  else if (s == "ntt.big_tile_log") c.ntt.config().big_tile_log = value ? (uint32_t)std::max(6, std::min(12, value)) : 0u;   // 0: one shape for all
  else if (s == "ntt.big_threads") c.ntt.config().big_threads = (uint32_t)std::max(64, std::min(1024, value));
  else if (s == "ntt.batch_min") c.ntt.config().batch_min = (uint32_t)std::max(1, value);
  else if (s == "ntt.big_log") c.ntt.config().big_log = (uint32_t)std::max(1, value);
  else if (s == "ntt.radix4") c.ntt.config().radix4 = value == 1 ? 2u : value == 2 ? 0u : 1u;   // 0: by size (default), 1: always, 2: never
  else if (s == "ntt.max_single_log") { c.ntt.config().max_single_log = (uint32_t)std::max(1, std::min(12, value)); c.ntt.clear(); }
  else if (s == "ntt.max_multi_log") { c.ntt.config().max_multi_log = (uint32_t)std::max(4, std::min(12, value)); c.ntt.clear(); }
  else return fail(SG_ERR_INVALID, "sg_set_param: unknown parameter");
  return SG_OK;
}

}  // namespace

extern "C" {

int sg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
const char* sg_version(void) { return "summa_gpu 0.1.0 gfx950"; }
int sg_device(void) {
  std::lock_guard<std::mutex> lk(g_sh.mu);
  return g_sh.device;
}
int sg_bind_thread(void) {
  const int device = sg_device();
  if (device < 0) return SG_OK;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return hip_fail("sg_bind_thread", e);
  return SG_OK;
}
const char* sg_last_error(void) { return g_err; }

int sg_init(int device) {
  {
    std::lock_guard<std::mutex> lk(g_sh.mu);
    if (g_sh.device == device) return SG_OK;
    if (g_sh.device >= 0) return fail(SG_ERR_INVALID, "sg_init: already bound to another device (one process per GPU)");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(SG_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(SG_ERR_INVALID, "sg_init: device index out of range");
    g_sh.device = device;
  }
  LOCKED_CTX();   // creates lane 0: streams, plans' home, MSM engine attributes
  return SG_OK;
}

// host ranges page-locked for the library (sg_host_register, below)
struct HostRange {
  uintptr_t base;
  size_t bytes;
};
static std::mutex g_host_mu;
static std::vector<HostRange> g_host_ranges;
void sg_shutdown(void) {
  if (g_depth > 0) return;   // never from inside a call
  for (auto& l : g_lanes) l.mu.lock();
  int device;
  {
    std::lock_guard<std::mutex> lk(g_sh.mu);
    device = g_sh.device;
  }
  if (device >= 0) {
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    for (auto& l : g_lanes) {
      if (l.ctx) destroy_context(l.ctx);
      l.ctx = nullptr;
    }
    {
      std::lock_guard<std::mutex> lk(g_sh.mu);
      for (auto& st : g_lane_main) {
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
      }
    }
    {
      std::lock_guard<std::mutex> lk(g_host_mu);
      for (const HostRange& r : g_host_ranges) (void)hipHostUnregister(reinterpret_cast<void*>(r.base));
      g_host_ranges.clear();
    }
    std::lock_guard<std::mutex> lk(g_sh.mu);
    for (auto& kv : g_sh.srs) {
      (void)hipFree(kv.second.g);
      (void)hipFree(kv.second.g_lagrange);
      if (kv.second.lagrange_prefix) (void)hipFree(kv.second.lagrange_prefix);
      for (auto& t : kv.second.tab)
        if (t.table) (void)hipFree(t.table);
    }
    g_sh.srs.clear();
    g_sh.params.clear();
    g_sh.device = -1;
    retired_device_memory_collect();
    summa::prover::release_orphans();   // what the prover sessions of ended threads left behind
  }
  for (auto& l : g_lanes) l.mu.unlock();
}

// Returns the device memory the library has outgrown since it started (work spaces that were reallocated larger, window
// tables replaced by sg_srs_precompute).  It waits for the device: call it when no other call is in flight.
// ---- host memory the caller has page-locked for the library (sg_host_register): the host-pointer entry points move such
// ranges by direct DMA at link speed, asynchronously; anything else goes through the runtime's pageable path
[[maybe_unused]] static bool host_registered(const void* p, size_t bytes) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  std::lock_guard<std::mutex> lk(g_host_mu);
  for (const HostRange& r : g_host_ranges)
    if (a >= r.base && a + bytes <= r.base + r.bytes) return true;
  return false;
}
int sg_host_register(void* host, size_t bytes) {
  if (!host || !bytes) return fail(SG_ERR_INVALID, "sg_host_register: null or empty range");
  {
    LOCKED_CTX();   // a device is bound (the registration is made for it)
    const uintptr_t a = reinterpret_cast<uintptr_t>(host);
    {
      std::lock_guard<std::mutex> lk(g_host_mu);
      for (const HostRange& r : g_host_ranges)
        if (a < r.base + r.bytes && r.base < a + bytes) return fail(SG_ERR_INVALID, "sg_host_register: overlaps a registered range");
    }
    CHECK_HIP(hipHostRegister(host, bytes, hipHostRegisterDefault), "hipHostRegister");
    std::lock_guard<std::mutex> lk(g_host_mu);
    g_host_ranges.push_back({a, bytes});
  }
  return SG_OK;
}
int sg_host_unregister(void* host) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(host);
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    auto it = std::find_if(g_host_ranges.begin(), g_host_ranges.end(), [&](const HostRange& r) { return r.base == a; });
    if (it == g_host_ranges.end()) return fail(SG_ERR_INVALID, "sg_host_unregister: not the start of a registered range");
    g_host_ranges.erase(it);
  }
  CHECK_HIP(hipHostUnregister(host), "hipHostUnregister");
  return SG_OK;
}

int sg_stream_wait(void* stream) {
  const hipError_t e = host_wait_stream(static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail("sg_stream_wait", e);
  return SG_OK;
}

int sg_collect_retired(void) {
  if (g_depth > 0) return fail(SG_ERR_INVALID, "sg_collect_retired: not from inside a call");
  int device;
  {
    std::lock_guard<std::mutex> lk(g_sh.mu);
    device = g_sh.device;
  }
  if (device < 0) return SG_OK;
  for (auto& l : g_lanes) l.mu.lock();       // index order, no lane of our own held: no call is in flight meanwhile
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) {
    retired_device_memory_collect();
    summa::prover::release_orphans();
    // the lanes' own work spaces too (MSM buckets and sort buffers, NTT plans and twiddle tables, staging, scratch): which
    // lane met the largest job of the last workload is a matter of timing, so "what the library holds" would otherwise
    // creep up to lanes x the largest work space over a long-lived process.  They are rebuilt by the next call that
    // needs them (milliseconds); the lanes' main streams, the SRS cache and the proving keys stay.
    for (auto& l : g_lanes) {
      if (l.ctx) destroy_context(l.ctx);
      l.ctx = nullptr;
    }
  }
  for (auto& l : g_lanes) l.mu.unlock();
  if (e != hipSuccess) return hip_fail("sg_collect_retired", e);
  return SG_OK;
}

// ------------------------------------------------------------------ MSM
int sg_msm_g1_dev_timed(const void* d_scalars, const void* d_bases, size_t n, void* stream, uint8_t out_affine[64],
                        sg_msm_timings* timings) {
  if (!out_affine || (n && (!d_scalars || !d_bases))) return fail(SG_ERR_INVALID, "sg_msm_g1: null argument");
  LOCKED_CTX();
  MsmTimings tm;
  // on the lane's own stream, after everything the caller has enqueued on his: the call returns the point, so nothing of
  // it is left on any stream afterwards, and the lanes' streams sit on different hardware queues (make_lane_streams) --
  // which the streams of callers on different threads may or may not
  hipStream_t st = pick_stream(stream);
  if (st != g_ctx->stream) {
    // (an idle caller stream needs no edge -- and a marker on it would queue behind whatever shares ITS hardware queue,
    // another lane's accumulation for instance)
    if (hipStreamQuery(st) != hipSuccess) {
      CHECK_HIP(hipEventRecord(g_ctx->ev_in, st), "event");
      CHECK_HIP(hipStreamWaitEvent(g_ctx->stream, g_ctx->ev_in, 0), "wait");
    }
    st = g_ctx->stream;
  }
  hipError_t e = g_ctx->msm.run(static_cast<const fp_words*>(d_scalars), static_cast<const g1_affine_mem*>(d_bases), n,
                                st, out_affine, timings ? &tm : nullptr);
  if (e != hipSuccess) return hip_fail("msm", e);
  if (timings) {
    timings->digits_ms = tm.digits_ms; timings->sort_ms = tm.sort_ms; timings->accumulate_ms = tm.accumulate_ms;
    timings->reduce_ms = tm.reduce_ms; timings->total_ms = tm.total_ms; timings->window_bits = tm.window_bits;
    timings->windows = tm.windows; timings->tasks = tm.tasks; timings->max_bucket = tm.max_bucket;
    timings->accumulate_threads = tm.accumulate_threads;
    timings->order_ms = tm.order_ms;
  }
  return SG_OK;
}
int sg_msm_g1_dev(const void* d_scalars, const void* d_bases, size_t n, void* stream, uint8_t out_affine[64]) {
  return sg_msm_g1_dev_timed(d_scalars, d_bases, n, stream, out_affine, nullptr);
}
// The host-pointer MSM entry points (sg_msm_g1: scalars and bases in host memory; sg_commit: scalars in host memory, bases
// resident) pay the link -- 96 or 32 bytes per pair at the 56 GB/s this platform reaches from pageable memory just as from
// page-locked memory -- before the last addition can run, and a lone MSM is a third latency chains (sort front end, bucket
// reduction, host tail) besides.  Large inputs are therefore cut into K chunks that run as K jobs on the lane's two
// engines (two streams) while a third stream carries the copies: chunk i's job runs while chunk i + 1 is still travelling,
// and one job's latency chains run under the other's accumulation.  The K partial points are added on the host.
//   [S0 B0] front(0) { back(i) [S i+1 B i+1] finish(i-1) front(i+1) } ... finish, sum
// -- chunk i + 1 crosses the link while chunk i's accumulation runs.  K: "msm.host_chunks" (0 = by size: 2 from 2^18 pairs; more
// chunks lose: every job brings its own latency chains, and small kernels beside an accumulation run slowly).
static constexpr size_t MSM_HOST_SPLIT_MIN = (size_t)1 << 18;
static std::atomic<int> g_host_chunks{0};
static std::atomic<int> g_msm_tiny_max{(int)MSM_TINY_MAX};   // "msm.tiny_max": host-pointer MSMs of at most this many points run as one launch
static int msm_host_chunked(const uint8_t* scalars, const uint8_t* bases_host, const g1_affine_mem* d_bases_resident, size_t n,
                            uint8_t out_affine[64]) {
  Context& c = *g_ctx;
  hipError_t e = c.stage_a.reserve(n ? n * 32 : 1);
  if (e == hipSuccess && bases_host) e = c.stage_b.reserve(n ? n * 64 : 1);
  if (e != hipSuccess) return hip_fail("staging buffer", e);
  const fp_words* d_s = reinterpret_cast<const fp_words*>(c.stage_a.p);
  const g1_affine_mem* d_b = bases_host ? reinterpret_cast<const g1_affine_mem*>(c.stage_b.p) : d_bases_resident;
  uint32_t K = (uint32_t)g_host_chunks.load();
  if (K == 0) K = n < MSM_HOST_SPLIT_MIN ? 1u : 2u;   // measured at 2^20 (profiles/r04_sweeps/host_chunks.txt): 2 is the best for both entry points
  K = std::min<uint32_t>(K, 8u);
  if (n < 2 * (size_t)K) K = 1;
  if (K == 1) {
    if (n) {
      CHECK_HIP(hipMemcpyAsync(c.stage_a.p, scalars, n * 32, hipMemcpyHostToDevice, c.stream), "H2D copy");
      if (bases_host) CHECK_HIP(hipMemcpyAsync(c.stage_b.p, bases_host, n * 64, hipMemcpyHostToDevice, c.stream), "H2D copy");
    }
    e = c.msm.run(d_s, d_b, n, c.stream, out_affine, nullptr);
    if (e != hipSuccess) return hip_fail("msm", e);
    return SG_OK;
  }
  MsmEngine* eng[2] = {&c.msm, &c.msm_b};
  hipStream_t st[2] = {c.stream, c.bstream[0]}, copy = c.bstream[1];
  std::vector<size_t> lo(K + 1);
  for (uint32_t i = 0; i <= K; i++) lo[i] = n * i / K;
  std::vector<uint8_t> part(64 * (size_t)K, 0);
  std::vector<hipEvent_t> ev_s(K, nullptr), ev_b(K, nullptr);
  struct Events {
    std::vector<hipEvent_t>&a, &b;
    ~Events() {
      for (auto v : {&a, &b})
        for (hipEvent_t x : *v)
          if (x) (void)hipEventDestroy(x);
    }
  } events_guard{ev_s, ev_b};
  for (uint32_t i = 0; i < K; i++) {
    CHECK_HIP(hipEventCreateWithFlags(&ev_s[i], hipEventDisableTiming), "event");
    if (bases_host) CHECK_HIP(hipEventCreateWithFlags(&ev_b[i], hipEventDisableTiming), "event");
  }
  // the staging buffers may still be read by earlier work of the lane's stream: the other two streams start behind it
  CHECK_HIP(hipEventRecord(c.ev_in, c.stream), "event");
  CHECK_HIP(hipStreamWaitEvent(st[1], c.ev_in, 0), "stream wait");
  CHECK_HIP(hipStreamWaitEvent(copy, c.ev_in, 0), "stream wait");
  hipError_t err = hipSuccess;          // the first failure; from the first front on, every open job is still closed in order
  auto copy_scalars = [&](uint32_t i) {
    if (err != hipSuccess) return;
    err = hipMemcpyAsync(c.stage_a.p + lo[i] * 32, scalars + lo[i] * 32, (lo[i + 1] - lo[i]) * 32, hipMemcpyHostToDevice, copy);
    if (err == hipSuccess) err = hipEventRecord(ev_s[i], copy);
  };
  auto copy_bases = [&](uint32_t i) {
    if (err != hipSuccess || !bases_host) return;
    err = hipMemcpyAsync(c.stage_b.p + lo[i] * 64, bases_host + lo[i] * 64, (lo[i + 1] - lo[i]) * 64, hipMemcpyHostToDevice, copy);
    if (err == hipSuccess) err = hipEventRecord(ev_b[i], copy);
  };
  int state[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // per chunk: 0 nothing, 1 front enqueued, 2 back enqueued, 3 finished
  auto front = [&](uint32_t i) {
    if (err != hipSuccess) return;
    err = hipStreamWaitEvent(st[i & 1], ev_s[i], 0);
    if (err == hipSuccess) err = eng[i & 1]->enqueue_front(d_s + lo[i], d_b + lo[i], lo[i + 1] - lo[i], st[i & 1], part.data() + 64 * i, nullptr);
    if (err == hipSuccess) state[i] = 1;
  };
  auto back = [&](uint32_t i) {
    if (state[i] != 1) return;
    hipError_t e2 = bases_host ? hipStreamWaitEvent(st[i & 1], ev_b[i], 0) : hipSuccess;
    const hipError_t e3 = eng[i & 1]->enqueue_back();    // (always: an engine left with an open job would poison the lane's next call)
    state[i] = e3 == hipSuccess ? 2 : 3;
    if (err == hipSuccess) err = e2 != hipSuccess ? e2 : e3;
  };
  auto finish = [&](uint32_t i) {
    if (state[i] != 2) return;
    const hipError_t e2 = eng[i & 1]->finish();
    state[i] = 3;
    if (err == hipSuccess) err = e2;
  };
  // every job of the call is "one of several in flight" from the start: a first accumulation launched at three waves per SIMD
  // (a job that believes it has the device to itself) would leave the second job's front end no registers to run in
  struct InFlight {
    InFlight() { msm_hold_in_flight(true); }
    ~InFlight() { msm_hold_in_flight(false); }
  } in_flight_guard;
  copy_scalars(0);
  copy_bases(0);
  front(0);
  for (uint32_t i = 0; i < K; i++) {
    back(i);                              // (waits for chunk i's sort; then its accumulation is on the device ...)
    if (i + 1 < K) {
      copy_scalars(i + 1);                // ... and runs while the next chunk crosses the link (a copy from pageable memory blocks the host)
      copy_bases(i + 1);
      if (i >= 1) finish(i - 1);          // the engine chunk i + 1 runs on
      front(i + 1);
    }
  }
  for (uint32_t i = 0; i < K; i++) {       // whatever is still open (the last two jobs; everything after a failure)
    back(i);
    finish(i);
  }
  if (err != hipSuccess) return hip_fail("msm (host chunks)", err);
  return sg_g1_sum_affine(part.data(), K, out_affine);
}

int sg_msm_g1(const uint8_t* scalars, const uint8_t* bases, size_t n, uint8_t out_affine[64]) {
  if (!out_affine || (n && (!scalars || !bases))) return fail(SG_ERR_INVALID, "sg_msm_g1: null argument");
  LOCKED_CTX();
  if (n && n <= (size_t)g_msm_tiny_max.load()) {   // a handful of points (the verifier's 37): one launch, no staging (MsmEngine::run_tiny)
    const hipError_t e = g_ctx->msm.run_tiny(scalars, bases, n, g_ctx->stream, out_affine);
    if (e != hipSuccess) return hip_fail("msm (one launch)", e);
    return SG_OK;
  }
  return msm_host_chunked(scalars, bases, nullptr, n, out_affine);
}

// A batch of independent MSMs (the commitments of one prover phase): two engines on two
// streams, so that MSM i's latency-bound bucket reduction overlaps MSM i+1's sort/accumulate.
// batch driver shared by sg_msm_g1_batch_dev (d_bases given) and sg_commit_batch_dev (tab given: every
// MSM runs over the precomputed window table); caller holds the context lock
static int msm_batch_locked(const void* const* d_scalars, const void* const* d_bases, const FixedTable* tab,
                            const size_t* n, size_t count, void* stream, uint8_t* out_affine, const uint8_t* diff = nullptr) {
  Context& c = *g_ctx;
  MsmEngine* eng[2] = {&c.msm, &c.msm_b};
  for (int k = 0; k < 2; k++) {
    if (!c.tstream[k]) {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically lowest = highest priority
      CHECK_HIP(hipStreamCreateWithPriority(&c.tstream[k], hipStreamNonBlocking, hi), "priority stream");
    }
    eng[k]->set_tail_stream(c.tstream[k]);
  }
  struct Restore {
    MsmEngine** e;
    ~Restore() { e[0]->set_tail_stream(nullptr); e[1]->set_tail_stream(nullptr); }
  } restore{eng};
  // inputs are ordered on the caller's stream
  CHECK_HIP(hipEventRecord(c.ev_in, pick_stream(stream)), "event");
  for (auto& bs : c.bstream) CHECK_HIP(hipStreamWaitEvent(bs, c.ev_in, 0), "stream wait");
  // consecutive MSMs of equal length are fused into one job (all kernels span the whole
  // group); groups alternate between the two engines
  struct Group { size_t first, count; };
  std::vector<Group> groups;
  for (size_t i = 0; i < count;) {
    size_t lim = tab ? eng[0]->max_fused_fixed(*tab, n[i]) : eng[0]->max_fused(n[i]), g = 1;
    while (i + g < count && n[i + g] == n[i] && g < lim) g++;
    groups.push_back({i, g});
    i += g;
  }
  hipError_t e = hipSuccess;
  for (size_t gi = 0; gi < groups.size() && e == hipSuccess; gi++) {
    const int k = (int)(gi & 1);
    if (gi >= 2) {
      e = eng[k]->finish();
      if (e != hipSuccess) break;
    }
    const Group& g = groups[gi];
    if (tab) {  // d_bases then holds one window table per MSM (all with tab's plan)
      uint64_t diff_mask = 0;
      for (size_t m = 0; diff && m < g.count; m++) diff_mask |= (uint64_t)(diff[g.first + m] ? 1 : 0) << m;
      e = eng[k]->enqueue_front_fixed(reinterpret_cast<const fp_words* const*>(d_scalars + g.first), *tab, g.count,
                                      n[g.first], c.bstream[k], out_affine + 64 * g.first, nullptr,
                                      reinterpret_cast<const g1_affine_mem* const*>(d_bases + g.first), diff_mask);
    }
    else
      e = eng[k]->enqueue_front_fused(reinterpret_cast<const fp_words* const*>(d_scalars + g.first),
                                      reinterpret_cast<const g1_affine_mem* const*>(d_bases + g.first), g.count,
                                      n[g.first], c.bstream[k], out_affine + 64 * g.first, nullptr);
    if (e == hipSuccess) e = eng[k]->enqueue_back();
  }
  for (size_t gi = (groups.size() >= 2 ? groups.size() - 2 : 0); gi < groups.size() && e == hipSuccess; gi++)
    e = eng[gi & 1]->finish();
  if (e != hipSuccess) {
    (void)hipDeviceSynchronize();
    return hip_fail("msm batch", e);
  }
  return SG_OK;
}
int sg_msm_g1_batch_dev(const void* const* d_scalars, const void* const* d_bases, const size_t* n, size_t count,
                        void* stream, uint8_t* out_affine) {
  if (count && (!d_scalars || !d_bases || !n || !out_affine)) return fail(SG_ERR_INVALID, "sg_msm_g1_batch: null argument");
  for (size_t i = 0; i < count; i++) {
    if (n[i] && (!d_scalars[i] || !d_bases[i])) return fail(SG_ERR_INVALID, "sg_msm_g1_batch: null argument");
  }
  LOCKED_CTX();
  return msm_batch_locked(d_scalars, d_bases, nullptr, n, count, stream, out_affine);
}
int sg_msm_g1_batch(const uint8_t* const* scalars, const uint8_t* const* bases, const size_t* n, size_t count,
                    uint8_t* out_affine) {
  if (count && (!scalars || !bases || !n || !out_affine)) return fail(SG_ERR_INVALID, "sg_msm_g1_batch: null argument");
  std::vector<const void*> ds(count), db(count);
  LOCKED_CTX();   // held across staging AND the batch: the staging buffers are this lane's
  {
    size_t tot_s = 0, tot_b = 0;
    for (size_t i = 0; i < count; i++) { tot_s += n[i] * 32; tot_b += n[i] * 64; }
    hipError_t e = g_ctx->stage_a.reserve(tot_s + 64);
    if (e == hipSuccess) e = g_ctx->stage_b.reserve(tot_b + 64);
    if (e != hipSuccess) return hip_fail("staging buffer", e);
    size_t os = 0, ob = 0;
    for (size_t i = 0; i < count; i++) {
      if (n[i] && (!scalars[i] || !bases[i])) return fail(SG_ERR_INVALID, "sg_msm_g1_batch: null argument");
      if (n[i]) {
        CHECK_HIP(hipMemcpyAsync(g_ctx->stage_a.p + os, scalars[i], n[i] * 32, hipMemcpyHostToDevice, g_ctx->stream), "H2D copy");
        CHECK_HIP(hipMemcpyAsync(g_ctx->stage_b.p + ob, bases[i], n[i] * 64, hipMemcpyHostToDevice, g_ctx->stream), "H2D copy");
      }
      ds[i] = g_ctx->stage_a.p + os;
      db[i] = g_ctx->stage_b.p + ob;
      os += n[i] * 32;
      ob += n[i] * 64;
    }
  }
  return sg_msm_g1_batch_dev(ds.data(), db.data(), n, count, g_ctx->stream, out_affine);
}

// Sum of a handful of affine points on the host (combining the per-GPU partial results of a
// point-sharded MSM after the all_gather): a few Jacobian additions + one normalisation.
int sg_g1_sum_affine(const uint8_t* points, size_t n, uint8_t out_affine[64]) {
  if (!out_affine || (n && !points)) return fail(SG_ERR_INVALID, "sg_g1_sum_affine: null argument");
  if (n > 4096) return fail(SG_ERR_INVALID, "sg_g1_sum_affine: meant for a handful of points; use sg_msm_g1");
  using namespace sg::host;
  Jac acc = Jac::identity();
  for (size_t i = 0; i < n; i++) {
    Fq x, y;
    std::memcpy(x.v, points + 64 * i, 32);
    std::memcpy(y.v, points + 64 * i + 32, 32);
    if (x.is_zero() && y.is_zero()) continue;
    acc = jac_add(acc, Jac{x, y, Fq::one()});
  }
  jac_to_affine_bytes(acc, out_affine);
  return SG_OK;
}

int sg_srs_upload(uint32_t k, const uint8_t* g, const uint8_t* g_lagrange, uint64_t* handle_out) {
  if (!g || !g_lagrange || !handle_out || k > 28) return fail(SG_ERR_INVALID, "sg_srs_upload: bad argument");
  LOCKED_CTX();
  const size_t bytes = (size_t)64 << k;
  Srs s{k, nullptr, nullptr, {}};
  hipError_t e = hipMalloc(&s.g, bytes);
  if (e == hipSuccess) e = hipMalloc(&s.g_lagrange, bytes);
  if (e == hipSuccess) e = hipMemcpy(s.g, g, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(s.g_lagrange, g_lagrange, bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (s.g) (void)hipFree(s.g);
    if (s.g_lagrange) (void)hipFree(s.g_lagrange);
    return hip_fail("sg_srs_upload", e);
  }
  {
    std::lock_guard<std::mutex> lk(g_sh.mu);
    const uint64_t h = g_sh.next_handle++;
    g_sh.srs[h] = s;
    *handle_out = h;
  }
  return SG_OK;
}
// the same from device memory (e.g. the receive buffers of an RCCL broadcast): device-to-device copies on `stream`
int sg_srs_upload_dev(uint32_t k, const void* d_g, const void* d_g_lagrange, void* stream, uint64_t* handle_out) {
  if (!d_g || !d_g_lagrange || !handle_out || k > 28) return fail(SG_ERR_INVALID, "sg_srs_upload_dev: bad argument");
  LOCKED_CTX();
  const size_t bytes = (size_t)64 << k;
  hipStream_t st = pick_stream(stream);
  Srs s{k, nullptr, nullptr, {}};
  hipError_t e = hipMalloc(&s.g, bytes);
  if (e == hipSuccess) e = hipMalloc(&s.g_lagrange, bytes);
  if (e == hipSuccess) e = hipMemcpyAsync(s.g, d_g, bytes, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(s.g_lagrange, d_g_lagrange, bytes, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = host_wait_stream(st);   // the bases are read from other streams afterwards
  if (e != hipSuccess) {
    if (s.g) (void)hipFree(s.g);
    if (s.g_lagrange) (void)hipFree(s.g_lagrange);
    return hip_fail("sg_srs_upload_dev", e);
  }
  {
    std::lock_guard<std::mutex> lk(g_sh.mu);
    const uint64_t h = g_sh.next_handle++;
    g_sh.srs[h] = s;
    *handle_out = h;
  }
  return SG_OK;
}
// copies of the resident bases into caller-owned device buffers (2^k x 64 B each; either may be NULL)
int sg_srs_copy_dev(uint64_t handle, void* d_g_out, void* d_g_lagrange_out, void* stream) {
  LOCKED_CTX();
  Srs srs_v;
  if (!find_srs(handle, &srs_v)) return fail(SG_ERR_INVALID, "unknown SRS handle");
  const size_t bytes = (size_t)64 << srs_v.k;
  hipStream_t st = pick_stream(stream);
  if (d_g_out) CHECK_HIP(hipMemcpyAsync(d_g_out, srs_v.g, bytes, hipMemcpyDeviceToDevice, st), "sg_srs_copy_dev");
  if (d_g_lagrange_out) CHECK_HIP(hipMemcpyAsync(d_g_lagrange_out, srs_v.g_lagrange, bytes, hipMemcpyDeviceToDevice, st), "sg_srs_copy_dev");
  return SG_OK;
}
// `SerdeFormat::RawBytes` validation of ParamsKZG::read (halo2: from_raw_bytes rejects points off the curve; the
// `RawBytesUnchecked` format skips this): *bad_out = number of points of the resident SRS that fail y^2 = x^3 + 3
int sg_srs_check(uint64_t handle, uint64_t* bad_out) {
  if (!bad_out) return fail(SG_ERR_INVALID, "sg_srs_check: null argument");
  LOCKED_CTX();
  Srs srs_v;
  if (!find_srs(handle, &srs_v)) return fail(SG_ERR_INVALID, "unknown SRS handle");
  Srs* srs_p = &srs_v;
  uint8_t* cnt = nullptr;
  hipStream_t s = g_ctx->stream;
  hipError_t e = scratch_for(s, 7, 64, &cnt);
  uint32_t h[2] = {0, 0};
  const size_t n = (size_t)1 << srs_p->k;
  for (int b = 0; b < 2 && e == hipSuccess; b++) {
    e = g1_on_curve(b ? srs_p->g_lagrange : srs_p->g, n, reinterpret_cast<uint32_t*>(cnt), s);
    if (e == hipSuccess) e = host_copy_d2h(&h[b], cnt, 4, s);
  }
  if (e != hipSuccess) return hip_fail("sg_srs_check", e);
  *bad_out = (uint64_t)h[0] + h[1];
  return SG_OK;
}
int sg_srs_free(uint64_t handle) {
  LOCKED_CTX();
  Srs gone;
  {
    std::lock_guard<std::mutex> lk(g_sh.mu);
    auto it = g_sh.srs.find(handle);
    if (it == g_sh.srs.end()) return fail(SG_ERR_INVALID, "sg_srs_free: unknown handle");
    gone = it->second;
    g_sh.srs.erase(it);
  }
  (void)hipFree(gone.g);          // hipFree waits for the device: work in flight on these bases completes first
  (void)hipFree(gone.g_lagrange);
  if (gone.lagrange_prefix) (void)hipFree(gone.lagrange_prefix);
  for (auto& t : gone.tab)
    if (t.table) (void)hipFree(t.table);
  return SG_OK;
}
// Precompute the fixed-base window table of one basis: W x 2^k points, row w = 2^(offset_w) * basis.
// Later sg_commit* calls on this basis take the fixed-base path (same result bits).
int sg_srs_precompute(uint64_t handle, int basis, uint32_t window_bits) {
  if (basis < 0 || basis > 2) return fail(SG_ERR_INVALID, "sg_srs_precompute: bad basis");
  if (window_bits && (window_bits < 4 || window_bits > 16)) return fail(SG_ERR_INVALID, "sg_srs_precompute: window_bits in [4, 16]");
  LOCKED_CTX();
  Srs srs_v;
  if (!find_srs(handle, &srs_v)) return fail(SG_ERR_INVALID, "unknown SRS handle");
  Srs* srs_p = &srs_v;
  Srs& s = (*srs_p);   // a copy: the entry itself is updated under the lock once the table exists
  const size_t n = (size_t)1 << s.k;
  const uint32_t c = window_bits ? window_bits : fixed_window_bits_for(n);
  hipError_t e = hipSuccess;
  g1_affine_mem* new_prefix = nullptr;
  if (basis == 2 && !s.lagrange_prefix) {   // Q_i = L_0 + ... + L_i, once per SRS
    g1_affine_mem* q = nullptr;
    e = hipMalloc(&q, n * sizeof(g1_affine_mem));
    if (e == hipSuccess) e = g1_prefix_sums(s.g_lagrange, n, q, g_ctx->stream);
    if (e != hipSuccess) {
      if (q) (void)hipFree(q);
      return hip_fail("sg_srs_precompute: prefix sums", e);
    }
    s.lagrange_prefix = new_prefix = q;
  }
  FixedTable t;
  e = build_window_table(basis == 2 ? s.lagrange_prefix : basis ? s.g_lagrange : s.g, n, c, &t, g_ctx->stream);
  if (e == hipSuccess) e = host_wait_stream(g_ctx->stream);
  if (e != hipSuccess) {
    if (new_prefix) (void)hipFree(new_prefix);
    return hip_fail("sg_srs_precompute", e);
  }
  {
    std::lock_guard<std::mutex> lk(g_sh.mu);
    auto it = g_sh.srs.find(handle);
    if (it == g_sh.srs.end()) {   // freed by another thread meanwhile
      retire_device_memory(t.table);
      retire_device_memory(new_prefix);
      return fail(SG_ERR_INVALID, "sg_srs_precompute: the handle was freed during the call");
    }
    if (new_prefix) {
      if (it->second.lagrange_prefix) retire_device_memory(new_prefix);   // two concurrent precomputes: keep the first
      else it->second.lagrange_prefix = new_prefix;
    }
    retire_device_memory(it->second.tab[basis].table);   // commitments of other lanes may still be reading the old table
    it->second.tab[basis] = t;
  }
  return SG_OK;
}
int sg_srs_device_ptrs(uint64_t handle, const void** d_g, const void** d_g_lagrange, uint32_t* k) {
  LOCKED_CTX();
  Srs srs_v;
  if (!find_srs(handle, &srs_v)) return fail(SG_ERR_INVALID, "unknown SRS handle");
  Srs* srs_p = &srs_v;
  if (d_g) *d_g = (*srs_p).g;
  if (d_g_lagrange) *d_g_lagrange = (*srs_p).g_lagrange;
  if (k) *k = (*srs_p).k;
  return SG_OK;
}
// basis 2 = a Lagrange column taken in difference form (same commitment as basis 1): possible when the prefix-sum table
// exists and the column has the full 2^k rows
static bool diff_form_ready(const Srs& s, size_t n) { return s.tab[2].table != nullptr && n == ((size_t)1 << s.k); }
static hipError_t commit_run(const Srs& s, int basis, const fp_words* d_scalars, size_t n, hipStream_t stream,
                             uint8_t out_affine[64], MsmTimings* tm = nullptr) {
  MsmEngine& eng = g_ctx->msm;
  const bool diff = basis == 2 && diff_form_ready(s, n);
  if (basis == 2 && !diff) basis = 1;
  if (s.tab[basis].table && n) {
    const fp_words* sc[1] = {d_scalars};
    hipError_t e = eng.enqueue_front_fixed(sc, s.tab[basis], 1, n, stream, out_affine, tm, nullptr, diff ? 1u : 0u);
    if (e == hipSuccess) e = eng.enqueue_back();
    if (e == hipSuccess) e = eng.finish();
    return e;
  }
  return eng.run(d_scalars, basis ? s.g_lagrange : s.g, n, stream, out_affine, tm);
}
int sg_commit_dev_timed(uint64_t srs_handle, int basis, const void* d_scalars, size_t n, void* stream,
                        uint8_t out_affine[64], sg_msm_timings* timings) {
  if (!out_affine || (n && !d_scalars) || basis < 0 || basis > 2) return fail(SG_ERR_INVALID, "sg_commit: bad argument");
  LOCKED_CTX();
  Srs srs_v;
  if (!find_srs(srs_handle, &srs_v)) return fail(SG_ERR_INVALID, "unknown SRS handle");
  Srs* srs_p = &srs_v;
  if (n > ((size_t)1 << (*srs_p).k)) return fail(SG_ERR_INVALID, "sg_commit: polynomial longer than the SRS");
  MsmTimings tm;
  hipError_t e = commit_run((*srs_p), basis, static_cast<const fp_words*>(d_scalars), n, pick_stream(stream), out_affine,
                            timings ? &tm : nullptr);
  if (e != hipSuccess) return hip_fail("msm", e);
  if (timings) {
    timings->digits_ms = tm.digits_ms; timings->sort_ms = tm.sort_ms; timings->accumulate_ms = tm.accumulate_ms;
    timings->reduce_ms = tm.reduce_ms; timings->total_ms = tm.total_ms; timings->window_bits = tm.window_bits;
    timings->windows = tm.windows; timings->tasks = tm.tasks; timings->max_bucket = tm.max_bucket;
    timings->accumulate_threads = tm.accumulate_threads;
    timings->order_ms = tm.order_ms;
  }
  return SG_OK;
}
int sg_commit_dev(uint64_t srs_handle, int basis, const void* d_scalars, size_t n, void* stream,
                  uint8_t out_affine[64]) {
  return sg_commit_dev_timed(srs_handle, basis, d_scalars, n, stream, out_affine, nullptr);
}
// `count` commitments of equal length against one basis as fused jobs (the advice / quotient-piece
// commitments of one proof phase); takes the fixed-base path when the table exists
int sg_commit_batch_dev(uint64_t srs_handle, int basis, const void* const* d_scalars, size_t count, size_t n,
                        void* stream, uint8_t* out_affine) {
  if ((count && (!d_scalars || !out_affine)) || basis < 0 || basis > 2) return fail(SG_ERR_INVALID, "sg_commit_batch: bad argument");
  for (size_t i = 0; i < count; i++)
    if (n && !d_scalars[i]) return fail(SG_ERR_INVALID, "sg_commit_batch: null argument");
  LOCKED_CTX();
  Srs srs_v;
  if (!find_srs(srs_handle, &srs_v)) return fail(SG_ERR_INVALID, "unknown SRS handle");
  Srs* srs_p = &srs_v;
  if (n > ((size_t)1 << (*srs_p).k)) return fail(SG_ERR_INVALID, "sg_commit: polynomial longer than the SRS");
  const Srs& s = (*srs_p);
  std::vector<size_t> ns(count, n);
  const bool diff = basis == 2 && diff_form_ready(s, n);
  if (basis == 2 && !diff) basis = 1;
  const bool fixed = s.tab[basis].table != nullptr;
  std::vector<const void*> bases(count, fixed ? (const void*)s.tab[basis].table : (const void*)(basis ? s.g_lagrange : s.g));
  std::vector<uint8_t> flags(count, diff ? 1 : 0);
  return msm_batch_locked(d_scalars, bases.data(), fixed ? &s.tab[basis] : nullptr, ns.data(), count, stream, out_affine, flags.data());
}
// the same with one basis per polynomial (0 = g, 1 = g_lagrange): e.g. the grand-product commitments (Lagrange)
// and the random polynomial (coefficients) of one prover phase as ONE fused job
// ---- commit combiner.  Proofs in flight on several host threads (circuits_halo2_amd/batch.py) each issue five commitment
// jobs; alone, every job pays its own sort front-end, bucket reduction and host tail, and the jobs of different threads
// compete for the chip.  A thread that has declared itself (sg_commit_combine_begin) hands its sg_commit_batch*_dev calls
// to the combiner instead: the first caller to find no job running becomes the runner, waits a bounded time for the
// other declared threads to arrive (they do: after one fused job all of them get their points at the same moment and
// reach their next commitment together), takes EVERYTHING pending with the same SRS and length and runs it as ONE
// fused job on a lane of its own; callers that arrive while a job runs form the next one.  No caller ever waits for a
// thread that might not come -- only for a deadline -- so a failed or finished proof cannot block the others.
struct CommitReq {
  uint64_t srs;
  size_t n, count;
  const int* basis;
  const void* const* scalars;
  uint8_t* out;
  hipEvent_t ready;      // recorded on the caller's stream after its inputs were enqueued
  int rc = SG_OK;
  bool done = false;
  char err[256] = "";
};
struct Combiner {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<CommitReq*> pending;
  int runners = 0;                       // fused jobs running now
  int busy = 0;                          // requests inside those jobs
  int members = 0;                       // threads between sg_commit_combine_begin and _end
  std::atomic<int> wait_us{300};         // how long a runner waits for requests to arrive (sg_set_param "commit.combine_wait_us")
  std::atomic<int> target{4};            // ... or until this many are pending ("commit.combine_target")
  std::atomic<int> max_runners{1};       // fused jobs that may run side by side, each on a lane of its own ("commit.combine_runners")
  std::atomic<uint64_t> jobs{0}, requests{0};   // statistics: fused jobs run, requests served
  std::atomic<uint64_t> isolated{0};     // members re-run alone after their fused job failed as a whole
  std::atomic<int> fail_next{0};         // test hook ("debug.fail_next_fused_job"): the next fused job reports SG_ERR_NOMEM unrun
};
Combiner g_comb;
thread_local bool t_combine = false;
thread_local hipEvent_t t_ready = nullptr;

static int commit_batch_mixed_core(uint64_t srs_handle, const int* basis, const void* const* d_scalars, size_t count, size_t n,
                                   void* stream, uint8_t* out_affine);

// runs on the runner's thread: one fused job for all requests of `batch` (same SRS, same n)
static void combiner_run_unguarded(const std::vector<CommitReq*>& batch);
static void combiner_run(const std::vector<CommitReq*>& batch) {
  try {
    combiner_run_unguarded(batch);
  } catch (const std::exception& e) {   // (allocation failures of the host vectors: every member learns of it)
    for (CommitReq* r : batch) {
      r->rc = SG_ERR_NOMEM;
      std::snprintf(r->err, sizeof r->err, "commit combiner: %s", e.what());
    }
  }
}
static void combiner_run_unguarded(const std::vector<CommitReq*>& batch) {
  std::vector<int> basis;
  std::vector<const void*> scalars;
  size_t total = 0;
  for (CommitReq* r : batch) total += r->count;
  basis.reserve(total);
  scalars.reserve(total);
  for (CommitReq* r : batch)
    for (size_t i = 0; i < r->count; i++) {
      basis.push_back(r->basis[i]);
      scalars.push_back(r->scalars[i]);
    }
  std::vector<uint8_t> out(64 * total);
  int rc;
  {
    LaneHold hold;     // the job's own lane: its stream waits for every member's inputs
    rc = hold.rc;
    if (rc == SG_OK) {
      for (CommitReq* r : batch) {
        hipError_t e = hipStreamWaitEvent(g_ctx->stream, r->ready, 0);
        if (e != hipSuccess) { rc = hip_fail("commit combiner: stream wait", e); break; }
      }
    }
    if (rc == SG_OK && batch.size() > 1 && g_comb.fail_next.exchange(0)) rc = fail(SG_ERR_NOMEM, "commit combiner: injected failure of a fused job");
    else if (rc == SG_OK) rc = commit_batch_mixed_core(batch[0]->srs, basis.data(), scalars.data(), total, batch[0]->n, g_ctx->stream, out.data());
  }
  size_t at = 0;
  for (CommitReq* r : batch) {
    r->rc = rc;
    if (rc == SG_OK) std::memcpy(r->out, out.data() + 64 * at, 64 * r->count);
    else std::snprintf(r->err, sizeof r->err, "%s", g_err);
    at += r->count;
  }
  if (rc != SG_OK && batch.size() > 1) {
    // The fused job failed AS A WHOLE -- out of device memory at this size, one member's bad pointer or stale handle.  One
    // member's fault must not cost the others their proofs: every member gets a job of its own (same lane discipline:
    // the job's stream waits for that member's inputs) and its own return value and message.
    for (CommitReq* r : batch) {
      LaneHold hold;
      int rc1 = hold.rc;
      if (rc1 == SG_OK) {
        const hipError_t e = hipStreamWaitEvent(g_ctx->stream, r->ready, 0);
        if (e != hipSuccess) rc1 = hip_fail("commit combiner: stream wait", e);
      }
      if (rc1 == SG_OK) rc1 = commit_batch_mixed_core(r->srs, r->basis, r->scalars, r->count, r->n, g_ctx->stream, r->out);
      r->rc = rc1;
      if (rc1 != SG_OK) std::snprintf(r->err, sizeof r->err, "%s", g_err);
      else r->err[0] = 0;
      g_comb.isolated.fetch_add(1);
    }
  }
  g_comb.jobs.fetch_add(1);
  g_comb.requests.fetch_add(batch.size());
}

static int commit_combined(uint64_t srs_handle, const int* basis, const void* const* d_scalars, size_t count, size_t n,
                           void* stream, uint8_t* out_affine) {
  if (!t_ready) CHECK_HIP(hipEventCreateWithFlags(&t_ready, hipEventDisableTiming), "event");
  CHECK_HIP(hipEventRecord(t_ready, pick_stream(stream)), "event");
  CommitReq req{srs_handle, n, count, basis, d_scalars, out_affine, t_ready};
  std::unique_lock<std::mutex> lk(g_comb.mu);
  g_comb.pending.push_back(&req);
  g_comb.cv.notify_all();                         // a runner waiting for stragglers counts again
  while (!req.done) {
    const bool mine_pending = std::find(g_comb.pending.begin(), g_comb.pending.end(), &req) != g_comb.pending.end();
    if (!mine_pending || g_comb.runners >= g_comb.max_runners.load()) {   // my request is inside a running job, or no runner slot is free
      g_comb.cv.wait(lk);
      continue;
    }
    g_comb.runners++;                             // this thread runs the next job
    // the bounded wait shrinks with the company that can still come: a thread is a member for the whole of its proof, not
    // only around its commitments, so at the tail of a batch the few proofs left would otherwise sit out the full wait
    // (5 ms in batch.prove_batch) at every one of their five jobs for members that are busy elsewhere
    const int may_come = std::max(1, g_comb.members - g_comb.busy - (int)g_comb.pending.size());
    const int wait_us = std::min(g_comb.wait_us.load(), 400 * may_come);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(wait_us);
    // wait for company: until `target` requests are pending, or every declared thread that is not inside a running job
    // has arrived, or the deadline
    while ((int)g_comb.pending.size() < std::min(g_comb.target.load(), g_comb.members - g_comb.busy))
      if (g_comb.cv.wait_until(lk, deadline) == std::cv_status::timeout) break;
    // everything pending with the first request's SRS and length, up to MAX_FUSED polynomials
    if (g_comb.pending.empty()) {   // another runner took everything meanwhile (this thread's request included)
      g_comb.runners--;
      g_comb.cv.notify_all();
      continue;
    }
    std::vector<CommitReq*> batch;
    size_t polys = 0;
    CommitReq* first = g_comb.pending.front();
    for (auto it = g_comb.pending.begin(); it != g_comb.pending.end();) {
      CommitReq* r = *it;
      if (r->srs == first->srs && r->n == first->n && polys + r->count <= MAX_FUSED) {
        batch.push_back(r);
        polys += r->count;
        it = g_comb.pending.erase(it);
      } else {
        ++it;
      }
    }
    g_comb.busy += (int)batch.size();
    lk.unlock();
    combiner_run(batch);
    lk.lock();
    for (CommitReq* r : batch) r->done = true;
    g_comb.busy -= (int)batch.size();
    g_comb.runners--;
    g_comb.cv.notify_all();
  }
  if (req.rc != SG_OK) std::snprintf(g_err, sizeof g_err, "%s", req.err);
  return req.rc;
}

int sg_commit_combine_begin(void) {
  if (t_combine) return SG_OK;
  t_combine = true;
  std::lock_guard<std::mutex> lk(g_comb.mu);
  g_comb.members++;
  return SG_OK;
}
int sg_commit_combine_end(void) {
  if (!t_combine) return SG_OK;
  t_combine = false;
  {
    std::lock_guard<std::mutex> lk(g_comb.mu);
    g_comb.members--;
    g_comb.cv.notify_all();          // a runner waiting for this thread stops counting it
  }
  if (t_ready) {                     // this thread's requests have all returned: nothing waits on the event any more
    (void)hipEventDestroy(t_ready);
    t_ready = nullptr;
  }
  return SG_OK;
}
int sg_commit_combining(void) { return t_combine ? 1 : 0; }
int sg_commit_combine_stats(uint64_t* jobs, uint64_t* requests) {
  if (jobs) *jobs = g_comb.jobs.load();
  if (requests) *requests = g_comb.requests.load();
  return SG_OK;
}

int sg_commit_batch_mixed_dev(uint64_t srs_handle, const int* basis, const void* const* d_scalars, size_t count, size_t n,
                              void* stream, uint8_t* out_affine) {
  if (count && (!d_scalars || !out_affine || !basis)) return fail(SG_ERR_INVALID, "sg_commit_batch_mixed: bad argument");
  for (size_t i = 0; i < count; i++)
    if ((n && !d_scalars[i]) || basis[i] < 0 || (basis[i] & ~SG_BASIS_SPARSE) > 2) return fail(SG_ERR_INVALID, "sg_commit_batch_mixed: bad argument");
  if (t_combine && count && n && count <= MAX_FUSED && g_depth == 0)
    return commit_combined(srs_handle, basis, d_scalars, count, n, stream, out_affine);
  return commit_batch_mixed_core(srs_handle, basis, d_scalars, count, n, stream, out_affine);
}
static int commit_batch_mixed_core(uint64_t srs_handle, const int* basis, const void* const* d_scalars, size_t count, size_t n,
                                   void* stream, uint8_t* out_affine) {
  LOCKED_CTX();
  Srs srs_v;
  if (!find_srs(srs_handle, &srs_v)) return fail(SG_ERR_INVALID, "unknown SRS handle");
  Srs* srs_p = &srs_v;
  if (n > ((size_t)1 << (*srs_p).k)) return fail(SG_ERR_INVALID, "sg_commit: polynomial longer than the SRS");
  const Srs& s = (*srs_p);
  // fixed-base only when both tables exist with one plan; otherwise the generic fused path over g / g_lagrange
  const bool fixed = s.tab[0].table && s.tab[1].table && s.tab[0].c == s.tab[1].c && s.tab[0].n == s.tab[1].n;
  // difference form (basis 2) needs the prefix-sum table on the same plan; otherwise such a column is an ordinary Lagrange one
  const bool diff_ok = fixed && diff_form_ready(s, n) && s.tab[2].c == s.tab[0].c && s.tab[2].n == s.tab[0].n;
  std::vector<size_t> ns(count, n);
  std::vector<const void*> bases(count);
  std::vector<uint8_t> flags(count, 0);
  bool all_sparse = count > 0;
  for (size_t i = 0; i < count; i++) {
    const int want = basis[i] & ~SG_BASIS_SPARSE;
    all_sparse = all_sparse && (basis[i] & SG_BASIS_SPARSE);
    const int b = want == 2 ? (diff_ok ? 2 : 1) : want;
    flags[i] = b == 2;
    bases[i] = fixed ? (const void*)s.tab[b].table : (const void*)(b ? s.g_lagrange : s.g);
  }
  // A job whose columns are all witness-like (mostly zeros and small values: few entries, some of them in heavy buckets) is a
  // latency chain of one task length whatever its size: tasks of 8 instead of 16 halve it (a proof's first commitment job
  // 1.07 -> 1.00 ms) where dense jobs lose by them (profiles/r04_sweeps/task_length_by_phase.txt).  A hint, never semantics.
  struct SegRestore {
    Context& c;
    uint32_t a, b;
    ~SegRestore() { c.msm.config().log_seg = a; c.msm_b.config().log_seg = b; }
  } seg_restore{*g_ctx, g_ctx->msm.config().log_seg, g_ctx->msm_b.config().log_seg};
  // (only for jobs small enough for the 2-D reduction, which adds up to eight partial sums per bucket itself: a fused job of many
  // proofs' columns goes through merge rounds, and shorter tasks would add one)
  if (all_sparse && fixed && count <= 5 && g_ctx->msm.config().log_seg == 0 && n >= ((size_t)1 << 14))
    g_ctx->msm.config().log_seg = g_ctx->msm_b.config().log_seg = 3;
  return msm_batch_locked(d_scalars, bases.data(), fixed ? &s.tab[0] : nullptr, ns.data(), count, stream, out_affine, flags.data());
}
int sg_commit(uint64_t srs_handle, int basis, const uint8_t* scalars, size_t n, uint8_t out_affine[64]) {
  if (!out_affine || (n && !scalars) || basis < 0 || basis > 2) return fail(SG_ERR_INVALID, "sg_commit: bad argument");
  LOCKED_CTX();
  Srs srs_v;
  if (!find_srs(srs_handle, &srs_v)) return fail(SG_ERR_INVALID, "unknown SRS handle");
  Srs* srs_p = &srs_v;
  if (n > ((size_t)1 << (*srs_p).k)) return fail(SG_ERR_INVALID, "sg_commit: polynomial longer than the SRS");
  {
    // no window table for this basis (sg_srs_precompute not called): the generic MSM over the resident bases, in chunks, so that
    // the scalars' upload and one job's latency chains run under another job's accumulation (msm_host_chunked)
    const Srs& sr = *srs_p;
    const int b = (basis == 2 && !diff_form_ready(sr, n)) ? 1 : basis;
    if (b != 2 && !sr.tab[b].table && n >= MSM_HOST_SPLIT_MIN)
      return msm_host_chunked(scalars, nullptr, b ? sr.g_lagrange : sr.g, n, out_affine);
  }
  TRY(upload(g_ctx->stage_a, scalars, n * 32, g_ctx->stream));
  hipError_t e = commit_run((*srs_p), basis, reinterpret_cast<const fp_words*>(g_ctx->stage_a.p), n, g_ctx->stream,
                            out_affine);
  if (e != hipSuccess) return hip_fail("msm", e);
  return SG_OK;
}

// ------------------------------------------------------------------ NTT family
int sg_ntt_fr_dev(void* d_a, const uint8_t omega[32], uint32_t log_n, void* stream) {
  if (!d_a || !omega) return fail(SG_ERR_INVALID, "sg_ntt_fr: null argument");
  LOCKED_CTX();
  words8 w;
  std::memcpy(&w, omega, 32);
  fp_words* a = static_cast<fp_words*>(d_a);
  return ntt_dev(a, (size_t)1 << log_n, a, log_n, w, nullptr, nullptr, nullptr, pick_stream(stream));
}
int sg_ntt_fr(uint8_t* a, const uint8_t omega[32], uint32_t log_n) {
  if (!a || !omega || log_n > 28) return fail(SG_ERR_INVALID, "sg_ntt_fr: bad argument");
  LOCKED_CTX();
  const size_t bytes = (size_t)32 << log_n;
  TRY(upload(g_ctx->stage_a, a, bytes, g_ctx->stream));
  words8 w;
  std::memcpy(&w, omega, 32);
  fp_words* d = reinterpret_cast<fp_words*>(g_ctx->stage_a.p);
  TRY(ntt_dev(d, (size_t)1 << log_n, d, log_n, w, nullptr, nullptr, nullptr, g_ctx->stream));
  return download(a, d, bytes, g_ctx->stream);
}
// A batch of independent in-place transforms of one size (the 9 lagrange_to_coeff / 9
// coeff_to_extended calls of a proof): round-robin over the two batch streams so that one
// transform's tail overlaps the next one's head.  divisor == NULL: plain best_fft.
int sg_ntt_fr_batch_dev(void* const* d_a, size_t count, const uint8_t omega[32], const uint8_t* divisor,
                        uint32_t log_n, void* stream) {
  if ((count && !d_a) || !omega || log_n > 28) return fail(SG_ERR_INVALID, "sg_ntt_fr_batch: bad argument");
  LOCKED_CTX();
  Context& c = *g_ctx;
  words8 w, dv;
  std::memcpy(&w, omega, 32);
  if (divisor) std::memcpy(&dv, divisor, 32);
  const size_t n = (size_t)1 << log_n;
  const bool need_scratch = log_n > c.ntt.config().max_single_log;
  if (log_n >= 1 && log_n <= 18) {
    // small transforms: one launch per pass for up to 16 vectors (each launch is at its ~5 us floor otherwise);
    // asynchronous on the caller's stream, scratch per stream
    hipStream_t s = pick_stream(stream);
    for (size_t first = 0; first < count; first += NTT_BATCH_MAX) {
      const uint32_t cnt = (uint32_t)std::min<size_t>(NTT_BATCH_MAX, count - first);
      fp_words* ptrs[NTT_BATCH_MAX];
      for (uint32_t i = 0; i < cnt; i++) {
        if (!d_a[first + i]) return fail(SG_ERR_INVALID, "sg_ntt_fr_batch: null vector");
        ptrs[i] = static_cast<fp_words*>(d_a[first + i]);
      }
      uint8_t* scr = nullptr;
      if (need_scratch) {
        hipError_t e = scratch_for(s, 3, std::min<size_t>(count, NTT_BATCH_MAX) * n * 32, &scr);
        if (e != hipSuccess) return hip_fail("ntt scratch", e);
      }
      hipError_t e = c.ntt.transform_batch(ptrs, cnt, reinterpret_cast<fp_words*>(scr), log_n, w, divisor ? &dv : nullptr, s);
      if (e != hipSuccess) return hip_fail("ntt batch", e);
    }
    return SG_OK;
  }
  // one scratch area per stream
  if (need_scratch) {
    hipError_t e = c.scratch.reserve(2 * n * 32);
    if (e != hipSuccess) return hip_fail("ntt scratch", e);
  }
  CHECK_HIP(hipEventRecord(c.ev_in, pick_stream(stream)), "event");
  for (auto& bs : c.bstream) CHECK_HIP(hipStreamWaitEvent(bs, c.ev_in, 0), "stream wait");
  for (size_t i = 0; i < count; i++) {
    if (!d_a[i]) return fail(SG_ERR_INVALID, "sg_ntt_fr_batch: null vector");
    const int k = (int)(i & 1);
    fp_words* a = static_cast<fp_words*>(d_a[i]);
    fp_words* scratch = need_scratch ? reinterpret_cast<fp_words*>(c.scratch.p) + (size_t)k * n : nullptr;
    hipError_t e = c.ntt.transform(a, n, a, scratch, log_n, w, divisor ? &dv : nullptr, nullptr, nullptr, c.bstream[k]);
    if (e != hipSuccess) return hip_fail("ntt batch", e);
  }
  for (auto& bs : c.bstream) CHECK_HIP(host_wait_stream(bs), "stream sync");
  return SG_OK;
}

// the same out of place (d_out[i] = transform of d_in[i]; the inputs stay): what `lagrange_to_coeff` of a column that is
// still needed in Lagrange form costs without a device-to-device copy in front of it
int sg_ntt_fr_batch_oop_dev(const void* const* d_in, void* const* d_out, size_t count, const uint8_t omega[32], const uint8_t* divisor,
                            uint32_t log_n, void* stream) {
  if ((count && (!d_in || !d_out)) || !omega || log_n > 28) return fail(SG_ERR_INVALID, "sg_ntt_fr_batch_oop: bad argument");
  const size_t n = (size_t)1 << log_n;
  {
    // the vectors of a launch are transformed side by side: an output that overlaps ANY input, or another output, would be
    // read or written by two workgroups at once -- refused here ("the inputs untouched" is the call's promise)
    auto overlap = [&](const void* a, const void* b) {
      const uintptr_t x = reinterpret_cast<uintptr_t>(a), y = reinterpret_cast<uintptr_t>(b);
      return x < y + 32 * n && y < x + 32 * n;
    };
    for (size_t i = 0; i < count; i++) {
      if (!d_in[i] || !d_out[i]) return fail(SG_ERR_INVALID, "sg_ntt_fr_batch_oop: null vector");
      for (size_t j = 0; j < count; j++) {
        if (overlap(d_out[i], d_in[j])) return fail(SG_ERR_INVALID, "sg_ntt_fr_batch_oop: an output overlaps an input");
        if (j != i && overlap(d_out[i], d_out[j])) return fail(SG_ERR_INVALID, "sg_ntt_fr_batch_oop: two outputs overlap");
      }
    }
  }
  if (log_n < 1 || log_n > 18) {   // outside the batched plans: copy, then in place
    for (size_t i = 0; i < count; i++)
      CHECK_HIP(hipMemcpyAsync(d_out[i], d_in[i], n * 32, hipMemcpyDeviceToDevice, pick_stream(stream)), "D2D copy");
    return sg_ntt_fr_batch_dev(d_out, count, omega, divisor, log_n, stream);
  }
  LOCKED_CTX();
  Context& c = *g_ctx;
  words8 w, dv;
  std::memcpy(&w, omega, 32);
  if (divisor) std::memcpy(&dv, divisor, 32);
  hipStream_t s = pick_stream(stream);
  for (size_t first = 0; first < count; first += NTT_BATCH_MAX) {
    const uint32_t cnt = (uint32_t)std::min<size_t>(NTT_BATCH_MAX, count - first);
    fp_words* outs[NTT_BATCH_MAX];
    const fp_words* ins[NTT_BATCH_MAX];
    for (uint32_t i = 0; i < cnt; i++) {
      outs[i] = static_cast<fp_words*>(d_out[first + i]);
      ins[i] = static_cast<const fp_words*>(d_in[first + i]);
    }
    hipError_t e = c.ntt.transform_batch(outs, cnt, nullptr, log_n, w, divisor ? &dv : nullptr, s, ins, n);
    if (e != hipSuccess) return hip_fail("ntt batch", e);
  }
  return SG_OK;
}

int sg_intt_fr_dev(void* d_a, const uint8_t omega_inv[32], const uint8_t divisor[32], uint32_t log_n, void* stream) {
  if (!d_a || !omega_inv || !divisor) return fail(SG_ERR_INVALID, "sg_intt_fr: null argument");
  LOCKED_CTX();
  words8 w, d;
  std::memcpy(&w, omega_inv, 32);
  std::memcpy(&d, divisor, 32);
  fp_words* a = static_cast<fp_words*>(d_a);
  return ntt_dev(a, (size_t)1 << log_n, a, log_n, w, &d, nullptr, nullptr, pick_stream(stream));
}
int sg_intt_fr(uint8_t* a, const uint8_t omega_inv[32], const uint8_t divisor[32], uint32_t log_n) {
  if (!a || !omega_inv || !divisor || log_n > 28) return fail(SG_ERR_INVALID, "sg_intt_fr: bad argument");
  LOCKED_CTX();
  const size_t bytes = (size_t)32 << log_n;
  TRY(upload(g_ctx->stage_a, a, bytes, g_ctx->stream));
  words8 w, dv;
  std::memcpy(&w, omega_inv, 32);
  std::memcpy(&dv, divisor, 32);
  fp_words* d = reinterpret_cast<fp_words*>(g_ctx->stage_a.p);
  TRY(ntt_dev(d, (size_t)1 << log_n, d, log_n, w, &dv, nullptr, nullptr, g_ctx->stream));
  return download(a, d, bytes, g_ctx->stream);
}
int sg_lagrange_to_coeff_dev(void* d_a, uint32_t k, void* stream) {
  if (!d_a || k > 28) return fail(SG_ERR_INVALID, "sg_lagrange_to_coeff: bad argument");
  LOCKED_CTX();
  const DomainConsts* dc;
  TRY(get_consts(k, &dc));
  TRY(sync_own_stream_into(pick_stream(stream)));
  fp_words* a = static_cast<fp_words*>(d_a);
  return ntt_dev(a, (size_t)1 << k, a, k, dc->omega_inv, &dc->n_inv, nullptr, nullptr, pick_stream(stream));
}
int sg_lagrange_to_coeff(uint8_t* a, uint32_t k) {
  if (!a || k > 28) return fail(SG_ERR_INVALID, "sg_lagrange_to_coeff: bad argument");
  LOCKED_CTX();
  const DomainConsts* dc;
  TRY(get_consts(k, &dc));
  const size_t bytes = (size_t)32 << k;
  TRY(upload(g_ctx->stage_a, a, bytes, g_ctx->stream));
  fp_words* d = reinterpret_cast<fp_words*>(g_ctx->stage_a.p);
  TRY(ntt_dev(d, (size_t)1 << k, d, k, dc->omega_inv, &dc->n_inv, nullptr, nullptr, g_ctx->stream));
  return download(a, d, bytes, g_ctx->stream);
}

int sg_coeff_to_extended_dev(const void* d_coeffs, uint32_t k, uint32_t ext_k, void* d_out, void* stream) {
  if (!d_coeffs || !d_out || ext_k > 28 || k > ext_k || d_coeffs == d_out)
    return fail(SG_ERR_INVALID, "sg_coeff_to_extended: bad argument");
  LOCKED_CTX();
  const DomainConsts* dc;
  TRY(get_consts(ext_k, &dc));
  TRY(sync_own_stream_into(pick_stream(stream)));
  words8 pre[3] = {dc->one, dc->zeta, dc->zeta2};
  return ntt_dev(static_cast<const fp_words*>(d_coeffs), (size_t)1 << k, static_cast<fp_words*>(d_out), ext_k, dc->omega,
                 nullptr, pre, nullptr, pick_stream(stream));
}
// several columns at once: one launch per pass while the extended domain is small (<= 2^18), one transform after
// the other above that (a 2^20 transform fills the chip on its own)
int sg_coeff_to_extended_batch_dev(const void* const* d_coeffs, void* const* d_out, size_t count, uint32_t k, uint32_t ext_k,
                                   void* stream) {
  if ((count && (!d_coeffs || !d_out)) || ext_k > 28 || k > ext_k) return fail(SG_ERR_INVALID, "sg_coeff_to_extended_batch: bad argument");
  for (size_t i = 0; i < count; i++)
    if (!d_coeffs[i] || !d_out[i] || d_coeffs[i] == d_out[i]) return fail(SG_ERR_INVALID, "sg_coeff_to_extended_batch: bad vector");
  LOCKED_CTX();
  const DomainConsts* dc;
  TRY(get_consts(ext_k, &dc));
  hipStream_t s = pick_stream(stream);
  TRY(sync_own_stream_into(s));
  words8 pre[3] = {dc->one, dc->zeta, dc->zeta2};
  if (ext_k >= 1 && ext_k <= 18) {
    for (size_t first = 0; first < count; first += NTT_BATCH_MAX) {
      const uint32_t cnt = (uint32_t)std::min<size_t>(NTT_BATCH_MAX, count - first);
      hipError_t e = g_ctx->ntt.transform_batch(reinterpret_cast<fp_words* const*>(d_out + first), cnt, nullptr, ext_k, dc->omega,
                                                nullptr, s, reinterpret_cast<const fp_words* const*>(d_coeffs + first),
                                                (size_t)1 << k, pre);
      if (e != hipSuccess) return hip_fail("coeff_to_extended batch", e);
    }
    return SG_OK;
  }
  for (size_t i = 0; i < count; i++)
    TRY(ntt_dev(static_cast<const fp_words*>(d_coeffs[i]), (size_t)1 << k, static_cast<fp_words*>(d_out[i]), ext_k, dc->omega,
                nullptr, pre, nullptr, s));
  return SG_OK;
}
int sg_coeff_to_extended(const uint8_t* coeffs, uint32_t k, uint32_t ext_k, uint8_t* out) {
  if (!coeffs || !out || ext_k > 28 || k > ext_k) return fail(SG_ERR_INVALID, "sg_coeff_to_extended: bad argument");
  LOCKED_CTX();
  const DomainConsts* dc;
  TRY(get_consts(ext_k, &dc));
  TRY(upload(g_ctx->stage_a, coeffs, (size_t)32 << k, g_ctx->stream));
  hipError_t e = g_ctx->stage_b.reserve((size_t)32 << ext_k);
  if (e != hipSuccess) return hip_fail("staging buffer", e);
  words8 pre[3] = {dc->one, dc->zeta, dc->zeta2};
  TRY(ntt_dev(reinterpret_cast<const fp_words*>(g_ctx->stage_a.p), (size_t)1 << k, reinterpret_cast<fp_words*>(g_ctx->stage_b.p),
              ext_k, dc->omega, nullptr, pre, nullptr, g_ctx->stream));
  return download(out, g_ctx->stage_b.p, (size_t)32 << ext_k, g_ctx->stream);
}
int sg_extended_to_coeff_dev(void* d_ext, uint32_t k, uint32_t ext_k, void* stream) {
  if (!d_ext || ext_k > 28 || k > ext_k) return fail(SG_ERR_INVALID, "sg_extended_to_coeff: bad argument");
  LOCKED_CTX();
  const DomainConsts* dc;
  TRY(get_consts(ext_k, &dc));
  TRY(sync_own_stream_into(pick_stream(stream)));
  // undo the coset: a[i] *= zeta^-(i mod 3) = {1, zeta^2, zeta}; the 2^-ext_k divisor rides along
  words8 post[3] = {dc->n_inv, dc->ninv_zeta2, dc->ninv_zeta};
  fp_words* a = static_cast<fp_words*>(d_ext);
  return ntt_dev(a, (size_t)1 << ext_k, a, ext_k, dc->omega_inv, nullptr, nullptr, post, pick_stream(stream));
}
int sg_extended_to_coeff(uint8_t* ext, uint32_t k, uint32_t ext_k) {
  if (!ext || ext_k > 28 || k > ext_k) return fail(SG_ERR_INVALID, "sg_extended_to_coeff: bad argument");
  LOCKED_CTX();
  const DomainConsts* dc;
  TRY(get_consts(ext_k, &dc));
  const size_t bytes = (size_t)32 << ext_k;
  TRY(upload(g_ctx->stage_a, ext, bytes, g_ctx->stream));
  words8 post[3] = {dc->n_inv, dc->ninv_zeta2, dc->ninv_zeta};
  fp_words* a = reinterpret_cast<fp_words*>(g_ctx->stage_a.p);
  TRY(ntt_dev(a, (size_t)1 << ext_k, a, ext_k, dc->omega_inv, nullptr, nullptr, post, g_ctx->stream));
  return download(ext, a, bytes, g_ctx->stream);
}

// ------------------------------------------------------------------ the quotient on d cosets (quotient.h)
static int coset_tables_for(uint32_t k, uint32_t ext_k, uint32_t nc, const Context::CosetTables** out) {
  using summa::prover::Fr;
  Context& c = *g_ctx;
  const auto key = std::make_tuple(k, ext_k, nc);
  auto it = c.coset_tables.find(key);
  if (it == c.coset_tables.end()) {
    const DomainConsts *dk, *de;
    TRY(get_consts(k, &dk));
    TRY(get_consts(ext_k, &de));
    Context::CosetTables t;
    Fr zeta, w_ext;
    std::memcpy(zeta.l, &dk->zeta, 32);
    std::memcpy(w_ext.l, &de->omega, 32);
    std::vector<Fr> shift(nc), gamma(nc);
    words8 inv_shift[MAX_COSETS];
    const uint64_t n_limbs[4] = {(uint64_t)1 << k, 0, 0, 0};
    for (uint32_t b = 0; b < nc; b++) {
      shift[b] = zeta * w_ext.pow((uint64_t)b);
      gamma[b] = shift[b].pow(n_limbs);
      std::memcpy(&t.shift[b], shift[b].l, 32);
      const Fr si = shift[b].inv();
      std::memcpy(&inv_shift[b], si.l, 32);
    }
    // V[b][t] = gamma_b^t; inverse by Gauss-Jordan on [V | I] (the gammas are distinct: the cosets differ)
    std::vector<std::vector<Fr>> a(nc, std::vector<Fr>(2 * nc, Fr::zero()));
    for (uint32_t b = 0; b < nc; b++) {
      Fr pw = Fr::one();
      for (uint32_t tt = 0; tt < nc; tt++) {
        a[b][tt] = pw;
        pw = pw * gamma[b];
      }
      a[b][nc + b] = Fr::one();
    }
    for (uint32_t col = 0; col < nc; col++) {
      uint32_t piv = col;
      while (piv < nc && a[piv][col] == Fr::zero()) piv++;
      if (piv == nc) return fail(SG_ERR_INVALID, "cosets: singular Vandermonde matrix");
      std::swap(a[piv], a[col]);
      const Fr inv = a[col][col].inv();
      for (auto& v : a[col]) v = v * inv;
      for (uint32_t r = 0; r < nc; r++) {
        if (r == col || a[r][col] == Fr::zero()) continue;
        const Fr f = a[r][col];
        for (uint32_t q = 0; q < 2 * nc; q++) a[r][q] = a[r][q] - f * a[col][q];
      }
    }
    std::memset(t.m, 0, sizeof(t.m));
    for (uint32_t b = 0; b < nc; b++) {
      const Fr d = gamma[b] - Fr::one();
      if (d == Fr::zero()) return fail(SG_ERR_INVALID, "cosets: a coset inside the domain");
      const Fr di = d.inv();
      for (uint32_t tt = 0; tt < nc; tt++) {
        const Fr v = a[tt][nc + b] * di;        // V^-1[t][b] / (gamma_b - 1)
        std::memcpy(t.m[tt * MAX_COSETS + b], v.l, 32);
      }
    }
    const size_t n = (size_t)1 << k;
    CHECK_HIP(hipMalloc(&t.fwd, sizeof(fp_words) * n * nc), "coset tables");
    CHECK_HIP(hipMalloc(&t.inv, sizeof(fp_words) * n * nc), "coset tables");
    hipError_t e = coset_fill_powers(t.fwd, t.shift, nc, k, c.stream);
    if (e == hipSuccess) e = coset_fill_powers(t.inv, inv_shift, nc, k, c.stream);
    if (e == hipSuccess) e = host_wait_stream(c.stream);
    if (e != hipSuccess) return hip_fail("coset tables", e);
    it = c.coset_tables.emplace(key, t).first;
  }
  *out = &it->second;
  return SG_OK;
}
// debug / A-B: 1 = the coset shift as a pass of its own before the transforms (rounds 3-4), 0 = folded into the first NTT pass
static std::atomic<int> g_coset_scale_pass{0};
static bool coset_shape_ok(uint32_t k, uint32_t ext_k, uint32_t nc) {
  return k >= 1 && ext_k > k && ext_k <= 28 && nc >= 1 && nc <= MAX_COSETS && nc <= (1u << (ext_k - k));
}
// size-2^k transforms of `count` vectors in place (forward: omega, no scale; inverse: omega^-1, 2^-k)
static int coset_ntts(fp_words* const* ptrs, size_t count, uint32_t k, bool inverse, hipStream_t s) {
  Context& c = *g_ctx;
  const DomainConsts* dk;
  TRY(get_consts(k, &dk));
  const size_t n = (size_t)1 << k;
  if (k <= 18) {
    const bool need_scratch = k > c.ntt.config().max_single_log;
    for (size_t first = 0; first < count; first += NTT_BATCH_MAX) {
      const uint32_t cnt = (uint32_t)std::min<size_t>(NTT_BATCH_MAX, count - first);
      uint8_t* scr = nullptr;
      if (need_scratch) {
        hipError_t e = scratch_for(s, 3, (size_t)cnt * n * 32, &scr);
        if (e != hipSuccess) return hip_fail("ntt scratch", e);
      }
      hipError_t e = c.ntt.transform_batch(ptrs + first, cnt, reinterpret_cast<fp_words*>(scr), k, inverse ? dk->omega_inv : dk->omega,
                                           inverse ? &dk->n_inv : nullptr, s);
      if (e != hipSuccess) return hip_fail("coset ntt batch", e);
    }
    return SG_OK;
  }
  for (size_t i = 0; i < count; i++)
    TRY(ntt_dev(ptrs[i], n, ptrs[i], k, inverse ? dk->omega_inv : dk->omega, inverse ? &dk->n_inv : nullptr, nullptr, nullptr, s));
  return SG_OK;
}
int sg_coeff_to_cosets_batch_dev(const void* const* d_coeffs, void* const* d_out, size_t count, uint32_t k, uint32_t ext_k,
                                 uint32_t n_cosets, void* stream) {
  if ((count && (!d_coeffs || !d_out)) || !coset_shape_ok(k, ext_k, n_cosets)) return fail(SG_ERR_INVALID, "sg_coeff_to_cosets_batch: bad argument");
  for (size_t i = 0; i < count; i++)
    if (!d_coeffs[i] || !d_out[i] || d_coeffs[i] == d_out[i]) return fail(SG_ERR_INVALID, "sg_coeff_to_cosets_batch: bad vector");
  LOCKED_CTX();
  const Context::CosetTables* t;
  TRY(coset_tables_for(k, ext_k, n_cosets, &t));
  hipStream_t s = pick_stream(stream);
  TRY(sync_own_stream_into(s));
  const size_t n = (size_t)1 << k;
  if (k <= 18 && !g_coset_scale_pass.load()) {
    // the coset shift c_b^i rides on the load of the first NTT pass (a table of 2^261-domain words per coset, indexed like the
    // input): no pass over HBM of its own, and every block of every column is a vector of ONE batched launch per pass
    const DomainConsts* dk;
    TRY(get_consts(k, &dk));
    std::vector<fp_words*> blocks;
    std::vector<const fp_words*> srcs, tabs;
    for (size_t j = 0; j < count; j++)
      for (uint32_t b = 0; b < n_cosets; b++) {
        blocks.push_back(static_cast<fp_words*>(d_out[j]) + b * n);
        srcs.push_back(static_cast<const fp_words*>(d_coeffs[j]));
        tabs.push_back(t->fwd + b * n);
      }
    for (size_t first = 0; first < blocks.size(); first += NTT_BATCH_MAX) {
      const uint32_t cnt = (uint32_t)std::min<size_t>(NTT_BATCH_MAX, blocks.size() - first);
      hipError_t e = g_ctx->ntt.transform_batch(blocks.data() + first, cnt, nullptr, k, dk->omega, nullptr, s, srcs.data() + first, n, nullptr,
                                                tabs.data() + first);
      if (e != hipSuccess) return hip_fail("coset ntt batch", e);
    }
    return SG_OK;
  }
  std::vector<fp_words*> blocks;
  for (size_t first = 0; first < count; first += COSET_BATCH_MAX) {
    const uint32_t cnt = (uint32_t)std::min<size_t>(COSET_BATCH_MAX, count - first);
    CosetScaleArgs a{};
    for (uint32_t j = 0; j < cnt; j++) {
      a.in[j] = static_cast<const fp_words*>(d_coeffs[first + j]);
      a.out[j] = static_cast<fp_words*>(d_out[first + j]);
      for (uint32_t b = 0; b < n_cosets; b++) blocks.push_back(a.out[j] + b * n);
    }
    a.table = t->fwd;
    a.log_n = k;
    a.nc = n_cosets;
    hipError_t e = coset_scale(a, cnt, s);
    if (e != hipSuccess) return hip_fail("coset scale", e);
  }
  return coset_ntts(blocks.data(), blocks.size(), k, false, s);
}
int sg_cosets_to_pieces_dev(void* d_values, void* const* d_pieces, uint32_t k, uint32_t ext_k, uint32_t n_cosets, void* stream) {
  if (!d_values || !d_pieces || !coset_shape_ok(k, ext_k, n_cosets)) return fail(SG_ERR_INVALID, "sg_cosets_to_pieces: bad argument");
  for (uint32_t t = 0; t < n_cosets; t++)
    if (!d_pieces[t]) return fail(SG_ERR_INVALID, "sg_cosets_to_pieces: null piece");
  LOCKED_CTX();
  const Context::CosetTables* t;
  TRY(coset_tables_for(k, ext_k, n_cosets, &t));
  hipStream_t s = pick_stream(stream);
  TRY(sync_own_stream_into(s));
  const size_t n = (size_t)1 << k;
  fp_words* v = static_cast<fp_words*>(d_values);
  std::vector<fp_words*> blocks;
  for (uint32_t b = 0; b < n_cosets; b++) blocks.push_back(v + b * n);
  TRY(coset_ntts(blocks.data(), blocks.size(), k, true, s));
  CosetCombineArgs a{};
  a.raw = v;
  for (uint32_t i = 0; i < n_cosets; i++) {
    a.pieces[i] = static_cast<fp_words*>(d_pieces[i]);
    const uint8_t* lo = reinterpret_cast<const uint8_t*>(a.pieces[i]);
    const uint8_t* vb = reinterpret_cast<const uint8_t*>(v);
    if (lo < vb + 32 * n * n_cosets && vb < lo + 32 * n) return fail(SG_ERR_INVALID, "sg_cosets_to_pieces: the pieces may not overlap the values");
  }
  a.table_inv = t->inv;
  a.log_n = k;
  a.nc = n_cosets;
  std::memcpy(a.m, t->m, sizeof(a.m));
  hipError_t e = coset_combine(a, s);
  if (e != hipSuccess) return hip_fail("coset combine", e);
  return SG_OK;
}

static int t_eval_table(uint32_t k, uint32_t ext_k, const fp_words** out) {
  Context& c = *g_ctx;
  uint64_t key = ((uint64_t)k << 32) | ext_k;
  auto it = c.t_evals.find(key);
  if (it == c.t_evals.end()) {
    const DomainConsts* dc;
    TRY(get_consts(ext_k, &dc));
    uint32_t cnt = 1u << (ext_k - k);
    fp_words* d = nullptr;
    CHECK_HIP(hipMalloc(&d, sizeof(fp_words) * cnt), "t_evaluations");
    t_eval_kernel<<<(cnt + 63) / 64, 64, 0, c.stream>>>(k, ext_k, dc->omega, d);
    CHECK_HIP(host_wait_stream(c.stream), "t_evaluations");
    it = c.t_evals.emplace(key, d).first;
  }
  *out = it->second;
  return SG_OK;
}
int sg_divide_by_vanishing_poly_dev(void* d_ext, uint32_t k, uint32_t ext_k, void* stream) {
  if (!d_ext || ext_k > 28 || k > ext_k) return fail(SG_ERR_INVALID, "sg_divide_by_vanishing_poly: bad argument");
  LOCKED_CTX();
  const fp_words* tab;
  TRY(t_eval_table(k, ext_k, &tab));
  hipError_t e = ntt_scale_periodic(static_cast<fp_words*>(d_ext), tab, 1u << (ext_k - k), (size_t)1 << ext_k,
                                    pick_stream(stream));
  if (e != hipSuccess) return hip_fail("divide_by_vanishing_poly", e);
  return SG_OK;
}
int sg_divide_by_vanishing_poly(uint8_t* ext, uint32_t k, uint32_t ext_k) {
  if (!ext || ext_k > 28 || k > ext_k) return fail(SG_ERR_INVALID, "sg_divide_by_vanishing_poly: bad argument");
  LOCKED_CTX();
  const fp_words* tab;
  TRY(t_eval_table(k, ext_k, &tab));
  const size_t bytes = (size_t)32 << ext_k;
  TRY(upload(g_ctx->stage_a, ext, bytes, g_ctx->stream));
  hipError_t e = ntt_scale_periodic(reinterpret_cast<fp_words*>(g_ctx->stage_a.p), tab, 1u << (ext_k - k),
                                    (size_t)1 << ext_k, g_ctx->stream);
  if (e != hipSuccess) return hip_fail("divide_by_vanishing_poly", e);
  return download(ext, g_ctx->stage_a.p, bytes, g_ctx->stream);
}

int sg_domain_constant(uint32_t k, int which, uint8_t out[32]) {
  if (!out || k > 28 || which < 0 || which > 3) return fail(SG_ERR_INVALID, "sg_domain_constant: bad argument");
  LOCKED_CTX();
  const DomainConsts* dc;
  TRY(get_consts(k, &dc));
  const words8* src = which == 0 ? &dc->omega : which == 1 ? &dc->omega_inv : which == 2 ? &dc->n_inv : &dc->zeta;
  std::memcpy(out, src, 32);
  return SG_OK;
}

// ------------------------------------------------------------------ misc
int sg_g1_fixed_base_mul_dev(const void* d_scalars, size_t n, void* d_out_affine, void* stream) {
  if (n && (!d_scalars || !d_out_affine)) return fail(SG_ERR_INVALID, "sg_g1_fixed_base_mul: null argument");
  LOCKED_CTX();
  hipError_t e = fixed_base_mul(static_cast<const fp_words*>(d_scalars), n, static_cast<g1_affine_mem*>(d_out_affine),
                                pick_stream(stream));
  if (e != hipSuccess) return hip_fail("fixed_base_mul", e);
  return SG_OK;
}
int sg_g1_fixed_base_mul(const uint8_t* scalars, size_t n, uint8_t* out_affine) {
  if (n && (!scalars || !out_affine)) return fail(SG_ERR_INVALID, "sg_g1_fixed_base_mul: null argument");
  LOCKED_CTX();
  TRY(upload(g_ctx->stage_a, scalars, n * 32, g_ctx->stream));
  hipError_t e = g_ctx->stage_b.reserve(n * 64 + 64);
  if (e != hipSuccess) return hip_fail("staging buffer", e);
  e = fixed_base_mul(reinterpret_cast<const fp_words*>(g_ctx->stage_a.p), n, reinterpret_cast<g1_affine_mem*>(g_ctx->stage_b.p),
                     g_ctx->stream);
  if (e != hipSuccess) return hip_fail("fixed_base_mul", e);
  if (!n) return SG_OK;
  return download(out_affine, g_ctx->stage_b.p, n * 64, g_ctx->stream);
}
// Verifier side of ParamsKZG::setup: scalar * (G2 generator) on the host (g2 = 1 * G2, s_g2 = tau * G2)
int sg_g2_generator_mul(const uint8_t scalar[32], uint8_t out[128]) {
  if (!scalar || !out) return fail(SG_ERR_INVALID, "sg_g2_generator_mul: null argument");
  sg::host::g2_generator_mul(scalar, out);
  return SG_OK;
}
// The verifier's last step (halo2 `SingleStrategy` -> multi_miller_loop + final_exponentiation; the EVM's precompile
// 0x08): *ok = (prod_i e(g1[i], g2[i]) == 1).  Host code (host_pairing.h); the slopes of a G2 point are computed once
// and cached by its bytes (the two G2 points of a KZG check are fixed per SRS).
static int pairing_check_impl(const uint8_t* g1_points, const uint8_t* g2_points, size_t n, int* ok, bool plain_check) {
  if (!ok || (n && (!g1_points || !g2_points))) return fail(SG_ERR_INVALID, "sg_pairing_check: null argument");
  if (n > 64) return fail(SG_ERR_INVALID, "sg_pairing_check: at most 64 pairs");
  using namespace sg::host;
  static std::mutex cache_mu;
  static std::map<std::string, PreparedG2> cache;
  std::vector<Affine> ps;
  std::vector<const PreparedG2*> qs;
  const Fq three = fq_from_u64(3);
  for (size_t i = 0; i < n; i++) {
    Affine p;
    std::memcpy(p.x.v, g1_points + 64 * i, 32);
    std::memcpy(p.y.v, g1_points + 64 * i + 32, 32);
    const bool p_inf = p.x.is_zero() && p.y.is_zero();
    if (Fq::geq_p(p.x.v) || Fq::geq_p(p.y.v)) return fail(SG_ERR_INVALID, "sg_pairing_check: G1 coordinate not reduced");
    if (!p_inf && !(p.y.sqr() == p.x.sqr() * p.x + three)) return fail(SG_ERR_INVALID, "sg_pairing_check: G1 point not on the curve");
    const uint8_t* qb = g2_points + 128 * i;
    G2AffinePt q;
    std::memcpy(q.x.c0.v, qb, 32); std::memcpy(q.x.c1.v, qb + 32, 32);
    std::memcpy(q.y.c0.v, qb + 64, 32); std::memcpy(q.y.c1.v, qb + 96, 32);
    q.inf = q.x.is_zero() && q.y.is_zero();
    if (Fq::geq_p(q.x.c0.v) || Fq::geq_p(q.x.c1.v) || Fq::geq_p(q.y.c0.v) || Fq::geq_p(q.y.c1.v))
      return fail(SG_ERR_INVALID, "sg_pairing_check: G2 coordinate not reduced");
    if (!g2_on_curve(q)) return fail(SG_ERR_INVALID, "sg_pairing_check: G2 point not on the twist");
    if (p_inf || q.inf) continue;  // e(O, Q) = e(P, O) = 1
    const PreparedG2* prep;
    {
      std::lock_guard<std::mutex> lk(cache_mu);
      std::string key(reinterpret_cast<const char*>(qb), 128);
      auto it = cache.find(key);
      if (it == cache.end()) {
        if (cache.size() >= 64) cache.clear();
        it = cache.emplace(key, prepare_g2(q)).first;
      }
      prep = &it->second;   // std::map nodes are stable; entries are only dropped by the clear() above
      ps.push_back(p);
      qs.push_back(new PreparedG2(*prep));
    }
  }
  const Fq12 ml = multi_miller_loop(ps, qs);
  for (const PreparedG2* q : qs) delete q;
  const bool one = final_exponentiation(ml).is_one();
  if (plain_check && final_exponentiation_plain(ml).is_one() != one) return fail(SG_ERR_HIP, "sg_pairing_check: the two final exponentiations disagree");
  *ok = one ? 1 : 0;
  return SG_OK;
}
int sg_pairing_check(const uint8_t* g1_points, const uint8_t* g2_points, size_t n, int* ok) {
  return pairing_check_impl(g1_points, g2_points, n, ok, false);
}
// the same with the final exponentiation cross-checked against its definition (tests)
int sg_pairing_check_slow(const uint8_t* g1_points, const uint8_t* g2_points, size_t n, int* ok) {
  return pairing_check_impl(g1_points, g2_points, n, ok, true);
}
// Keccak-256 as Ethereum uses it (`ethers::utils::keccak256`, zk_prover/src/merkle_sum_tree/entry.rs:21; the EVM
// transcript's hash, contracts/src/InclusionVerifier.sol:85-110): host utility for the host-language bindings
int sg_keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  if (!out || (len && !data)) return fail(SG_ERR_INVALID, "sg_keccak256: null argument");
  const auto h = summa::prover::keccak256(data, len);
  std::memcpy(out, h.data(), 32);
  return SG_OK;
}
// ParamsKZG::<Bn256>::setup(k, rng) with tau supplied by the caller's RNG (zk_prover/src/circuits/
// utils.rs:70): g[i] = tau^i G, g_lagrange[i] = L_i(tau) G.  (g2 / s_g2 are verifier-side, not built.)
int sg_kzg_setup_dev(uint32_t k, const uint8_t tau[32], void* d_g, void* d_g_lagrange, void* stream) {
  if (!tau || !d_g || !d_g_lagrange || k > 28) return fail(SG_ERR_INVALID, "sg_kzg_setup: bad argument");
  LOCKED_CTX();
  const size_t n = (size_t)1 << k;
  hipError_t e = g_ctx->stage_a.reserve(n * 32);
  if (e == hipSuccess) e = g_ctx->stage_b.reserve(n * 32);
  if (e != hipSuccess) return hip_fail("staging buffer", e);
  hipStream_t s = pick_stream(stream);
  words8 t;
  std::memcpy(&t, tau, 32);
  fp_words* pw = reinterpret_cast<fp_words*>(g_ctx->stage_a.p);
  fp_words* lg = reinterpret_cast<fp_words*>(g_ctx->stage_b.p);
  kzg_setup_scalars<<<(unsigned)((n + 127) / 128), 128, 0, s>>>(k, t, pw, lg);
  e = fixed_base_mul(pw, n, static_cast<g1_affine_mem*>(d_g), s);
  if (e == hipSuccess) e = fixed_base_mul(lg, n, static_cast<g1_affine_mem*>(d_g_lagrange), s);
  if (e == hipSuccess) e = host_wait_stream(s);  // staging buffers are reused by later calls
  if (e != hipSuccess) return hip_fail("kzg_setup", e);
  return SG_OK;
}
int sg_kzg_setup(uint32_t k, const uint8_t tau[32], uint8_t* g, uint8_t* g_lagrange) {
  if (!tau || !g || !g_lagrange || k > 28) return fail(SG_ERR_INVALID, "sg_kzg_setup: bad argument");
  const size_t bytes = (size_t)64 << k;
  void *dg = nullptr, *dl = nullptr;
  {
    LOCKED_CTX();
    CHECK_HIP(hipMalloc(&dg, bytes), "sg_kzg_setup");
    if (hipMalloc(&dl, bytes) != hipSuccess) {
      (void)hipFree(dg);
      return fail(SG_ERR_NOMEM, "sg_kzg_setup: out of device memory");
    }
  }
  int rc = sg_kzg_setup_dev(k, tau, dg, dl, nullptr);
  if (rc == SG_OK) {
    hipError_t e = hipMemcpy(g, dg, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(g_lagrange, dl, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = hip_fail("sg_kzg_setup", e);
  }
  (void)hipFree(dg);
  (void)hipFree(dl);
  return rc;
}

// N5: best_fft over G1 (FftGroup = G1) as used by ParamsKZG::downsize / g_to_lagrange:
// out[j] = sum_i omega^(ij) * in[i], optionally times `scale`; affine in, affine out.
int sg_g1_fft_dev(const void* d_in, void* d_out, const uint8_t omega[32], const uint8_t* scale, uint32_t log_n,
                  void* stream) {
  if (!d_in || !d_out || !omega || log_n > 28) return fail(SG_ERR_INVALID, "sg_g1_fft: bad argument");
  LOCKED_CTX();
  hipError_t e = g_ctx->scratch.reserve((size_t)144 << log_n);
  if (e != hipSuccess) return hip_fail("g1 fft work space", e);
  words8 w, sc;
  std::memcpy(&w, omega, 32);
  if (scale) std::memcpy(&sc, scale, 32);
  e = g1_fft(static_cast<const g1_affine_mem*>(d_in), static_cast<g1_affine_mem*>(d_out), log_n, w,
             scale ? &sc : nullptr, reinterpret_cast<xyzz29_mem*>(g_ctx->scratch.p), pick_stream(stream));
  if (e == hipSuccess) e = host_wait_stream(pick_stream(stream));  // scratch is shared with the NTT engine
  if (e != hipSuccess) return hip_fail("g1 fft", e);
  return SG_OK;
}
// ParamsKZG::downsize's recomputation of g_lagrange from g[0..2^k): iFFT over G1 (omega^-1, n^-1)
int sg_g1_to_lagrange(const uint8_t* g, uint32_t k, uint8_t* g_lagrange) {
  if (!g || !g_lagrange || k > 28) return fail(SG_ERR_INVALID, "sg_g1_to_lagrange: bad argument");
  const size_t bytes = (size_t)64 << k;
  words8 wi, ni;
  LOCKED_CTX();   // one lane for staging, transform and read-back
  {
    const DomainConsts* dc;
    TRY(get_consts(k, &dc));
    wi = dc->omega_inv;
    ni = dc->n_inv;
    TRY(upload(g_ctx->stage_a, g, bytes, g_ctx->stream));
    hipError_t e = g_ctx->stage_b.reserve(bytes);
    if (e != hipSuccess) return hip_fail("staging buffer", e);
    CHECK_HIP(host_wait_stream(g_ctx->stream), "stream sync");
  }
  int rc = sg_g1_fft_dev(g_ctx->stage_a.p, g_ctx->stage_b.p, reinterpret_cast<const uint8_t*>(&wi),
                         reinterpret_cast<const uint8_t*>(&ni), k, g_ctx->stream);
  if (rc != SG_OK) return rc;
  return download(g_lagrange, g_ctx->stage_b.p, bytes, g_ctx->stream);
}

int sg_fr_to_montgomery_dev(const void* d_in, void* d_out, size_t n, void* stream) {
  if (n && (!d_in || !d_out)) return fail(SG_ERR_INVALID, "null argument");
  LOCKED_CTX();
  hipError_t e = fr_montgomery(static_cast<const fp_words*>(d_in), static_cast<fp_words*>(d_out), n, 1, pick_stream(stream));
  if (e != hipSuccess) return hip_fail("fr_to_montgomery", e);
  return SG_OK;
}
int sg_lookup_permute_small_dev(const void* d_input, const void* d_table, size_t rows, void* d_permuted_input,
                                void* d_permuted_table, void* stream) {
  if (rows && (!d_input || !d_table || !d_permuted_input || !d_permuted_table)) return fail(SG_ERR_INVALID, "sg_lookup_permute: null argument");
  if (rows == 0) return SG_OK;
  LOCKED_CTX();
  hipStream_t s = pick_stream(stream);
  uint8_t* wb = nullptr;
  hipError_t e = scratch_for(s, 4, (LOOKUP_PERMUTE_WORK + 16) * sizeof(uint32_t), &wb);
  if (e != hipSuccess) return hip_fail("lookup permutation work space", e);
  uint32_t* work = reinterpret_cast<uint32_t*>(wb);
  uint32_t* flag = work + LOOKUP_PERMUTE_WORK;
  fp_words *pa = static_cast<fp_words*>(d_permuted_input), *ps = static_cast<fp_words*>(d_permuted_table);
  e = poly_lookup_permute_small(static_cast<const fp_words*>(d_input), static_cast<const fp_words*>(d_table), rows, work, pa, ps, flag, s);
  if (e == hipSuccess) e = fr_montgomery(pa, pa, rows, 1, s);
  if (e == hipSuccess) e = fr_montgomery(ps, ps, rows, 1, s);
  uint32_t h_flag = 0;
  if (e == hipSuccess) e = host_copy_d2h(&h_flag, flag, sizeof h_flag, s);
  if (e != hipSuccess) return hip_fail("lookup permutation", e);
  if (h_flag == 2) return fail(SG_ERR_UNSUPPORTED, "sg_lookup_permute_small: a table value is not below 2^16 (use the general path)");
  if (h_flag == 1) return fail(SG_ERR_WITNESS, "sg_lookup_permute_small: an input value is not in the table");
  return SG_OK;
}
int sg_lookup_permute_small_async_dev(const void* d_input, const void* d_table, size_t rows, void* d_permuted_input,
                                      void* d_permuted_table, void* d_status, void* stream) {
  if (!d_status || (rows && (!d_input || !d_table || !d_permuted_input || !d_permuted_table)))
    return fail(SG_ERR_INVALID, "sg_lookup_permute_small_async: null argument");
  if (rows == 0) return SG_OK;
  LOCKED_CTX();
  hipStream_t s = pick_stream(stream);
  Context::LookupWork& lw = g_ctx->lookup_work[s];
  constexpr size_t ONE = LOOKUP_PERMUTE_WORK + 16;   // words per work space, the two flag words behind the tables
  if (!lw.buf.p) {
    hipError_t e = lw.buf.reserve(2 * ONE);
    if (e == hipSuccess) e = hipMemsetAsync(lw.buf.p, 0, lw.buf.cap * sizeof(uint32_t), s);   // once per stream; afterwards every call cleans for the next
    if (e != hipSuccess) return hip_fail("lookup permutation work space", e);
    lw.next = 0;
  }
  uint32_t* work = lw.buf.p + lw.next * ONE;
  uint32_t* other = lw.buf.p + (lw.next ^ 1u) * ONE;
  lw.next ^= 1u;
  hipError_t e = poly_lookup_permute_small_chained(static_cast<const fp_words*>(d_input), static_cast<const fp_words*>(d_table), rows, work,
                                                   work + LOOKUP_PERMUTE_WORK, other, other + LOOKUP_PERMUTE_WORK,
                                                   static_cast<fp_words*>(d_permuted_input), static_cast<fp_words*>(d_permuted_table),
                                                   static_cast<uint32_t*>(d_status), s);
  if (e != hipSuccess) return hip_fail("lookup permutation", e);
  return SG_OK;
}
int sg_fr_flag_noncanonical_dev(const void* const* d_cols, uint32_t m, size_t n, void* d_flag, void* stream) {
  if (!d_flag || (m && !d_cols) || m > 16) return fail(SG_ERR_INVALID, "sg_fr_flag_noncanonical: bad argument");
  for (uint32_t j = 0; j < m; j++)
    if (n && !d_cols[j]) return fail(SG_ERR_INVALID, "sg_fr_flag_noncanonical: null column");
  if (n >= (1ull << 32)) return fail(SG_ERR_INVALID, "sg_fr_flag_noncanonical: column too long");
  LOCKED_CTX();
  hipError_t e = poly_flag_noncanonical(reinterpret_cast<const fp_words* const*>(d_cols), m, n, static_cast<uint32_t*>(d_flag), pick_stream(stream));
  if (e != hipSuccess) return hip_fail("flag_noncanonical", e);
  return SG_OK;
}
int sg_fr_random_dev(const uint8_t key[32], uint64_t stream_id, void* d_out, size_t n, void* stream) {
  if (!key || (n && !d_out)) return fail(SG_ERR_INVALID, "sg_fr_random: null argument");
  LOCKED_CTX();
  uint32_t k[8];
  std::memcpy(k, key, 32);
  hipError_t e = poly_random(k, stream_id, n, static_cast<fp_words*>(d_out), pick_stream(stream));
  if (e != hipSuccess) return hip_fail("fr_random", e);
  return SG_OK;
}
int sg_fr_random_batch_dev(const uint8_t key[32], uint64_t first_stream_id, void* const* d_out, const size_t* n, uint32_t m, void* stream) {
  if (!key || (m && (!d_out || !n)) || m > RANDOM_BATCH_MAX) return fail(SG_ERR_INVALID, "sg_fr_random_batch: bad argument");
  for (uint32_t d = 0; d < m; d++)
    if (n[d] && !d_out[d]) return fail(SG_ERR_INVALID, "sg_fr_random_batch: null output");
  LOCKED_CTX();
  uint32_t k[8];
  std::memcpy(k, key, 32);
  hipError_t e = poly_random_batch(k, first_stream_id, m, reinterpret_cast<fp_words* const*>(d_out), n, pick_stream(stream));
  if (e != hipSuccess) return hip_fail("fr_random", e);
  return SG_OK;
}
int sg_fr_from_montgomery_dev(const void* d_in, void* d_out, size_t n, void* stream) {
  if (n && (!d_in || !d_out)) return fail(SG_ERR_INVALID, "null argument");
  LOCKED_CTX();
  hipError_t e = fr_montgomery(static_cast<const fp_words*>(d_in), static_cast<fp_words*>(d_out), n, 0, pick_stream(stream));
  if (e != hipSuccess) return hip_fail("fr_from_montgomery", e);
  return SG_OK;
}

// ------------------------------------------------------------------ polynomial helpers
int sg_fr_eval_poly_dev(const void* d_coeffs, size_t n, const uint8_t x[32], void* stream, uint8_t out[32]) {
  if (!x || !out || (n && !d_coeffs)) return fail(SG_ERR_INVALID, "sg_fr_eval_poly: null argument");
  if (n >= (1ull << 32)) return fail(SG_ERR_INVALID, "sg_fr_eval_poly: polynomial too long");
  if (n == 0) {
    std::memset(out, 0, 32);
    return SG_OK;
  }
  LOCKED_CTX();
  const size_t t = poly_eval_tmp_elems(n);
  hipError_t e = g_ctx->scratch.reserve((2 * t + 1) * 32);
  if (e != hipSuccess) return hip_fail("eval_poly work space", e);
  fp_words* tmp = reinterpret_cast<fp_words*>(g_ctx->scratch.p);
  words8 xw;
  std::memcpy(&xw, x, 32);
  hipStream_t s = pick_stream(stream);
  e = poly_eval(static_cast<const fp_words*>(d_coeffs), n, xw, tmp, tmp + t, tmp + 2 * t, s);
  if (e != hipSuccess) return hip_fail("eval_poly", e);
  return download(out, tmp + 2 * t, 32, s);
}
// The evaluation phase of a proof (35 eval_polynomial calls for MstInclusion) as batches: every polynomial
// at its own point, two launches and one read-back per batch of 40
int sg_fr_eval_poly_batch_dev(const void* const* d_polys, size_t n, const uint8_t* points, uint32_t m, void* stream,
                              uint8_t* out) {
  if (m && (!d_polys || !points || !out)) return fail(SG_ERR_INVALID, "sg_fr_eval_poly_batch: null argument");
  if (n > (1ull << 26)) return fail(SG_ERR_INVALID, "sg_fr_eval_poly_batch: polynomial too long");
  if (m == 0) return SG_OK;
  if (n == 0) {
    std::memset(out, 0, 32 * (size_t)m);
    return SG_OK;
  }
  for (uint32_t j = 0; j < m; j++)
    if (!d_polys[j]) return fail(SG_ERR_INVALID, "sg_fr_eval_poly_batch: null polynomial");
  LOCKED_CTX();
  const size_t blocks = poly_eval_batch_blocks(n);
  hipError_t e = g_ctx->scratch.reserve((EVAL_BATCH_MAX * (blocks + 1)) * 32);
  if (e != hipSuccess) return hip_fail("eval_poly work space", e);
  fp_words* partial = reinterpret_cast<fp_words*>(g_ctx->scratch.p);
  // the values land in page-locked host memory the last kernel writes directly: the host waits for the stream once and reads them
  uint8_t *h_mail = nullptr, *d_mail = nullptr;
  e = mailbox(&h_mail, &d_mail);
  if (e != hipSuccess) return hip_fail("eval_poly mailbox", e);
  static_assert(EVAL_BATCH_MAX * 32 <= Context::MAIL_BYTES, "the mailbox holds one batch of evaluations");
  hipStream_t s = pick_stream(stream);
  for (uint32_t first = 0; first < m; first += EVAL_BATCH_MAX) {
    const uint32_t cnt = std::min<uint32_t>(EVAL_BATCH_MAX, m - first);
    words8 xs[EVAL_BATCH_MAX];
    std::memcpy(xs, points + 32 * (size_t)first, 32 * (size_t)cnt);
    e = poly_eval_batch(reinterpret_cast<const fp_words* const*>(d_polys + first), xs, cnt, n, partial, reinterpret_cast<fp_words*>(d_mail), s);
    if (e == hipSuccess) e = host_wait_stream(s);   // synchronises: scratch and mailbox are reused
    if (e != hipSuccess) return hip_fail("eval_poly_batch", e);
    std::memcpy(out + 32 * (size_t)first, h_mail, 32 * (size_t)cnt);
  }
  return SG_OK;
}
int sg_fr_eval_poly(const uint8_t* coeffs, size_t n, const uint8_t x[32], uint8_t out[32]) {
  if (!x || !out || (n && !coeffs)) return fail(SG_ERR_INVALID, "sg_fr_eval_poly: null argument");
  LOCKED_CTX();   // the staging buffer is this lane's until the evaluation has read it
  TRY(upload(g_ctx->stage_a, coeffs, n * 32, g_ctx->stream));
  return sg_fr_eval_poly_dev(g_ctx->stage_a.p, n, x, g_ctx->stream, out);
}
int sg_fr_batch_invert_dev(void* d_a, size_t n, void* stream) {
  if (n && !d_a) return fail(SG_ERR_INVALID, "sg_fr_batch_invert: null argument");
  if (n >= (1ull << 32)) return fail(SG_ERR_INVALID, "sg_fr_batch_invert: vector too long");
  LOCKED_CTX();
  hipError_t e = poly_batch_invert(static_cast<fp_words*>(d_a), n, pick_stream(stream));
  if (e != hipSuccess) return hip_fail("batch_invert", e);
  return SG_OK;
}
int sg_fr_prefix_product_dev(const void* d_a, size_t n, void* d_out, void* stream) {
  if (!d_out || (n && !d_a)) return fail(SG_ERR_INVALID, "sg_fr_prefix_product: null argument");
  if (n > (1ull << 21) - 1) return fail(SG_ERR_INVALID, "sg_fr_prefix_product: at most 2^21 - 1 elements");
  LOCKED_CTX();
  hipStream_t s = pick_stream(stream);
  uint8_t* tmp = nullptr;
  hipError_t e = scratch_for(s, 0, prefix_product_tmp_elems(n + 1) * 32 + 64, &tmp);
  if (e != hipSuccess) return hip_fail("prefix_product work space", e);
  e = poly_prefix_product(static_cast<const fp_words*>(d_a), n, reinterpret_cast<fp_words*>(tmp),
                          static_cast<fp_words*>(d_out), n + 1, nullptr, s);
  if (e != hipSuccess) return hip_fail("prefix_product", e);
  return SG_OK;
}
// delta = 7^(2^28): generator of the 2^28-torsion-free part used to separate permutation columns
static const uint32_t DELTA_M[8] = {0xefd78855u, 0x9a0c322bu, 0x249b563cu, 0x46e82d14u,
                                    0xe0b0b7a7u, 0x5983a663u, 0xaaa111adu, 0x22ab452bu};  // Montgomery-2^256 words
static int grand_product_tail(fp_words* d_mod, size_t n, const uint8_t* z0, void* d_z, hipStream_t s) {
  // z[0] = z0 (or 1), z[i] = z[i-1] * mod[i-1], n values
  uint8_t* tmp = nullptr;
  hipError_t e = scratch_for(s, 0, prefix_product_tmp_elems(n + 1) * 32 + 64, &tmp);
  if (e != hipSuccess) return hip_fail("grand product work space", e);
  words8 init;
  if (z0) std::memcpy(&init, z0, 32);
  e = poly_prefix_product(d_mod, n, reinterpret_cast<fp_words*>(tmp), static_cast<fp_words*>(d_z), n,
                          z0 ? &init : nullptr, s);
  if (e != hipSuccess) return hip_fail("grand product", e);
  return SG_OK;  // asynchronous: ordered on the caller's stream
}
int sg_permutation_product_dev(const void* const* d_values, const void* const* d_sigma, uint32_t ncols,
                               const uint8_t beta[32], const uint8_t gamma[32], const uint8_t delta_start[32],
                               uint32_t k, const uint8_t* z0, void* d_z, void* stream) {
  if (!d_values || !d_sigma || !beta || !gamma || !delta_start || !d_z || ncols == 0 || ncols > PERM_MAX_COLS ||
      k > 21)
    return fail(SG_ERR_INVALID, "sg_permutation_product: bad argument");
  LOCKED_CTX();
  const size_t n = (size_t)1 << k;
  const DomainConsts* dc;
  TRY(get_consts(k, &dc));
  PermCols cols{};
  for (uint32_t c = 0; c < ncols; c++) {
    if (!d_values[c] || !d_sigma[c]) return fail(SG_ERR_INVALID, "sg_permutation_product: null column");
    cols.values[c] = static_cast<const fp_words*>(d_values[c]);
    cols.sigma[c] = static_cast<const fp_words*>(d_sigma[c]);
  }
  hipStream_t s = pick_stream(stream);
  uint8_t* modb = nullptr;
  hipError_t e = scratch_for(s, 1, n * 32 + 64, &modb);
  if (e != hipSuccess) return hip_fail("grand product work space", e);
  fp_words* mod = reinterpret_cast<fp_words*>(modb);
  words8 b, g, ds, dl;
  std::memcpy(&b, beta, 32); std::memcpy(&g, gamma, 32); std::memcpy(&ds, delta_start, 32);
  std::memcpy(&dl, DELTA_M, 32);
  e = poly_perm_fraction(cols, ncols, b, g, ds, dl, dc->omega, n, 0, mod, s);
  if (e == hipSuccess) e = poly_batch_invert(mod, n, s);
  fp_words* pw = nullptr;   // omega^i, i < n (cached per domain): one product instead of one exponentiation per row
  if (e == hipSuccess && n == ((size_t)1 << k)) e = g_ctx->ntt.local_twiddles(dc->omega, k + 1, s, &pw);
  if (e == hipSuccess) e = poly_perm_fraction(cols, ncols, b, g, ds, dl, dc->omega, n, 1, mod, s, pw);
  if (e != hipSuccess) return hip_fail("permutation product", e);
  return grand_product_tail(mod, n, z0, d_z, s);
}
int sg_lookup_product_dev(const void* d_input, const void* d_table, const void* d_permuted_input,
                          const void* d_permuted_table, const uint8_t beta[32], const uint8_t gamma[32], size_t n,
                          void* d_z, void* stream) {
  if (!d_input || !d_table || !d_permuted_input || !d_permuted_table || !beta || !gamma || !d_z || n == 0 ||
      n > (1u << 21) - 1)
    return fail(SG_ERR_INVALID, "sg_lookup_product: bad argument");
  LOCKED_CTX();
  hipStream_t s = pick_stream(stream);
  uint8_t* modb = nullptr;
  hipError_t e = scratch_for(s, 1, n * 32 + 64, &modb);
  if (e != hipSuccess) return hip_fail("grand product work space", e);
  fp_words* mod = reinterpret_cast<fp_words*>(modb);
  words8 b, g;
  std::memcpy(&b, beta, 32); std::memcpy(&g, gamma, 32);
  e = poly_lookup_fraction(static_cast<const fp_words*>(d_permuted_input), static_cast<const fp_words*>(d_permuted_table),
                           b, g, n, 0, mod, s);
  if (e == hipSuccess) e = poly_batch_invert(mod, n, s);
  if (e == hipSuccess)
    e = poly_lookup_fraction(static_cast<const fp_words*>(d_input), static_cast<const fp_words*>(d_table), b, g, n, 1,
                             mod, s);
  if (e != hipSuccess) return hip_fail("lookup product", e);
  return grand_product_tail(mod, n, nullptr, d_z, s);
}

int sg_grand_products_dev(const void* const* d_values, const void* const* d_sigma, const uint32_t* chunk_cols, uint32_t n_chunks,
                          const void* const* d_lookup_cols, uint32_t n_lookups, const uint8_t beta[32], const uint8_t gamma[32],
                          uint32_t k, size_t usable_rows, void* const* d_z, void* stream) {
  return sg_grand_products_closing_dev(d_values, d_sigma, chunk_cols, n_chunks, d_lookup_cols, n_lookups, beta, gamma, k, usable_rows, d_z,
                                       nullptr, stream);
}
int sg_grand_products_closing_dev(const void* const* d_values, const void* const* d_sigma, const uint32_t* chunk_cols, uint32_t n_chunks,
                                  const void* const* d_lookup_cols, uint32_t n_lookups, const uint8_t beta[32], const uint8_t gamma[32],
                                  uint32_t k, size_t usable_rows, void* const* d_z, void* d_closing, void* stream) {
  if (!beta || !gamma || !d_z || (n_chunks && (!d_values || !d_sigma || !chunk_cols)) || (n_lookups && !d_lookup_cols))
    return fail(SG_ERR_INVALID, "sg_grand_products: null argument");
  if (n_chunks + n_lookups == 0) return SG_OK;
  if (n_chunks + n_lookups > GRAND_MAX) return fail(SG_ERR_INVALID, "sg_grand_products: at most 8 products per call");
  if (k == 0 || k > 20) return fail(SG_ERR_INVALID, "sg_grand_products: 1 <= k <= 20");
  const size_t n = (size_t)1 << k;
  if (usable_rows >= n) return fail(SG_ERR_INVALID, "sg_grand_products: usable_rows must be below 2^k");
  LOCKED_CTX();
  const DomainConsts* dc;
  TRY(get_consts(k, &dc));
  GrandProducts g{};
  GrandOut outs{};
  g.n_perm = n_chunks;
  g.n_lookup = n_lookups;
  // delta^(index of the chunk's first column), on the host: a handful of products in the memory domain
  summa::prover::Fr dpow = summa::prover::Fr::one(), dlt;
  std::memcpy(dlt.l, DELTA_M, 32);
  uint32_t col = 0;
  for (uint32_t j = 0; j < n_chunks; j++) {
    if (chunk_cols[j] == 0 || chunk_cols[j] > PERM_MAX_COLS) return fail(SG_ERR_INVALID, "sg_grand_products: 1 .. 8 columns per chunk");
    g.ncols[j] = chunk_cols[j];
    std::memcpy(g.delta_start[j].l, dpow.l, 32);
    for (uint32_t c = 0; c < chunk_cols[j]; c++, col++) {
      if (!d_values[col] || !d_sigma[col]) return fail(SG_ERR_INVALID, "sg_grand_products: null column");
      g.perm[j].values[c] = static_cast<const fp_words*>(d_values[col]);
      g.perm[j].sigma[c] = static_cast<const fp_words*>(d_sigma[col]);
      dpow = dpow * dlt;
    }
  }
  for (uint32_t l = 0; l < n_lookups; l++)
    for (int q = 0; q < 4; q++) {
      if (!d_lookup_cols[4 * l + q]) return fail(SG_ERR_INVALID, "sg_grand_products: null lookup column");
      g.lookup[l][q] = static_cast<const fp_words*>(d_lookup_cols[4 * l + q]);
    }
  for (uint32_t p = 0; p < n_chunks + n_lookups; p++) {
    if (!d_z[p]) return fail(SG_ERR_INVALID, "sg_grand_products: null output");
    outs.z[p] = static_cast<fp_words*>(d_z[p]);
  }
  outs.closing = static_cast<fp_words*>(d_closing);
  outs.closing_row = (uint32_t)usable_rows;
  hipStream_t s = pick_stream(stream);
  uint8_t *modb = nullptr, *tmpb = nullptr;
  hipError_t e = scratch_for(s, 1, grand_products_mod_elems(n, n_chunks + n_lookups) * 32 + 64, &modb);
  if (e == hipSuccess) e = scratch_for(s, 0, grand_products_tmp_elems(n, n_chunks + n_lookups) * 32 + 64, &tmpb);
  if (e != hipSuccess) return hip_fail("grand products work space", e);
  fp_words* pw = nullptr;   // omega^i, i < n (cached per domain)
  if (n_chunks) {
    e = g_ctx->ntt.local_twiddles(dc->omega, k + 1, s, &pw);
    if (e != hipSuccess) return hip_fail("grand products: power table", e);
  }
  words8 b, gm, dl;
  std::memcpy(&b, beta, 32); std::memcpy(&gm, gamma, 32); std::memcpy(&dl, DELTA_M, 32);
  e = poly_grand_products(g, b, gm, dl, n, usable_rows, pw, reinterpret_cast<fp_words*>(modb), reinterpret_cast<fp_words*>(tmpb), outs, s);
  if (e != hipSuccess) return hip_fail("grand products", e);
  return SG_OK;   // asynchronous: ordered on the caller's stream
}

int sg_fr_mul_dev(const void* d_a, const void* d_b, size_t n, void* d_out, void* stream) {
  if (n && (!d_a || !d_b || !d_out)) return fail(SG_ERR_INVALID, "sg_fr_mul: null argument");
  if (n >= (1ull << 32)) return fail(SG_ERR_INVALID, "sg_fr_mul: vector too long");
  LOCKED_CTX();
  hipError_t e = poly_mul_elementwise(static_cast<const fp_words*>(d_a), static_cast<const fp_words*>(d_b), n,
                                      static_cast<fp_words*>(d_out), pick_stream(stream));
  if (e != hipSuccess) return hip_fail("fr_mul", e);
  return SG_OK;
}

// halo2's kate_division(a, b) (arithmetic.rs): the quotient of a(X) by (X - b), as SHPLONK's multi-open
// applies it once per opening point; remainder = a(b) comes for free
int sg_fr_kate_division_dev(const void* d_a, size_t n, const uint8_t b[32], void* d_q, uint8_t* remainder_out,
                            void* stream) {
  if (!b || (n && (!d_a || !d_q))) return fail(SG_ERR_INVALID, "sg_fr_kate_division: null argument");
  if (n > (1ull << 21)) return fail(SG_ERR_INVALID, "sg_fr_kate_division: at most 2^21 coefficients");
  if (d_a == d_q && n) return fail(SG_ERR_INVALID, "sg_fr_kate_division: the quotient must not alias the input");
  if (n == 0) {
    if (remainder_out) std::memset(remainder_out, 0, 32);
    return SG_OK;
  }
  LOCKED_CTX();
  words8 bw;
  std::memcpy(&bw, b, 32);
  hipStream_t s = pick_stream(stream);
  uint8_t* tb = nullptr;
  hipError_t e = scratch_for(s, 0, 1025 * 32 + 64, &tb);
  if (e != hipSuccess) return hip_fail("kate_division work space", e);
  fp_words* tmp = reinterpret_cast<fp_words*>(tb);
  uint8_t *h_mail = nullptr, *d_mail = nullptr;
  if (remainder_out) {
    e = mailbox(&h_mail, &d_mail);
    if (e != hipSuccess) return hip_fail("kate_division mailbox", e);
  }
  e = poly_kate_division(static_cast<const fp_words*>(d_a), n, bw, tmp, static_cast<fp_words*>(d_q),
                         remainder_out ? reinterpret_cast<fp_words*>(d_mail) : nullptr, s);
  // asynchronous unless the caller wants the remainder on the host (written by the kernel into mapped host memory)
  if (e == hipSuccess && remainder_out) {
    e = host_wait_stream(s);
    if (e == hipSuccess) std::memcpy(remainder_out, h_mail, 32);
  }
  if (e != hipSuccess) return hip_fail("kate_division", e);
  return SG_OK;
}
int sg_fr_kate_division_rem_dev(const void* d_a, size_t n, const uint8_t b[32], void* d_q, void* d_remainder, void* stream) {
  if (!b || !d_remainder || (n && (!d_a || !d_q))) return fail(SG_ERR_INVALID, "sg_fr_kate_division_rem: null argument");
  if (n == 0 || n > (1ull << 21)) return fail(SG_ERR_INVALID, "sg_fr_kate_division_rem: between 1 and 2^21 coefficients");
  if (d_a == d_q) return fail(SG_ERR_INVALID, "sg_fr_kate_division_rem: the quotient must not alias the input");
  LOCKED_CTX();
  words8 bw;
  std::memcpy(&bw, b, 32);
  hipStream_t s = pick_stream(stream);
  uint8_t* tb = nullptr;
  hipError_t e = scratch_for(s, 0, 1025 * 32 + 64, &tb);
  if (e != hipSuccess) return hip_fail("kate_division work space", e);
  e = poly_kate_division(static_cast<const fp_words*>(d_a), n, bw, reinterpret_cast<fp_words*>(tb), static_cast<fp_words*>(d_q),
                         static_cast<fp_words*>(d_remainder), s);
  if (e != hipSuccess) return hip_fail("kate_division", e);
  return SG_OK;
}
// m <= 16 exact divisions q_j = a_j / (X - b_j) in one launch per scan step (a_j may repeat: by partial fractions the
// divisions of one rotation set are independent divisions of the same polynomial).  Asynchronous on `stream`.
// how many elements of the given columns are not canonical (word value >= r)?  Asynchronous: *d_count (a u32 in device
// memory) holds the number once the stream reaches this point.
int sg_fr_count_noncanonical_dev(const void* const* d_cols, uint32_t m, size_t n, void* d_count, void* stream) {
  if (!d_count || (m && !d_cols)) return fail(SG_ERR_INVALID, "sg_fr_count_noncanonical: null argument");
  if (m > 16) return fail(SG_ERR_INVALID, "sg_fr_count_noncanonical: at most 16 columns per call");
  if (n > 0xffffffffull) return fail(SG_ERR_INVALID, "sg_fr_count_noncanonical: too many rows");
  for (uint32_t j = 0; j < m; j++)
    if (n && !d_cols[j]) return fail(SG_ERR_INVALID, "sg_fr_count_noncanonical: null column");
  LOCKED_CTX();
  hipError_t e = poly_count_noncanonical(reinterpret_cast<const fp_words* const*>(d_cols), m, n, reinterpret_cast<uint32_t*>(d_count),
                                         reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail("count_noncanonical", e);
  return SG_OK;
}
int sg_fr_kate_division_batch_dev(const void* const* d_a, size_t n, const uint8_t* points, uint32_t m, void* const* d_q,
                                  void* stream) {
  if (m && (!d_a || !points || !d_q)) return fail(SG_ERR_INVALID, "sg_fr_kate_division_batch: null argument");
  if (m > 16) return fail(SG_ERR_INVALID, "sg_fr_kate_division_batch: at most 16 divisions per call");
  if (n > (1ull << 21)) return fail(SG_ERR_INVALID, "sg_fr_kate_division_batch: at most 2^21 coefficients");
  for (uint32_t j = 0; j < m; j++)
    if (n && (!d_a[j] || !d_q[j] || d_a[j] == d_q[j])) return fail(SG_ERR_INVALID, "sg_fr_kate_division_batch: bad vector");
  if (m == 0 || n == 0) return SG_OK;
  LOCKED_CTX();
  hipStream_t s = pick_stream(stream);
  uint8_t* d_tmp = nullptr;
  hipError_t e = scratch_for(s, 6, kate_batch_tmp_elems(n, 16) * 32 + 64, &d_tmp);
  if (e != hipSuccess) return hip_fail("kate_division_batch work space", e);
  std::vector<words8> b(m);
  std::memcpy(b.data(), points, 32 * (size_t)m);
  // the power tables are computed on the host into page-locked memory of a ring slot and copied from there: asynchronous (the
  // call used to wait for the whole stream so that a local staging buffer could die -- 0.16 ms of a proof with the device idle
  // behind it); a slot is reused once the kernels that read it have run (an event; normally long complete)
  Context::BlobSlot& slot = g_ctx->kate_ring[g_ctx->kate_next++ % Context::KATE_RING];
  if (!slot.ev) e = hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming);
  else if (hipEventQuery(slot.ev) != hipSuccess) e = host_wait_event(slot.ev);
  if (e == hipSuccess && !slot.host) {
    const size_t want = kate_batch_powers_bytes(16);
    e = hipHostMalloc(reinterpret_cast<void**>(&slot.host), want, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&slot.dev), want);
    if (e == hipSuccess) slot.cap = want;
  }
  if (e == hipSuccess)
    e = poly_kate_division_batch(reinterpret_cast<const fp_words* const*>(d_a), n, b.data(), m, reinterpret_cast<fp_words* const*>(d_q),
                                 slot.host, slot.dev, reinterpret_cast<fp_words*>(d_tmp), s);
  if (e == hipSuccess) e = hipEventRecord(slot.ev, s);
  if (e != hipSuccess) return hip_fail("kate_division_batch", e);
  return SG_OK;
}
// out[i] = sum_j coeffs[j] * polys[j][i]: the random linear combinations of SHPLONK / multi-open
int sg_fr_lincomb_dev(const void* const* d_polys, const uint8_t* coeffs, uint32_t m, size_t n, void* d_out, void* stream) {
  if (!d_polys || !coeffs || (n && !d_out)) return fail(SG_ERR_INVALID, "sg_fr_lincomb: null argument");
  if (m == 0 || m > LINCOMB_MAX) return fail(SG_ERR_INVALID, "sg_fr_lincomb: between 1 and 32 polynomials");
  if (n >= (1ull << 32)) return fail(SG_ERR_INVALID, "sg_fr_lincomb: vector too long");
  for (uint32_t j = 0; j < m; j++)
    if (n && !d_polys[j]) return fail(SG_ERR_INVALID, "sg_fr_lincomb: null polynomial");
  LOCKED_CTX();
  words8 cw[LINCOMB_MAX];
  std::memcpy(cw, coeffs, 32 * (size_t)m);
  hipError_t e = poly_lincomb(reinterpret_cast<const fp_words* const*>(d_polys), cw, m, n, static_cast<fp_words*>(d_out),
                              pick_stream(stream));
  if (e != hipSuccess) return hip_fail("lincomb", e);
  return SG_OK;
}

int sg_fr_lincomb_low_dev(const void* const* d_polys, const uint8_t* coeffs, uint32_t m, size_t n, const uint8_t* low, uint32_t n_low,
                          void* d_out, void* stream) {
  if ((m && (!d_polys || !coeffs)) || (n && !d_out) || (n_low && !low)) return fail(SG_ERR_INVALID, "sg_fr_lincomb_low: null argument");
  if (m > LINCOMB_MAX) return fail(SG_ERR_INVALID, "sg_fr_lincomb_low: at most 32 polynomials");
  if (n >= (1ull << 32) || n_low > LINCOMB_LOW_MAX || n_low > n) return fail(SG_ERR_INVALID, "sg_fr_lincomb_low: bad length");
  for (uint32_t j = 0; j < m; j++)
    if (n && !d_polys[j]) return fail(SG_ERR_INVALID, "sg_fr_lincomb_low: null polynomial");
  LOCKED_CTX();
  words8 cw[LINCOMB_MAX], lw[LINCOMB_LOW_MAX];
  if (m) std::memcpy(cw, coeffs, 32 * (size_t)m);
  if (n_low) std::memcpy(lw, low, 32 * (size_t)n_low);
  hipError_t e = poly_lincomb(reinterpret_cast<const fp_words* const*>(d_polys), cw, m, n, static_cast<fp_words*>(d_out),
                              pick_stream(stream), lw, n_low);
  if (e != hipSuccess) return hip_fail("lincomb", e);
  return SG_OK;
}

int sg_fr_lincomb_sets_dev(const void* const* d_polys, const uint8_t* coeffs, const uint32_t* set_sizes, uint32_t n_sets, size_t n,
                           const uint8_t* lows, const uint32_t* n_lows, void* const* d_outs, void* stream) {
  if (!d_polys || !coeffs || !set_sizes || !d_outs || n_sets == 0 || n_sets > LINCOMB_SETS_MAX) return fail(SG_ERR_INVALID, "sg_fr_lincomb_sets: bad argument");
  if (n >= (1ull << 32)) return fail(SG_ERR_INVALID, "sg_fr_lincomb_sets: vector too long");
  uint32_t first[LINCOMB_SETS_MAX + 1] = {0}, nl[LINCOMB_SETS_MAX] = {0};
  for (uint32_t s = 0; s < n_sets; s++) {
    if (set_sizes[s] > LINCOMB_MAX) return fail(SG_ERR_INVALID, "sg_fr_lincomb_sets: at most 32 polynomials per combination");
    first[s + 1] = first[s] + set_sizes[s];
    nl[s] = n_lows ? n_lows[s] : 0;
    if (nl[s] > LINCOMB_SETS_LOW || nl[s] > n || (nl[s] && !lows)) return fail(SG_ERR_INVALID, "sg_fr_lincomb_sets: at most 4 low coefficients per combination");
    if (n && !d_outs[s]) return fail(SG_ERR_INVALID, "sg_fr_lincomb_sets: null output");
  }
  if (first[n_sets] > LINCOMB_SETS_POLYS) return fail(SG_ERR_INVALID, "sg_fr_lincomb_sets: at most 48 polynomials in all");
  for (uint32_t j = 0; j < first[n_sets]; j++)
    if (n && !d_polys[j]) return fail(SG_ERR_INVALID, "sg_fr_lincomb_sets: null polynomial");
  LOCKED_CTX();
  words8 cw[LINCOMB_SETS_POLYS], lw[LINCOMB_SETS_MAX * LINCOMB_SETS_LOW];
  std::memcpy(cw, coeffs, 32 * (size_t)first[n_sets]);
  std::memset(lw, 0, sizeof lw);
  if (lows) std::memcpy(lw, lows, 32 * (size_t)n_sets * LINCOMB_SETS_LOW);
  hipError_t e = poly_lincomb_sets(reinterpret_cast<const fp_words* const*>(d_polys), cw, first, n_sets, n, lw, nl,
                                   reinterpret_cast<fp_words* const*>(d_outs), pick_stream(stream));
  if (e != hipSuccess) return hip_fail("lincomb sets", e);
  return SG_OK;
}

// ------------------------------------------------------------------ quotient numerator (evaluate_h, generic parts)
// cosets = 0: the whole extended domain (row i = zeta omega_ext^i); cosets = c: coset-major arrays, block b = the 2^k rows
// of the coset zeta omega_ext^b H
static int quotient_permutation_impl(void* d_values, const void* const* d_z, uint32_t nsets, const void* const* d_cols,
                                     const void* const* d_sigma, uint32_t ncols, uint32_t chunk_len, const void* d_l0,
                                     const void* d_l_last, const void* d_l_active, const uint8_t beta[32],
                                     const uint8_t gamma[32], const uint8_t y[32], uint32_t k, uint32_t ext_k,
                                     uint32_t last_rotation_abs, uint32_t cosets, void* stream) {
  if (!d_values || !d_z || !d_cols || !d_sigma || !d_l0 || !d_l_last || !d_l_active || !beta || !gamma || !y)
    return fail(SG_ERR_INVALID, "sg_quotient_permutation: null argument");
  if (k == 0 || ext_k < k || ext_k > 28 || nsets == 0 || nsets > QUOT_MAX_SETS || ncols == 0 ||
      ncols > QUOT_MAX_COLS || chunk_len == 0 || chunk_len > 11 ||
      (size_t)nsets * chunk_len < ncols || (size_t)(nsets - 1) * chunk_len >= ncols ||
      last_rotation_abs >= (1u << k))
    return fail(SG_ERR_INVALID, "sg_quotient_permutation: bad shape");
  LOCKED_CTX();
  QuotPermArgs a;
  std::memset(&a, 0, sizeof(a));
  a.values = static_cast<fp_words*>(d_values);
  for (uint32_t i = 0; i < nsets; i++) {
    if (!d_z[i]) return fail(SG_ERR_INVALID, "sg_quotient_permutation: null z");
    a.z[i] = static_cast<const fp_words*>(d_z[i]);
  }
  for (uint32_t i = 0; i < ncols; i++) {
    if (!d_cols[i] || !d_sigma[i]) return fail(SG_ERR_INVALID, "sg_quotient_permutation: null column");
    a.cols[i] = static_cast<const fp_words*>(d_cols[i]);
    a.sigma[i] = static_cast<const fp_words*>(d_sigma[i]);
  }
  a.l0 = static_cast<const fp_words*>(d_l0);
  a.l_last = static_cast<const fp_words*>(d_l_last);
  a.l_active = static_cast<const fp_words*>(d_l_active);
  a.nsets = nsets; a.ncols = ncols; a.chunk_len = chunk_len; a.k = k; a.ext_k = cosets ? k : ext_k;
  a.cosets = cosets;
  a.last_rot_abs = last_rotation_abs;
  const DomainConsts *dk, *de;
  TRY(get_consts(k, &dk));
  TRY(get_consts(ext_k, &de));
  std::memcpy(a.beta, beta, 32); std::memcpy(a.gamma, gamma, 32); std::memcpy(a.y, y, 32);
  std::memcpy(a.delta, DELTA_M, 32); std::memcpy(a.zeta, &dk->zeta, 32); std::memcpy(a.omega_ext, &de->omega, 32);
  if (cosets) {   // every block is the 2^k domain shifted by c_b = zeta omega_ext^b
    const Context::CosetTables* t;
    TRY(coset_tables_for(k, ext_k, cosets, &t));
    for (uint32_t b = 0; b < cosets; b++) std::memcpy(a.shift[b], &t->shift[b], 32);
    std::memcpy(a.omega_ext, &dk->omega, 32);
  }
  hipStream_t s = pick_stream(stream);
  fp_words* pw = nullptr;
  hipError_t e = g_ctx->ntt.local_twiddles(cosets ? dk->omega : de->omega, 9, s, &pw);   // omega_ext^t, t < 256
  if (e != hipSuccess) return hip_fail("quotient twiddles", e);
  a.pow_lo = pw;
  e = quotient_permutation(a, s);
  if (e != hipSuccess) return hip_fail("quotient_permutation", e);
  return SG_OK;
}
int sg_quotient_permutation_dev(void* d_values, const void* const* d_z, uint32_t nsets, const void* const* d_cols,
                                const void* const* d_sigma, uint32_t ncols, uint32_t chunk_len, const void* d_l0,
                                const void* d_l_last, const void* d_l_active, const uint8_t beta[32],
                                const uint8_t gamma[32], const uint8_t y[32], uint32_t k, uint32_t ext_k,
                                uint32_t last_rotation_abs, void* stream) {
  return quotient_permutation_impl(d_values, d_z, nsets, d_cols, d_sigma, ncols, chunk_len, d_l0, d_l_last, d_l_active, beta, gamma, y, k,
                                   ext_k, last_rotation_abs, 0, stream);
}
int sg_quotient_permutation_cosets_dev(void* d_values, const void* const* d_z, uint32_t nsets, const void* const* d_cols,
                                       const void* const* d_sigma, uint32_t ncols, uint32_t chunk_len, const void* d_l0,
                                       const void* d_l_last, const void* d_l_active, const uint8_t beta[32],
                                       const uint8_t gamma[32], const uint8_t y[32], uint32_t k, uint32_t ext_k,
                                       uint32_t n_cosets, uint32_t last_rotation_abs, void* stream) {
  if (!coset_shape_ok(k, ext_k, n_cosets)) return fail(SG_ERR_INVALID, "sg_quotient_permutation_cosets: bad shape");
  return quotient_permutation_impl(d_values, d_z, nsets, d_cols, d_sigma, ncols, chunk_len, d_l0, d_l_last, d_l_active, beta, gamma, y, k,
                                   ext_k, last_rotation_abs, n_cosets, stream);
}
static int quotient_lookup_impl(void* d_values, const void* d_z, const void* d_permuted_input, const void* d_permuted_table,
                                const void* d_input, const void* d_table, const void* d_l0, const void* d_l_last,
                                const void* d_l_active, const uint8_t beta[32], const uint8_t gamma[32], const uint8_t y[32],
                                uint32_t k, uint32_t ext_k, uint32_t cosets, void* stream) {
  if (!d_values || !d_z || !d_permuted_input || !d_permuted_table || !d_input || !d_table || !d_l0 || !d_l_last ||
      !d_l_active || !beta || !gamma || !y)
    return fail(SG_ERR_INVALID, "sg_quotient_lookup: null argument");
  if (k == 0 || ext_k < k || ext_k > 28) return fail(SG_ERR_INVALID, "sg_quotient_lookup: bad shape");
  LOCKED_CTX();
  QuotLookupArgs a;
  a.values = static_cast<fp_words*>(d_values);
  a.z = static_cast<const fp_words*>(d_z);
  a.permuted_input = static_cast<const fp_words*>(d_permuted_input);
  a.permuted_table = static_cast<const fp_words*>(d_permuted_table);
  a.input = static_cast<const fp_words*>(d_input);
  a.table = static_cast<const fp_words*>(d_table);
  a.l0 = static_cast<const fp_words*>(d_l0);
  a.l_last = static_cast<const fp_words*>(d_l_last);
  a.l_active = static_cast<const fp_words*>(d_l_active);
  a.k = k; a.ext_k = cosets ? k : ext_k;
  a.cosets = cosets;
  std::memcpy(a.beta, beta, 32); std::memcpy(a.gamma, gamma, 32); std::memcpy(a.y, y, 32);
  hipError_t e = quotient_lookup(a, pick_stream(stream));
  if (e != hipSuccess) return hip_fail("quotient_lookup", e);
  return SG_OK;
}
int sg_quotient_lookup_dev(void* d_values, const void* d_z, const void* d_permuted_input, const void* d_permuted_table,
                           const void* d_input, const void* d_table, const void* d_l0, const void* d_l_last,
                           const void* d_l_active, const uint8_t beta[32], const uint8_t gamma[32], const uint8_t y[32],
                           uint32_t k, uint32_t ext_k, void* stream) {
  return quotient_lookup_impl(d_values, d_z, d_permuted_input, d_permuted_table, d_input, d_table, d_l0, d_l_last, d_l_active, beta, gamma,
                              y, k, ext_k, 0, stream);
}
int sg_quotient_lookup_cosets_dev(void* d_values, const void* d_z, const void* d_permuted_input, const void* d_permuted_table,
                                  const void* d_input, const void* d_table, const void* d_l0, const void* d_l_last,
                                  const void* d_l_active, const uint8_t beta[32], const uint8_t gamma[32], const uint8_t y[32],
                                  uint32_t k, uint32_t n_cosets, void* stream) {
  if (n_cosets == 0 || n_cosets > QUOT_MAX_COSETS) return fail(SG_ERR_INVALID, "sg_quotient_lookup_cosets: bad shape");
  return quotient_lookup_impl(d_values, d_z, d_permuted_input, d_permuted_table, d_input, d_table, d_l0, d_l_last, d_l_active, beta, gamma,
                              y, k, k, n_cosets, stream);
}

// the lowered program of a graph, from the lane's cache (compiled on first sight), with its constant table refreshed for this
// call: constants ++ challenges ++ beta, gamma, theta, y (compile_gates' order).  The caller holds the lane.
static int gate_program_for(const sg_graph* graph, uint32_t n_fixed, uint32_t n_advice, uint32_t n_instance, const uint8_t* challenges,
                            uint32_t n_challenges, const uint8_t beta[32], const uint8_t gamma[32], const uint8_t theta[32],
                            const uint8_t y[32], GateProgram** out) {
  // the lowered program depends on the graph's structure only (constants / challenges are a table refreshed per call):
  // cached under the structure itself.  A prover sends the same two programs proof after proof, so the lane's most recent
  // hits are tried first with one memcmp each (the structure of the reference circuit's gate program is 100+ KB: hashing
  // it byte by byte cost 0.3 ms of host time per proof, with the device idle behind it)
  std::vector<uint8_t> sig;
  {
    const uint32_t hdr[8] = {graph->n_constants, graph->n_rotations, graph->n_calculations, graph->n_horner_parts, n_fixed,
                             n_advice, n_instance, n_challenges};
    const size_t parts[4] = {sizeof hdr, graph->rotations ? sizeof(int32_t) * graph->n_rotations : 0,
                             graph->calculations ? sizeof(sg_calculation) * graph->n_calculations : 0,
                             graph->horner_parts ? sizeof(sg_value_source) * graph->n_horner_parts : 0};
    const void* src[4] = {hdr, graph->rotations, graph->calculations, graph->horner_parts};
    sig.resize(parts[0] + parts[1] + parts[2] + parts[3]);
    size_t at = 0;
    for (int i = 0; i < 4; i++) {
      if (parts[i]) std::memcpy(sig.data() + at, src[i], parts[i]);
      at += parts[i];
    }
  }
  auto hit = g_ctx->gate_cache.end();
  for (uint64_t recent : g_ctx->gate_recent) {
    auto it = g_ctx->gate_cache.find(recent);
    if (it != g_ctx->gate_cache.end() && it->second.signature == sig) {
      hit = it;
      break;
    }
  }
  uint64_t key = 1469598103934665603ull;
  if (hit == g_ctx->gate_cache.end()) {
    size_t i = 0;
    for (; i + 8 <= sig.size(); i += 8) {
      uint64_t w;
      std::memcpy(&w, sig.data() + i, 8);
      key = (key ^ w) * 1099511628211ull;
    }
    for (; i < sig.size(); i++) key = (key ^ sig[i]) * 1099511628211ull;
    hit = g_ctx->gate_cache.find(key);
    if (hit != g_ctx->gate_cache.end() && hit->second.signature != sig) {  // 64-bit collision: recompile
      g_ctx->gate_cache.erase(hit);
      hit = g_ctx->gate_cache.end();
    }
  }
  std::string err = "";
  if (hit == g_ctx->gate_cache.end()) {
    GateProgram fresh;
    err = compile_gates(*graph, n_fixed, n_advice, n_instance, challenges, n_challenges, beta, gamma, theta, y, &fresh);
    if (!err.empty()) return fail(SG_ERR_INVALID, ("sg_quotient_gates: " + err).c_str());
    fresh.signature = std::move(sig);
    if (g_ctx->gate_cache.size() >= 64) g_ctx->gate_cache.clear();
    hit = g_ctx->gate_cache.emplace(key, std::move(fresh)).first;
  }
  {
    bool listed = false;
    for (uint64_t recent : g_ctx->gate_recent) listed = listed || recent == hit->first;
    if (!listed) g_ctx->gate_recent[g_ctx->gate_recent_next++ % 4] = hit->first;
  }
  GateProgram& prog = hit->second;
  if ((graph->n_constants && !graph->constants)) return fail(SG_ERR_INVALID, "sg_quotient_gates: null constants");
  {  // constant table of this call: constants ++ challenges ++ beta, gamma, theta, y (compile_gates' order)
    prog.const_words.clear();
    auto push = [&](const uint8_t* p) {
      uint32_t w[8];
      std::memcpy(w, p, 32);
      prog.const_words.insert(prog.const_words.end(), w, w + 8);
    };
    for (uint32_t i = 0; i < graph->n_constants; i++) push(graph->constants + 32 * (size_t)i);
    for (uint32_t i = 0; i < n_challenges; i++) push(challenges + 32 * (size_t)i);
    push(beta); push(gamma); push(theta); push(y);
  }
  *out = &prog;
  return SG_OK;
}
static int quotient_gates_impl(void* d_values, const sg_graph* graph, const void* const* d_fixed, uint32_t n_fixed,
                               const void* const* d_advice, uint32_t n_advice, const void* const* d_instance,
                               uint32_t n_instance, const uint8_t* challenges, uint32_t n_challenges,
                               const uint8_t beta[32], const uint8_t gamma[32], const uint8_t theta[32], const uint8_t y[32],
                               uint32_t k, uint32_t ext_k, uint32_t cosets, void* stream) {
  if (!d_values || !graph || !beta || !gamma || !theta || !y || (n_fixed && !d_fixed) || (n_advice && !d_advice) ||
      (n_instance && !d_instance) || (n_challenges && !challenges))
    return fail(SG_ERR_INVALID, "sg_quotient_gates: null argument");
  if (k == 0 || ext_k < k || ext_k > 28) return fail(SG_ERR_INVALID, "sg_quotient_gates: bad shape");
  LOCKED_CTX();
  GateProgram* prog_p = nullptr;
  TRY(gate_program_for(graph, n_fixed, n_advice, n_instance, challenges, n_challenges, beta, gamma, theta, y, &prog_p));
  GateProgram& prog = *prog_p;
  if (prog.n_slots > 64) return fail(SG_ERR_INVALID, "sg_quotient_gates: more than 64 simultaneously live values");
  std::vector<const void*> cols;
  for (uint32_t i = 0; i < n_fixed; i++) cols.push_back(d_fixed[i]);
  for (uint32_t i = 0; i < n_advice; i++) cols.push_back(d_advice[i]);
  for (uint32_t i = 0; i < n_instance; i++) cols.push_back(d_instance[i]);
  for (const void* c : cols)
    if (!c) return fail(SG_ERR_INVALID, "sg_quotient_gates: null column");
  hipStream_t s = pick_stream(stream);
  {
    hipError_t ev = hipSuccess;
    if (gates_run_by_value(prog, cols.data(), static_cast<fp_words*>(d_values), k, ext_k, s, cosets, &ev)) {
      if (ev != hipSuccess) return hip_fail("quotient_gates", ev);
      return SG_OK;
    }
  }
  // program + column pointers + constants travel as one small blob.  Ring of page-locked host / device buffer pairs, each
  // guarded by an event recorded after the kernel that reads it: the call is asynchronous (no host wait unless the ring
  // has wrapped onto a launch that is still running)
  const size_t bytes = gates_blob(prog, cols.data(), &g_ctx->gate_blob_host);
  Context::BlobSlot& slot = g_ctx->blob_ring[g_ctx->blob_next++ % Context::BLOB_RING];
  hipError_t e = hipSuccess;
  if (!slot.ev) e = hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming);
  else if (hipEventQuery(slot.ev) != hipSuccess) e = host_wait_event(slot.ev);   // (a query first: waiting on an event that
                                                                                     // has long completed still costs a wake-up, 0.3 ms)
  if (e == hipSuccess && slot.cap < bytes) {
    // (never small: the ring rotates, and a slot sized by a small program would be re-allocated -- two allocations, 0.25 ms
    // with the device idle -- the first time the big program of the same prover comes round to it)
    const size_t want = std::max<size_t>(bytes + bytes / 2 + 256, (size_t)256 << 10);
    if (slot.host) (void)hipHostFree(slot.host);
    retire_device_memory(slot.dev);
    slot.host = nullptr;
    slot.dev = nullptr;
    slot.cap = 0;
    e = hipHostMalloc(reinterpret_cast<void**>(&slot.host), want, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&slot.dev), want);
    if (e == hipSuccess) slot.cap = want;
  }
  if (e == hipSuccess) {
    std::memcpy(slot.host, g_ctx->gate_blob_host.data(), bytes);
    e = hipMemcpyAsync(slot.dev, slot.host, bytes, hipMemcpyHostToDevice, s);
  }
  if (e == hipSuccess) e = gates_run(prog, slot.dev, static_cast<fp_words*>(d_values), k, ext_k, s, cosets);
  if (e == hipSuccess) e = hipEventRecord(slot.ev, s);
  if (e != hipSuccess) return hip_fail("quotient_gates", e);
  return SG_OK;
}
int sg_quotient_gates_dev(void* d_values, const sg_graph* graph, const void* const* d_fixed, uint32_t n_fixed,
                          const void* const* d_advice, uint32_t n_advice, const void* const* d_instance,
                          uint32_t n_instance, const uint8_t* challenges, uint32_t n_challenges,
                          const uint8_t beta[32], const uint8_t gamma[32], const uint8_t theta[32], const uint8_t y[32],
                          uint32_t k, uint32_t ext_k, void* stream) {
  return quotient_gates_impl(d_values, graph, d_fixed, n_fixed, d_advice, n_advice, d_instance, n_instance, challenges, n_challenges, beta,
                             gamma, theta, y, k, ext_k, 0, stream);
}
int sg_quotient_gates_cosets_dev(void* d_values, const sg_graph* graph, const void* const* d_fixed, uint32_t n_fixed,
                                 const void* const* d_advice, uint32_t n_advice, const void* const* d_instance,
                                 uint32_t n_instance, const uint8_t* challenges, uint32_t n_challenges,
                                 const uint8_t beta[32], const uint8_t gamma[32], const uint8_t theta[32], const uint8_t y[32],
                                 uint32_t k, uint32_t n_cosets, void* stream) {
  if (n_cosets == 0 || n_cosets > QUOT_MAX_COSETS) return fail(SG_ERR_INVALID, "sg_quotient_gates_cosets: bad shape");
  return quotient_gates_impl(d_values, graph, d_fixed, n_fixed, d_advice, n_advice, d_instance, n_instance, challenges, n_challenges, beta,
                             gamma, theta, y, k, k, n_cosets, stream);
}
// halo2's evaluate_h in one call: values <- gates, then the permutation argument, then the lookup argument (input expression
// evaluated on the way).  One fused kernel when the two programs are known ahead of time (csrc/numerator.hip), otherwise the
// separate kernels one after the other -- the same words either way.
static std::atomic<int> g_numerator_fused{1};   // "quotient.fused_numerator": 0 forces the separate kernels (A-B, tests)
int sg_quotient_numerator_cosets_dev(void* d_values, const sg_graph* gates, const sg_graph* lookup_input, const void* const* d_fixed,
                                     uint32_t n_fixed, const void* const* d_advice, uint32_t n_advice, const void* const* d_instance,
                                     uint32_t n_instance, const uint8_t* challenges, uint32_t n_challenges, const void* const* d_z,
                                     uint32_t nsets, const void* const* d_perm_cols, const void* const* d_sigma, uint32_t ncols,
                                     uint32_t chunk_len, const void* d_l0, const void* d_l_last, const void* d_l_active,
                                     const void* d_lookup_z, const void* d_permuted_input, const void* d_permuted_table,
                                     const void* d_table, void* d_input_work, const uint8_t beta[32], const uint8_t gamma[32],
                                     const uint8_t theta[32], const uint8_t y[32], uint32_t k, uint32_t ext_k, uint32_t n_cosets,
                                     uint32_t last_rotation_abs, void* stream) {
  if (!d_values || !gates || !lookup_input || !d_z || !d_perm_cols || !d_sigma || !d_l0 || !d_l_last || !d_l_active || !d_lookup_z ||
      !d_permuted_input || !d_permuted_table || !d_table || !beta || !gamma || !theta || !y || (n_fixed && !d_fixed) ||
      (n_advice && !d_advice) || (n_instance && !d_instance) || (n_challenges && !challenges))
    return fail(SG_ERR_INVALID, "sg_quotient_numerator_cosets: null argument");
  if (!coset_shape_ok(k, ext_k, n_cosets) || n_cosets > QUOT_MAX_COSETS) return fail(SG_ERR_INVALID, "sg_quotient_numerator_cosets: bad shape");
  if (nsets == 0 || nsets > QUOT_MAX_SETS || ncols == 0 || ncols > QUOT_MAX_COLS || chunk_len == 0 || (ncols + chunk_len - 1) / chunk_len != nsets)
    return fail(SG_ERR_INVALID, "sg_quotient_numerator_cosets: bad permutation shape");
  bool fused = false;
  if (g_numerator_fused.load() && n_fixed + n_advice + n_instance <= NUM_MAX_COLS) {
    LOCKED_CTX();
    if (g_ctx->gate_cache.size() >= 62) g_ctx->gate_cache.clear();   // neither look-up below may evict the other's program
    GateProgram *pg = nullptr, *pi = nullptr;
    const uint8_t none[32] = {0};
    TRY(gate_program_for(gates, n_fixed, n_advice, n_instance, challenges, n_challenges, beta, gamma, theta, y, &pg));
    TRY(gate_program_for(lookup_input, n_fixed, n_advice, n_instance, none, 0, beta, gamma, theta, y, &pi));
    if (numerator_fused_available(*pg, *pi)) {
      NumeratorArgs a;
      std::memset(&a, 0, sizeof a);
      a.values = static_cast<fp_words*>(d_values);
      uint32_t c = 0;
      for (uint32_t i = 0; i < n_fixed; i++) a.cols[c++] = static_cast<const fp_words*>(d_fixed[i]);
      for (uint32_t i = 0; i < n_advice; i++) a.cols[c++] = static_cast<const fp_words*>(d_advice[i]);
      for (uint32_t i = 0; i < n_instance; i++) a.cols[c++] = static_cast<const fp_words*>(d_instance[i]);
      for (uint32_t i = 0; i < c; i++)
        if (!a.cols[i]) return fail(SG_ERR_INVALID, "sg_quotient_numerator_cosets: null column");
      QuotPermArgs& pa = a.perm;
      for (uint32_t i = 0; i < nsets; i++) {
        if (!d_z[i]) return fail(SG_ERR_INVALID, "sg_quotient_numerator_cosets: null z");
        pa.z[i] = static_cast<const fp_words*>(d_z[i]);
      }
      for (uint32_t i = 0; i < ncols; i++) {
        if (!d_perm_cols[i] || !d_sigma[i]) return fail(SG_ERR_INVALID, "sg_quotient_numerator_cosets: null permutation column");
        pa.cols[i] = static_cast<const fp_words*>(d_perm_cols[i]);
        pa.sigma[i] = static_cast<const fp_words*>(d_sigma[i]);
      }
      pa.l0 = static_cast<const fp_words*>(d_l0);
      pa.l_last = static_cast<const fp_words*>(d_l_last);
      pa.l_active = static_cast<const fp_words*>(d_l_active);
      pa.nsets = nsets; pa.ncols = ncols; pa.chunk_len = chunk_len; pa.k = k; pa.ext_k = k; pa.cosets = n_cosets;
      pa.last_rot_abs = last_rotation_abs;
      const DomainConsts* dk;
      TRY(get_consts(k, &dk));
      std::memcpy(pa.beta, beta, 32); std::memcpy(pa.gamma, gamma, 32); std::memcpy(pa.y, y, 32);
      std::memcpy(pa.delta, DELTA_M, 32); std::memcpy(pa.zeta, &dk->zeta, 32); std::memcpy(pa.omega_ext, &dk->omega, 32);
      const Context::CosetTables* t;
      TRY(coset_tables_for(k, ext_k, n_cosets, &t));
      for (uint32_t b = 0; b < n_cosets; b++) std::memcpy(pa.shift[b], &t->shift[b], 32);
      hipStream_t s = pick_stream(stream);
      fp_words* pw = nullptr;
      hipError_t e = g_ctx->ntt.local_twiddles(dk->omega, 9, s, &pw);   // omega^t, t < 256
      if (e != hipSuccess) return hip_fail("quotient twiddles", e);
      pa.pow_lo = pw;
      QuotLookupArgs& la = a.look;
      la.z = static_cast<const fp_words*>(d_lookup_z);
      la.permuted_input = static_cast<const fp_words*>(d_permuted_input);
      la.permuted_table = static_cast<const fp_words*>(d_permuted_table);
      la.table = static_cast<const fp_words*>(d_table);
      la.l0 = pa.l0; la.l_last = pa.l_last; la.l_active = pa.l_active;
      la.k = k; la.ext_k = k; la.cosets = n_cosets;
      std::memcpy(la.beta, beta, 32); std::memcpy(la.gamma, gamma, 32); std::memcpy(la.y, y, 32);
      e = numerator_fused(*pg, *pi, a, s);
      if (e != hipSuccess) return hip_fail("quotient numerator", e);
      fused = true;
    }
  }
  if (fused) return SG_OK;
  // any other pair of programs: the blocks one after the other over `values` (zeroed: a fresh numerator)
  const size_t rows = (size_t)n_cosets << k;
  void* input_work = d_input_work;
  if (!input_work) {
    LOCKED_CTX();
    uint8_t* w = nullptr;
    hipError_t e = scratch_for(pick_stream(stream), 8, rows * 32, &w);
    if (e != hipSuccess) return hip_fail("numerator work space", e);
    input_work = w;
  }
  {
    LOCKED_CTX();
    CHECK_HIP(hipMemsetAsync(d_values, 0, rows * 32, pick_stream(stream)), "memset");
  }
  const uint8_t none[32] = {0};
  TRY(sg_quotient_gates_cosets_dev(d_values, gates, d_fixed, n_fixed, d_advice, n_advice, d_instance, n_instance, challenges, n_challenges, beta,
                                   gamma, theta, y, k, n_cosets, stream));
  TRY(sg_quotient_permutation_cosets_dev(d_values, d_z, nsets, d_perm_cols, d_sigma, ncols, chunk_len, d_l0, d_l_last, d_l_active, beta, gamma, y,
                                         k, ext_k, n_cosets, last_rotation_abs, stream));
  TRY(sg_quotient_gates_cosets_dev(input_work, lookup_input, d_fixed, n_fixed, d_advice, n_advice, d_instance, n_instance, none, 0, beta, gamma,
                                   theta, y, k, n_cosets, stream));
  return sg_quotient_lookup_cosets_dev(d_values, d_lookup_z, d_permuted_input, d_permuted_table, input_work, d_table, d_l0, d_l_last, d_l_active,
                                       beta, gamma, y, k, n_cosets, stream);
}

// how the interpreter would run a program: instructions and simultaneously live values (LDS slots per row; 8 or fewer keep
// two workgroups of 256 rows per CU).  Host-only: no device is touched.
int sg_gates_program_info(const sg_graph* graph, uint32_t n_fixed, uint32_t n_advice, uint32_t n_instance, uint32_t n_challenges,
                          uint32_t* n_ops_out, uint32_t* n_slots_out) {
  if (!graph || !n_ops_out || !n_slots_out) return fail(SG_ERR_INVALID, "sg_gates_program_info: null argument");
  std::vector<uint8_t> zeros(32 * (size_t)std::max<uint32_t>(1, n_challenges), 0);
  GateProgram prog;
  const std::string err = compile_gates(*graph, n_fixed, n_advice, n_instance, zeros.data(), n_challenges, zeros.data(), zeros.data(),
                                        zeros.data(), zeros.data(), &prog);
  if (!err.empty()) return fail(SG_ERR_INVALID, ("sg_gates_program_info: " + err).c_str());
  *n_ops_out = (uint32_t)prog.ops.size();
  *n_slots_out = prog.n_slots;
  return SG_OK;
}

// the lowered program itself, for tooling (tools/gen_gates_programs.py writes the ahead-of-time instantiations of the reference
// circuit's programs from it) and tests: words_out = [n_slots, result_kind, result_index, n_ops, then (w0, dst, a, b) per
// instruction].  *n_words_out is the size needed; nothing is written beyond cap_words.  Host only.
int sg_gates_program_words(const sg_graph* graph, uint32_t n_fixed, uint32_t n_advice, uint32_t n_instance, uint32_t n_challenges,
                           uint32_t* words_out, uint32_t cap_words, uint32_t* n_words_out) {
  if (!graph || !n_words_out || (cap_words && !words_out)) return fail(SG_ERR_INVALID, "sg_gates_program_words: null argument");
  std::vector<uint8_t> zeros(32 * (size_t)std::max<uint32_t>(1, n_challenges), 0);
  GateProgram prog;
  const std::string err = compile_gates(*graph, n_fixed, n_advice, n_instance, zeros.data(), n_challenges, zeros.data(), zeros.data(),
                                        zeros.data(), zeros.data(), &prog);
  if (!err.empty()) return fail(SG_ERR_INVALID, ("sg_gates_program_words: " + err).c_str());
  std::vector<uint32_t> w = {prog.n_slots, prog.result_kind, prog.result_index, (uint32_t)prog.ops.size()};
  for (const GateOp& o : prog.ops) { w.push_back(o.w0); w.push_back(o.dst); w.push_back(o.a); w.push_back(o.b); }
  *n_words_out = (uint32_t)w.size();
  if (cap_words >= w.size()) std::memcpy(words_out, w.data(), 4 * w.size());
  return SG_OK;
}

// ------------------------------------------------------------------ keygen's circuit side
// What `keygen_vk` / `keygen_pk` need of `MstInclusionCircuit::synthesize` over 2^k rows [REF zk_prover/src/circuits/
// merkle_sum_tree.rs:228-520 replayed over halo2's SimpleFloorPlanner: include/summa_circuit.hpp]: the 11 fixed columns (round
// constants, range table, compressed selectors, constants) and the 6 permutation columns sigma_c[row] = the label delta^c' omega^row'
// of the cell (c, row) is copy-constrained to.  Host only (no device needed); Montgomery words, column-major.
int sg_mst_inclusion_keygen_columns(uint32_t k, uint32_t levels, uint32_t n_currencies, uint32_t n_bytes, uint8_t* fixed_out,
                                    uint8_t* sigma_out, uint32_t* rows_used_out) {
  if (!fixed_out || !sigma_out || k < 4 || k > 25 || levels == 0 || levels > 48 || n_currencies == 0 || n_currencies > 16 || n_bytes == 0 ||
      n_bytes > 31)
    return fail(SG_ERR_INVALID, "sg_mst_inclusion_keygen_columns: bad argument");
  try {
    using summa::prover::Fr;
    const summa::circuit::FloorPlan fp(k, levels, n_currencies, n_bytes);
    const size_t n = (size_t)1 << k;
    if (fp.rows_used + summa::circuit::BLINDING_FACTORS + 1 > n) return fail(SG_ERR_INVALID, "sg_mst_inclusion_keygen_columns: not enough rows");
    static const uint8_t root_2_28[32] = {0x03, 0xdd, 0xb9, 0xf5, 0x16, 0x6d, 0x18, 0xb7, 0x98, 0x86, 0x5e, 0xa9, 0x3d, 0xd3, 0x1f, 0x74,
                                          0x32, 0x15, 0xcf, 0x6d, 0xd3, 0x93, 0x29, 0xc8, 0xd3, 0x4f, 0x1e, 0xd9, 0x60, 0xc3, 0x7c, 0x9c};
    Fr omega = Fr::from_be_bytes_reduced(root_2_28);
    for (uint32_t i = k; i < 28; i++) omega = omega * omega;
    for (uint32_t c = 0; c < summa::circuit::NUM_FIXED; c++) std::memcpy(fixed_out + 32 * n * c, fp.fixed[c].data(), 32 * n);
    const auto sigma = fp.sigma(omega);
    for (uint32_t c = 0; c < summa::circuit::NUM_PERM; c++) std::memcpy(sigma_out + 32 * n * c, sigma[c].data(), 32 * n);
    if (rows_used_out) *rows_used_out = fp.rows_used;
  } catch (const std::exception& ex) {
    return fail(SG_ERR_INVALID, (std::string("sg_mst_inclusion_keygen_columns: ") + ex.what()).c_str());
  }
  return SG_OK;
}

// ------------------------------------------------------------------ witness side (Merkle sum tree)
int sg_mst_leaves_dev(const void* d_usernames, const void* d_balances, size_t n, uint32_t n_currencies,
                      void* d_hashes, void* stream) {
  if (n && (!d_usernames || !d_balances || !d_hashes)) return fail(SG_ERR_INVALID, "sg_mst_leaves: null argument");
  if (n_currencies == 0 || n_currencies > 64 || n >= (1ull << 32)) return fail(SG_ERR_INVALID, "sg_mst_leaves: bad size");
  LOCKED_CTX();
  hipStream_t s = pick_stream(stream);
  hipError_t e = g_ctx->witness.init(g_ctx->stream);
  if (e == hipSuccess)
    e = g_ctx->witness.leaves(static_cast<const fp_words*>(d_usernames), static_cast<const fp_words*>(d_balances), n,
                              n_currencies, static_cast<fp_words*>(d_hashes), s);
  if (e != hipSuccess) return hip_fail("mst leaves", e);
  return SG_OK;
}
int sg_mst_level_dev(const void* d_child_hashes, const void* d_child_balances, size_t n_parents, uint32_t n_currencies,
                     void* d_hashes, void* d_balances, void* stream) {
  if (n_parents && (!d_child_hashes || !d_child_balances || !d_hashes || !d_balances))
    return fail(SG_ERR_INVALID, "sg_mst_level: null argument");
  if (n_currencies == 0 || n_currencies > 64 || n_parents >= (1ull << 31)) return fail(SG_ERR_INVALID, "sg_mst_level: bad size");
  LOCKED_CTX();
  hipError_t e = g_ctx->witness.init(g_ctx->stream);
  if (e == hipSuccess)
    e = g_ctx->witness.level(static_cast<const fp_words*>(d_child_hashes), static_cast<const fp_words*>(d_child_balances),
                             n_parents, n_currencies, static_cast<fp_words*>(d_hashes),
                             static_cast<fp_words*>(d_balances), pick_stream(stream));
  if (e != hipSuccess) return hip_fail("mst level", e);
  return SG_OK;
}
// whole tree: node arrays are level-major (2^depth leaves, then 2^(depth-1) parents, ..., the root)
int sg_mst_build_dev(const void* d_usernames, const void* d_leaf_balances, uint32_t depth, uint32_t n_currencies,
                     void* d_node_hashes, void* d_node_balances, void* stream) {
  if (!d_usernames || !d_leaf_balances || !d_node_hashes || !d_node_balances || depth > 30)
    return fail(SG_ERR_INVALID, "sg_mst_build: bad argument");
  const size_t n = (size_t)1 << depth;
  uint8_t* h = static_cast<uint8_t*>(d_node_hashes);
  uint8_t* b = static_cast<uint8_t*>(d_node_balances);
  int rc = sg_mst_leaves_dev(d_usernames, d_leaf_balances, n, n_currencies, h, stream);
  if (rc != SG_OK) return rc;
  {
    LOCKED_CTX();
    CHECK_HIP(hipMemcpyAsync(b, d_leaf_balances, n * n_currencies * 32, hipMemcpyDeviceToDevice, pick_stream(stream)),
              "mst balances");
  }
  size_t off = 0;
  for (size_t m = n >> 1; m >= 1; m >>= 1) {
    const size_t child = off, parent = off + 2 * m;
    rc = sg_mst_level_dev(h + 32 * child, b + 32 * child * n_currencies, m, n_currencies, h + 32 * parent,
                          b + 32 * parent * n_currencies, stream);
    if (rc != SG_OK) return rc;
    off = parent;
  }
  return SG_OK;
}

// Circuit::synthesize on the device for users of a device-resident tree (witness.hip: the program is the floor plan)
int sg_mst_inclusion_witness_dev(const void* d_program, uint32_t n_items, uint32_t n_absorbs, const void* d_usernames,
                                 const void* d_node_hashes, const void* d_node_balances, uint32_t depth, uint32_t n_currencies,
                                 const void* d_user_indices, uint32_t n_users, void* d_advice, uint64_t rows, void* stream) {
  if (!d_program || !d_usernames || !d_node_hashes || !d_node_balances || !d_user_indices || !d_advice)
    return fail(SG_ERR_INVALID, "sg_mst_inclusion_witness: null argument");
  if (depth > 30 || n_currencies == 0 || n_currencies > 64 || rows == 0 || rows > (1ull << 28) || n_items > (1u << 24) || n_users > 65535)
    return fail(SG_ERR_INVALID, "sg_mst_inclusion_witness: bad size");
  LOCKED_CTX();
  hipStream_t s = pick_stream(stream);
  hipError_t e = g_ctx->witness.init(g_ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(d_advice, 0, (size_t)n_users * 3 * rows * 32, s);
  if (e == hipSuccess)
    e = g_ctx->witness.inclusion_witness(static_cast<const uint32_t*>(d_program), n_items, n_absorbs,
                                         static_cast<const fp_words*>(d_usernames), static_cast<const fp_words*>(d_node_hashes),
                                         static_cast<const fp_words*>(d_node_balances), depth, n_currencies,
                                         static_cast<const uint32_t*>(d_user_indices), n_users, static_cast<fp_words*>(d_advice),
                                         (size_t)rows, s);
  if (e != hipSuccess) return hip_fail("mst inclusion witness", e);
  return SG_OK;
}

int sg_set_param(const char* name, int value) {
  if (!name || value < 0) return fail(SG_ERR_INVALID, "sg_set_param: bad argument");
  if (g_depth > 0) return fail(SG_ERR_INVALID, "sg_set_param: not from inside a call");
  const std::string s(name);
  if (s == "commit.combine_wait_us") {   // how long the combiner's runner waits for the other declared threads
    g_comb.wait_us.store(std::min(value, 100000));
    return SG_OK;
  }
  if (s == "commit.combine_target") {
    g_comb.target.store(std::max(1, std::min(value, 32)));
    return SG_OK;
  }
  if (s == "msm.host_chunks") {   // chunks of the host-pointer MSM entry points (0 = by size)
    g_host_chunks.store(std::min(value, 8));
    return SG_OK;
  }
  if (s == "msm.tiny_max") {   // sg_msm_g1 of at most this many points is ONE launch (0: always the engine's pipeline; at most 64)
    g_msm_tiny_max.store(std::min(value, (int)MSM_TINY_MAX));
    return SG_OK;
  }
  if (s == "host.wait_sleep_us") {   // how host threads wait for the device (csrc/host_wait.h): 0 = the runtime's wait, > 0 = poll and sleep
    host_wait_sleep_us().store(std::min(value, 1000));
    return SG_OK;
  }
  if (s == "debug.fail_next_fused_job") {
    g_comb.fail_next.store(value ? 1 : 0);
    return SG_OK;
  }
  if (s == "ntt.coset_scale_pass") {   // A-B: the coset shift of coeff_to_cosets as its own pass (1) or inside the first NTT pass (0, default)
    g_coset_scale_pass.store(value ? 1 : 0);
    return SG_OK;
  }
  if (s == "quotient.fused_numerator") {   // 1 (default): sg_quotient_numerator_cosets_dev may run its one-pass kernel; 0: always the separate kernels
    g_numerator_fused.store(value ? 1 : 0);
    return SG_OK;
  }
  if (s == "msm.acc_log") {   // profiling: record every msm_accumulate launch in issue order (sg_msm_launch_log); setting 1 clears the log
    msm_acc_log_enable(value != 0);
    return SG_OK;
  }
  if (s == "commit.combine_runners") {
    g_comb.max_runners.store(std::max(1, std::min(value, 4)));
    return SG_OK;
  }
  if (s == "lanes") {   // how many concurrent calls get a context of their own (1 .. 8); further callers wait for a lane
    if (value < 1 || value > kLanes) return fail(SG_ERR_INVALID, "sg_set_param: lanes in [1, 8]");
    g_lane_count.store(value);
    return SG_OK;
  }
  {
    LOCKED_CTX();   // makes sure a context exists to validate the name against
    int rc = apply_param(*g_ctx, s, value);
    if (rc != SG_OK) return rc;
    std::lock_guard<std::mutex> lk(g_sh.mu);
    g_sh.params.emplace_back(s, value);   // lanes created later replay it
  }
  // the lanes that exist already, one at a time and with none held (two threads setting parameters at once cannot
  // wait for each other's lane): each when it is idle
  for (auto& l : g_lanes) {
    std::lock_guard<std::mutex> lk(l.mu);
    if (l.ctx) (void)apply_param(*l.ctx, s, value);
  }
  return SG_OK;
}

int sg_get_param(const char* name, int* value) {
  if (!name || !value) return fail(SG_ERR_INVALID, "sg_get_param: bad argument");
  const std::string s(name);
  if (s == "commit.combine_wait_us") *value = g_comb.wait_us.load();
  else if (s == "commit.combine_target") *value = g_comb.target.load();
  else if (s == "commit.combine_runners") *value = g_comb.max_runners.load();
  else if (s == "msm.host_chunks") *value = g_host_chunks.load();
  else if (s == "msm.tiny_max") *value = g_msm_tiny_max.load();
  else if (s == "host.wait_sleep_us") *value = host_wait_sleep_us().load();
  else if (s == "lanes") *value = g_lane_count.load();
  else {
    // per-lane parameters: the value most recently set through sg_set_param (0 = never set: the built-in default applies)
    std::lock_guard<std::mutex> lk(g_sh.mu);
    *value = 0;
    bool known = false;
    for (const auto& kv : g_sh.params)
      if (kv.first == s) {
        *value = kv.second;
        known = true;
      }
    if (!known && s.rfind("msm.", 0) != 0 && s.rfind("ntt.", 0) != 0 && s != "side_prio")
      return fail(SG_ERR_INVALID, "sg_get_param: unknown parameter");
  }
  return SG_OK;
}

int sg_msm_launch_log(uint32_t* out_words, size_t cap_records, size_t* n_records) {
  if (!n_records || (cap_records && !out_words)) return fail(SG_ERR_INVALID, "sg_msm_launch_log: bad argument");
  static_assert(sizeof(AccLaunchRecord) == 8 * sizeof(uint32_t), "record layout is part of the ABI");
  *n_records = msm_acc_log_read(reinterpret_cast<AccLaunchRecord*>(out_words), cap_records);
  return SG_OK;
}

int sg_abi_version(void) { return SG_ABI_VERSION; }

int sg_time_ntt_dev(void* d_a, uint32_t log_n, int reps, float* ms_out) {
  if (!d_a || !ms_out || reps < 1 || log_n > 28) return fail(SG_ERR_INVALID, "sg_time_ntt_dev: bad argument");
  LOCKED_CTX();
  const DomainConsts* dc;
  TRY(get_consts(log_n, &dc));
  fp_words* a = static_cast<fp_words*>(d_a);
  hipStream_t s = g_ctx->stream;
  TRY(ntt_dev(a, (size_t)1 << log_n, a, log_n, dc->omega, nullptr, nullptr, nullptr, s));  // warm plan + caches
  hipEvent_t e0, e1;
  CHECK_HIP(hipEventCreate(&e0), "event");
  CHECK_HIP(hipEventCreate(&e1), "event");
  CHECK_HIP(hipEventRecord(e0, s), "event");
  for (int r = 0; r < reps; r++) TRY(ntt_dev(a, (size_t)1 << log_n, a, log_n, dc->omega, nullptr, nullptr, nullptr, s));
  CHECK_HIP(hipEventRecord(e1, s), "event");
  CHECK_HIP(host_wait_event(e1), "event");
  float ms = 0;
  CHECK_HIP(hipEventElapsedTime(&ms, e0, e1), "event");
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_out = ms / reps;
  return SG_OK;
}

}  // extern "C"
