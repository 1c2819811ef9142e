// Host-side interface of the MSM engine (see msm.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>

#include "bn254_curve29.cuh"
#include "devmem.h"

namespace sg {

#ifndef SG_WORDS8
#define SG_WORDS8
struct words8 {  // one field element as 8 LE u32 words (Montgomery-2^256), host side
  uint32_t l[8];
};
#endif

static constexpr uint32_t MSM_LOG_FUSE_ENTRIES_GENERIC = 25, MSM_LOG_FUSE_ENTRIES_FIXED = 27;
static constexpr uint32_t MSM_TINY_MAX = 64;   // MSMs of at most this many points run as ONE launch (MsmEngine::run_tiny)
struct MsmConfig {
  uint32_t window_bits = 0;    // 0: choose from n (log2 n - 2 single / - 4 fused, clamped to [4, 16])
  uint32_t log_seg = 0;        // L = 2^log_seg entries per accumulation task; 0: choose from n
  // fused batches hold at most 2^x (window, scalar) entries.  Fixed-base jobs (commitments): 27 = 64 polynomials of 2^17 rows over a
  // 16-window table in ONE job (round 5: with 25 a fused commitment job of a proof batch was cut into jobs of 16 polynomials, each
  // with a sort front end, reduction and host tail of its own: 269-272 -> 287-299 proofs/s at 64 proofs in flight,
  // profiles/r05_sweeps/batch_knobs.txt).  Generic jobs (sg_msm_g1_batch*) stay at 25: their groups alternate between two engines,
  // and eight MSMs of 2^20 run 6 % faster as four groups of two than as two of four (bench.py "batched": 1.45 against 1.54 ms per MSM)
  uint32_t log_fuse_entries = MSM_LOG_FUSE_ENTRIES_GENERIC;
  uint32_t log_fuse_entries_fixed = MSM_LOG_FUSE_ENTRIES_FIXED;
  uint32_t red_threads = 256;      // workgroup size of the level-0 bucket reduction (64, 128 or 256)
  uint32_t log_red_chunk = 0;  // G = 2^x buckets per thread in the bucket reduction; 0: auto
  uint32_t two_pass = 1;            // two-pass (coarse bin, in-LDS fine) sort: 0 never, 1 auto (>= 2^19 entries), 2 always
  uint32_t log_scatter_rounds = 0;  // the counting sort's scatter runs in 2^x bucket-range rounds
  uint32_t acc_threads = 0;    // workgroup size of msm_accumulate (0: 128)
  uint32_t acc_waves = 0;      // waves per SIMD of the persistent msm_accumulate launch: 0 = 3 (a full register file); >= 8: grid = tasks, one ticket per wave
  uint32_t acc_waves_fixed = 0; // ... of fixed-base jobs (the commitments of a proof, which run beside that proof's transforms on other streams): 0 = 2
  uint32_t merge_quad_tasks = 0xffffffffu;  // merge rounds with more tasks than this use one lane per addition even when `quad` holds
  uint32_t red2d_max_sets = 6; // ... host-weights variant up to this many bucket sets (measured: tools/sweep_red2d.sh; 6 since the partial sums are folded first)
  uint32_t red2d = 1;          // 2-D (row / column / bit) bucket reduction: 0 never, 1 jobs of <= 4 bucket sets, 2 always
  uint32_t red2d_fold = 8;     // ... whose line sums add up to this many partial sums per bucket themselves (no merge round below that)
  uint32_t red2d_prefold = 1;  // ... after a pass that adds every bucket's partial sums once (msm_fold_buckets); 0: the line sums add them on the way (twice)
  uint32_t prefold_quad_buckets = 1u << 15;  // ... with a quad per bucket up to this many buckets in the job, one lane per bucket beyond
  uint32_t acc_chain = 1;      // accumulations of different jobs run one after the other (each waits for the previous launch's event)
  uint32_t red_lean = 1;       // level-0 bucket reduction within 168 registers (fits beside a polite accumulation): 0 never, 1 when other jobs are in flight, 2 always
  uint32_t fused_frontend = 1; // two-pass sort: scans and task histogram inside the sort's own kernels (msm_fine_sort_fused): 0 never (the launches of rounds 1-4), 1 when no other job is in flight, 2 always
  uint32_t acc_trace = 0;      // debug: msm_accumulate records when each wave starts and leaves; finish() prints the percentiles to stderr
  uint32_t quad = 1;           // quad-cooperative point additions in merge / reduction: 0 never, 1 auto, 2 always
};

struct MsmTimings {
  float digits_ms = 0, sort_ms = 0, accumulate_ms = 0, reduce_ms = 0, total_ms = 0;
  uint32_t window_bits = 0, windows = 0, tasks = 0, max_bucket = 0, accumulate_threads = 0;
  float order_ms = 0;   // between the sort and the accumulation: task ordering, and the wait behind other jobs' accumulations
};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t need) {
    if (p && need <= cap) return hipSuccess;
    const size_t want = need + need / 4 + 64;
    T* q = nullptr;
    hipError_t e = hipMalloc(&q, want * sizeof(T));
    if (e != hipSuccess) {   // out of memory: give the retired blocks back and try once more
      retired_device_memory_collect();
      e = hipMalloc(&q, want * sizeof(T));
      if (e != hipSuccess) return e;
    }
    retire_device_memory(p);   // kernels enqueued earlier may still be reading the old block
    p = q;
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct WindowPlan {  // per-window digit widths (see msm_digits)
  uint32_t W;
  uint8_t width[64];
};

WindowPlan make_window_plan(uint32_t c);

// Fixed-base mode (resident SRS): row w of `table` holds 2^(bit offset of window w) * P_i, so the W
// digits of a scalar are W independent (digit, point) pairs of ONE bucket set: W - 1 of the W
// bucket reductions disappear and the window can be as wide as the sort allows.
struct FixedTable {
  g1_affine_mem* table = nullptr;  // W x n affine points, row major
  size_t n = 0;
  uint32_t c = 0;
  WindowPlan wp{};
};
uint32_t fixed_window_bits_for(size_t n);
hipError_t build_window_table(const g1_affine_mem* d_bases, size_t n, uint32_t c, FixedTable* out, hipStream_t stream);

static constexpr size_t MAX_FUSED = 64;
struct BatchPtrs {  // inputs of a fused batch (kernel argument)
  const fp_words* scalars[MAX_FUSED];
  const g1_affine_mem* bases[MAX_FUSED];
  uint64_t diff_mask;  // bit m: MSM m is taken in difference form -- its digits are those of s[i] - s[i+1] (s[n] = 0) and
                       // its bases the inclusive prefix sums of the basis (see g1_prefix_sums)
};

class MsmEngine {
 public:
  ~MsmEngine();
  hipError_t init();
  void release();
  MsmConfig& config() { return cfg_; }
  void set_tail_stream(hipStream_t s) { tail_stream_ = s; }
  uint32_t window_bits_for(size_t n, bool fused = false) const;
  // d_scalars: n x 32 B Montgomery Fr, d_bases: n x 64 B affine; result: 64 B affine on the host
  hipError_t run(const fp_words* d_scalars, const g1_affine_mem* d_bases, size_t n, hipStream_t stream, uint8_t out_affine[64],
                 MsmTimings* tm);
  // n <= MSM_TINY_MAX points from HOST memory (the verifier's 37): one launch, no staging copy, result on return (msm.hip)
  hipError_t run_tiny(const uint8_t* h_scalars, const uint8_t* h_bases, size_t n, hipStream_t stream, uint8_t out_affine[64]);
  // the same in three phases, so that two engines on two streams can overlap one MSM's
  // latency-bound tail with the next MSM's sort/accumulate (sg_msm_g1_batch):
  //   enqueue_front: digits .. counting sort (+ async copy of the task counters)
  //   enqueue_back:  waits for the counters, enqueues accumulate .. reduction + result copy
  //   finish:        waits for the result copy, host tail (Horner + normalisation)
  hipError_t enqueue_front(const fp_words* d_scalars, const g1_affine_mem* d_bases, size_t n, hipStream_t stream,
                           uint8_t* out_affine, MsmTimings* tm);
  // M same-length MSMs fused into one job (out_affine: M x 64 bytes); M <= max_fused(n)
  hipError_t enqueue_front_fused(const fp_words* const* d_scalars, const g1_affine_mem* const* d_bases, size_t M,
                                 size_t n, hipStream_t stream, uint8_t* out_affine, MsmTimings* tm);
  size_t max_fused(size_t n) const;
  // the same over a precomputed window table (n <= tab.n): all M MSMs use the table's bases
  hipError_t enqueue_front_fixed(const fp_words* const* d_scalars, const FixedTable& tab, size_t M, size_t n,
                                 hipStream_t stream, uint8_t* out_affine, MsmTimings* tm,
                                 const g1_affine_mem* const* tables = nullptr, uint64_t diff_mask = 0);
  size_t max_fused_fixed(const FixedTable& tab, size_t n) const;
  hipError_t enqueue_back();
  hipError_t finish();

 private:
  hipError_t enqueue_front_fused_impl(const fp_words* const* d_scalars, const g1_affine_mem* const* d_bases, size_t M, size_t n,
                                      hipStream_t stream, uint8_t* out_affine, MsmTimings* tm);
  hipError_t enqueue_back_impl();
  hipError_t finish_impl();
  void mark_in_flight(bool on);
  bool others_in_flight() const;   // another engine of this process has a job between its first kernel and its host tail
  bool counted_ = false;
  hipError_t chained_accumulate(hipStream_t stream, hipEvent_t after_wait, const std::function<void()>& launch);
  hipEvent_t ev_chain_[2] = {nullptr, nullptr};
  int chain_slot_ = 0;
  uint32_t cus_ = 256;
  struct Job {
    BatchPtrs bp{};
    uint32_t M = 1;
    size_t n = 0, entries = 0;
    hipStream_t stream = nullptr;
    uint8_t* out = nullptr;
    MsmTimings* tm = nullptr;
    bool trivial = false, all_zero = false, fixed = false, fe = false;
    uint32_t fe_parity = 0, NBc = 0;   // fused front end: which replica set of fe_ this job uses; coarse bins of the job
    uint32_t red2d = 0;  // 0: scan-based reduction, 1: 2-D with host weights, 2: 2-D with device weights
    uint32_t n_tab = 0;
    uint32_t c = 0, nbw = 0, NB = 0, log_L = 0, log_G = 0, log_N = 0, blocks = 0, ntasks = 0, max_cnt = 0, acc_threads = 0;
    WindowPlan wp{};
    hipEvent_t ev[6];
  };
  Job job_;
  const FixedTable* fixed_ = nullptr;  // set only while enqueue_front_fixed runs
  uint64_t diff_mask_ = 0;             // likewise
  uint32_t fe_parity_ = 0;             // fused front end: replica set of the most recent job (alternates)
  hipEvent_t ev_meta_ = nullptr, ev_done_ = nullptr, ev_acc_ = nullptr, ev_tiny_ = nullptr;
  hipStream_t tail_stream_ = nullptr;  // optional high-priority stream for reduce + export
  MsmConfig cfg_;
  DevBuf<int16_t> dig_;
  DevBuf<uint2> order_;
  DevBuf<uint32_t> thist_;
  DevBuf<uint32_t> sorted_, counts_, off_, ntask_[2], toff_[2], hist_, bsum_, meta_;
  DevBuf<xyzz29_mem> partial_[2], red_a_[2], red_s_[2], red_r_[2];
  DevBuf<uint32_t> win_words_, fe_, tbase_;
  DevBuf<uint64_t> trace_;
  DevBuf<uint32_t> part_entry_, ccnt_, coff_;  // two-pass sort: partitioned entries, coarse-bin counts / offsets
  DevBuf<uint16_t> part_fine_;
  uint32_t* h_meta_ = nullptr;   // page-locked, written by the scan kernels through d_hmeta_ (its device address)
  uint32_t* d_hmeta_ = nullptr;
  size_t h_win_cap_ = 0;
  uint32_t* h_win_ = nullptr;    // W x 3 x 32 words: canonical XYZZ of (A, S, T) per window; written by the export kernels through d_hwin_
  uint32_t* d_hwin_ = nullptr;
  uint8_t* h_tiny_ = nullptr;    // run_tiny: scalars, points (read by the kernel) and the window sums (written by it), mapped
  uint8_t* d_tiny_ = nullptr;
};

// one msm_accumulate launch as the engine issued it (parameter "msm.acc_log"; sg_msm_launch_log)
struct AccLaunchRecord {
  uint64_t entries;       // (digit, point) pairs the launch accumulates
  uint32_t n, M;          // the job: M polynomials / MSMs of n scalars each
  uint32_t threads;       // grid of the launch
  uint32_t fixed;         // 1: fixed-base job (a commitment), 0: generic
  uint32_t jobs_in_flight;  // jobs of the process in flight when it was issued (this one included)
  uint32_t task_len;      // L: entries per accumulation task
};
void msm_acc_log_enable(bool on);
size_t msm_acc_log_read(AccLaunchRecord* out, size_t cap);   // returns the number of records held
// +1 / -1 on the count of jobs in flight (MsmEngine::others_in_flight), for callers that run several jobs side by side
void msm_hold_in_flight(bool on);
// FFT over G1 (N5): out = DFT_omega(in) [* scale], natural order; d_work: 2^log_n xyzz29_mem
hipError_t g1_fft(const g1_affine_mem* d_in, g1_affine_mem* d_out, uint32_t log_n, const words8& omega,
                  const words8* scale, xyzz29_mem* d_work, hipStream_t stream);
hipError_t fixed_base_mul(const fp_words* d_scalars, size_t n, g1_affine_mem* d_out, hipStream_t stream);
// out[i] = in[0] + ... + in[i], affine (the basis of difference-form commitments: sum_i s_i P_i = sum_i (s_i - s_{i+1}) Q_i with
// Q the inclusive prefix sums and s_n = 0 -- a column that is constant over long runs becomes a sparse MSM).  Set-up
// time only; allocates and frees its own work space.  d_out may not alias d_in.
hipError_t g1_prefix_sums(const g1_affine_mem* d_in, size_t n, g1_affine_mem* d_out, hipStream_t stream);
// *d_bad = number of points that are neither on y^2 = x^3 + 3 nor the identity
hipError_t g1_on_curve(const g1_affine_mem* d_points, size_t n, uint32_t* d_bad, hipStream_t stream);

}  // namespace sg
