// Host-side interface of the NTT engine (see ntt.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <deque>
#include <vector>

#include "bn254_curve29.cuh"
#include "devmem.h"

namespace sg {

struct NttConfig {
  // measured on MI355X (profiles/r01_sweeps/ntt_sweep*.txt): small tiles (several workgroups per CU
  // hide the barrier and load latency) beat fewer, longer passes: the kernel is product-bound
  uint32_t max_single_log = 11;  // largest transform done in one LDS-resident pass
  uint32_t max_multi_log = 9;    // largest per-pass DFT length in multi-pass plans (2^17 = 2^9 x 2^8: two passes)
  uint32_t tile_log = 9;         // log2(elements per workgroup tile)  (2^9 * 36 B = 18 KiB LDS)
  uint32_t threads = 256;        // one butterfly per thread per stage at tile_log = 9
  // throughput shape, for launches that fill the chip anyway (batches of >= batch_min vectors, transforms of >= 2^big_log):
  // re-swept in round 3 with the round-2 closings in place (profiles/r03_sweeps/ntt_plans.txt) -- a lone 2^17 transform is
  // three or two launches at their latency floor and wants many small workgroups; 16 of them, or 2^22 points, want
  // wider tiles (two columns per 2^9-point DFT, coalesced 64-byte runs) on 512 threads
  uint32_t big_tile_log = 10;
  uint32_t big_threads = 512;
  uint32_t batch_min = 4;
  uint32_t big_log = 20;
  uint32_t radix4 = 1;           // two DIT stages per sweep over the LDS tile ("ntt.radix4"; same words): 0 never, 1 the throughput shapes (batches of >= batch_min vectors, transforms of >= 2^big_log points), 2 always
};

#ifndef SG_WORDS8
#define SG_WORDS8
struct words8 {  // one field element as 8 LE u32 words (Montgomery-2^256), host side
  uint32_t l[8];
};
#endif

static constexpr uint32_t NTT_BATCH_MAX = 32;   // vectors per batched launch (the 25 coset blocks of a proof phase: one launch per pass)
struct NttPlan {
  uint32_t log_n;
  words8 omega;
  words8 scale;
  bool has_scale;
  int npass;
  uint32_t l[3];      // log2 of the per-pass DFT lengths n1, n2, n3 (n = n1*n2*n3)
  fp_words* tw_local[3];  // powers of omega_{n_i} (2^261 domain)
  fp_words* tw_pass[3];   // inter-pass twiddles in output order (2^261 domain)
};

class NttEngine {
 public:
  ~NttEngine();
  hipError_t init();
  void clear();
  NttConfig& config() { return cfg_; }
  // out = DFT_omega(pre3 .* zero-extend(in)) .* post3 (* scale); `in == out` allowed (then
  // `scratch` of 2^log_n elements is used for multi-pass plans).
  hipError_t transform(const fp_words* in, size_t in_len, fp_words* out, fp_words* scratch, uint32_t log_n,
                       const words8& omega, const words8* scale, const words8* pre3, const words8* post3,
                       hipStream_t stream);
  // `count` <= NTT_BATCH_MAX in-place transforms of one size, one launch per pass (see ntt.hip)
  hipError_t transform_batch(fp_words* const* a, uint32_t count, fp_words* scratch, uint32_t log_n, const words8& omega,
                             const words8* scale, hipStream_t stream, const fp_words* const* src = nullptr,
                             size_t src_len = 0, const words8* pre3 = nullptr, const fp_words* const* pre_tab = nullptr);
  // (pre_tab[i]: optional table of 2^log_n 2^261-domain words; input element j of vector i is multiplied by pre_tab[i][j] on
  //  the way into the first pass -- the coset shift c^j of coeff_to_cosets, which used to be a pass over HBM of its own)
  // cached table of omega_r^t, t < 2^(log_r - 1), as 2^261-domain words
  hipError_t local_twiddles(const words8& omega_r, uint32_t log_r, hipStream_t stream, fp_words** out);

 private:
  struct LocalTw {
    uint32_t log_r;
    words8 omega_r;
    fp_words* tw;
  };
  hipError_t get_plan(uint32_t log_n, const words8& omega, const words8* scale, hipStream_t stream, const NttPlan** out);
  NttConfig cfg_;
  std::deque<NttPlan> plans_;
  std::vector<LocalTw> local_tw_;
};

__global__ void pow_single(fp_words* out, words8 w, uint64_t e);
hipError_t ntt_scale(fp_words* a, const words8& s, size_t n, hipStream_t stream);
hipError_t ntt_scale_periodic(fp_words* a, const fp_words* tab, uint32_t period, size_t n, hipStream_t stream);
hipError_t fr_montgomery(const fp_words* in, fp_words* out, size_t n, int to_mont, hipStream_t stream);

}  // namespace sg
