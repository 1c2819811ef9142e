// Host-side interface of the NTT engine (see ntt.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <deque>
#include <vector>

#include "bn254_field.cuh"

namespace sg {

struct NttConfig {
  uint32_t max_single_log = 11;  // largest transform done in one LDS-resident pass
  uint32_t max_multi_log = 10;   // largest per-pass DFT length in multi-pass plans
  uint32_t tile_log = 11;        // log2(elements per workgroup tile)  (2^11 * 32 B = 64 KiB LDS)
  uint32_t threads = 1024;       // one butterfly per thread per stage at tile_log = 11
};

struct NttPlan {
  uint32_t log_n;
  fp_t omega;
  fp_t scale;
  bool has_scale;
  int npass;
  uint32_t l[3];      // log2 of the per-pass DFT lengths n1, n2, n3 (n = n1*n2*n3)
  fp_t* tw_local[3];  // powers of omega_{n_i}
  fp_t* tw_pass[3];   // inter-pass twiddles in output order
};

class NttEngine {
 public:
  ~NttEngine();
  hipError_t init();
  void clear();
  NttConfig& config() { return cfg_; }
  // out = DFT_omega(pre3 .* zero-extend(in)) .* post3 (* scale); `in == out` allowed (then
  // `scratch` of 2^log_n elements is used for multi-pass plans).
  hipError_t transform(const fp_t* in, size_t in_len, fp_t* out, fp_t* scratch, uint32_t log_n, const fp_t& omega,
                       const fp_t* scale, const fp_t* pre3, const fp_t* post3, hipStream_t stream);

 private:
  struct LocalTw {
    uint32_t log_r;
    fp_t omega_r;
    fp_t* tw;
  };
  hipError_t get_plan(uint32_t log_n, const fp_t& omega, const fp_t* scale, hipStream_t stream, const NttPlan** out);
  hipError_t local_twiddles(const fp_t& omega_r, uint32_t log_r, hipStream_t stream, fp_t** out);
  NttConfig cfg_;
  std::deque<NttPlan> plans_;
  std::vector<LocalTw> local_tw_;
};

__global__ void pow_single(fp_t* out, fp_t w, uint64_t e);
hipError_t ntt_scale(fp_t* a, const fp_t& s, size_t n, hipStream_t stream);
hipError_t ntt_scale_periodic(fp_t* a, const fp_t* tab, uint32_t period, size_t n, hipStream_t stream);
hipError_t fr_montgomery(const fp_t* in, fp_t* out, size_t n, int to_mont, hipStream_t stream);

}  // namespace sg
