// Host-side interface of the witness kernels (see witness.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "msm.h"

namespace sg {
class WitnessEngine {
 public:
  ~WitnessEngine() { release(); }
  hipError_t init(hipStream_t stream);
  void release();
  // hashes[i] = Poseidon(users[i], balances[i][0..nc))
  hipError_t leaves(const fp_words* users, const fp_words* balances, size_t n, uint32_t nc, fp_words* hashes,
                    hipStream_t stream);
  // parent p of children 2p, 2p+1: balances summed, hash = Poseidon(sums.., hash_l, hash_r)
  hipError_t level(const fp_words* child_hash, const fp_words* child_bal, size_t m, uint32_t nc, fp_words* hashes,
                   fp_words* bal, hipStream_t stream);

 private:
  void* table_ = nullptr;
};
}  // namespace sg
