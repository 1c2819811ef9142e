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
  // advice columns of MstInclusionCircuit for `n_users` users of a device-resident tree (level-major node arrays):
  // d_program = n_items x 5 words of items followed by n_absorbs x 3 words of absorbs (see witness.hip);
  // advice: n_users x 3 x rows field elements, zero-initialised by the caller
  hipError_t inclusion_witness(const uint32_t* d_program, uint32_t n_items, uint32_t n_absorbs, const fp_words* users,
                               const fp_words* hashes, const fp_words* balances, uint32_t depth, uint32_t nc,
                               const uint32_t* d_user_index, uint32_t n_users, fp_words* advice, size_t rows,
                               hipStream_t stream);

 private:
  void* table_ = nullptr;
};
}  // namespace sg
