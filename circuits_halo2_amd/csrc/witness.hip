// Witness side of the prover: Summa's Merkle sum tree over
// Poseidon(t = 2, rate 1, R_F = 8, R_P = 56, x^5) on BN254 Fr -- the GPU counterpart of
// zk_prover/src/merkle_sum_tree/{node.rs:16-84, utils/build_tree.rs:5-78}:
//   leaf   = H(username, balance_0 .. balance_{NC-1})
//   middle = H(bal_l0 + bal_r0, .., hash_l, hash_r),  balances = element-wise sums
// H = halo2_gadgets' Pow5 sponge with ConstantLength<L>: state = [0, L * 2^64]; every input is
// added to state[0] and followed by one permutation; the output is state[0].
// One thread per hash; 472-ish Fr products per permutation, so the kernels are VALU-bound.
#include "witness.h"

#include "poseidon_constants.inc"

namespace sg {

typedef Fr29 P;
struct PoseidonTable {  // 2^261-domain limbs, built once per context
  f29 rc[64][2];
  f29 mds[2][2];
};

__global__ void poseidon_table_kernel(const uint32_t* __restrict__ rc_words, const uint32_t* __restrict__ mds_words,
                                      PoseidonTable* out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 128) {
    uint32_t w[8];
    for (int k = 0; k < 8; k++) w[k] = rc_words[8 * i + k];
    out->rc[i >> 1][i & 1] = f29_words_to_r261<P>(w);
  } else if (i < 132) {
    uint32_t w[8];
    for (int k = 0; k < 8; k++) w[k] = mds_words[8 * (i - 128) + k];
    out->mds[(i - 128) >> 1][(i - 128) & 1] = f29_words_to_r261<P>(w);
  }
}

struct LdsTable {
  uint32_t rc[128][9];
  uint32_t mds[4][9];
};
__device__ __forceinline__ f29 lds_f29(const uint32_t (*a)[9], uint32_t i) {
  f29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.l[k] = a[i][k];
  return r;
}
__device__ __forceinline__ void stage_table(const PoseidonTable* t, LdsTable* s) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(t);
  uint32_t* dst = reinterpret_cast<uint32_t*>(s);
  for (uint32_t i = threadIdx.x; i < 132 * 9; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
}
__device__ __forceinline__ f29 pow5(const f29& x) {   // x bound <= 6
  f29 x2 = f29_sqr<P>(x);                             // 36
  f29 x4 = f29_sqr<P>(x2);                            // 4
  return f29_mul<P>(x4, x);                           // 12
}
// state <- MDS * state with one reduction per row (f29_mul2): (1*2 + 1*2) / 170 + 1 < 2
__device__ __forceinline__ void mix(f29& s0, f29& s1, const LdsTable* t) {
  f29 a = f29_mul2<P>(lds_f29(t->mds, 0), s0, lds_f29(t->mds, 1), s1);
  f29 b = f29_mul2<P>(lds_f29(t->mds, 2), s0, lds_f29(t->mds, 3), s1);
  s0 = a;
  s1 = b;
}
__device__ void poseidon_permute(f29& s0, f29& s1, const LdsTable* t) {
  uint32_t r = 0;
  for (int k = 0; k < 4; k++, r++) {
    s0 = pow5(f29_add(s0, lds_f29(t->rc, 2 * r)));      // (<4 + <2)^5
    s1 = pow5(f29_add(s1, lds_f29(t->rc, 2 * r + 1)));
    mix(s0, s1, t);
  }
  for (int k = 0; k < 56; k++, r++) {
    s0 = pow5(f29_add(s0, lds_f29(t->rc, 2 * r)));
    s1 = f29_add(s1, lds_f29(t->rc, 2 * r + 1));        // < 4, enters mix with bound product 1*4
    mix(s0, s1, t);
  }
  for (int k = 0; k < 4; k++, r++) {
    s0 = pow5(f29_add(s0, lds_f29(t->rc, 2 * r)));
    s1 = pow5(f29_add(s1, lds_f29(t->rc, 2 * r + 1)));
    mix(s0, s1, t);
  }
}
__device__ __forceinline__ f29 load_hat_w(const fp_words* p) {
  uint32_t w[8];
  fp_words_load(p, w);
  return f29_words_to_r261<P>(w);
}
__device__ __forceinline__ void store_hat_w(fp_words* p, const f29& x_hat) {
  uint32_t w[8];
  f29_to_words(f29_reduce_with<P>(x_hat, P::r256), w);
  fp_words_store(p, w);
}
// capacity element L * 2^64 in the 2^261 domain: limbs of the integer, times 2^522 * 2^-261
__device__ __forceinline__ f29 capacity_element(uint32_t L) {
  f29 v = f29_zero();
  // L * 2^64 = L << (2*29 + 6): limb 2 gets the low bits, limb 3 the rest
  uint64_t x = (uint64_t)L << 6;
  v.l[2] = (uint32_t)(x & M29);
  v.l[3] = (uint32_t)(x >> 29);
  // integer -> 2^261 domain: v * 2^517 * 2^-261 = v~ (2^256 form), then * 2^266 * 2^-261
  return f29_mul<P>(f29_mul<P>(v, f29_const<P>(P::r517)), f29_const<P>(P::r266));
}

__global__ void __launch_bounds__(128) mst_leaves_kernel(const fp_words* __restrict__ users,
                                                         const fp_words* __restrict__ balances, uint32_t n,
                                                         uint32_t nc, const PoseidonTable* __restrict__ table,
                                                         fp_words* __restrict__ hashes) {
  __shared__ LdsTable tab;
  stage_table(table, &tab);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f29 s0 = f29_zero(), s1 = capacity_element(nc + 1);
  s0 = f29_add(s0, load_hat_w(users + i));
  poseidon_permute(s0, s1, &tab);
  for (uint32_t c = 0; c < nc; c++) {
    s0 = f29_add(s0, load_hat_w(balances + (size_t)i * nc + c));
    poseidon_permute(s0, s1, &tab);
  }
  store_hat_w(hashes + i, s0);
}
// parents p = 0..m-1 from children 2p, 2p+1
__global__ void __launch_bounds__(128) mst_level_kernel(const fp_words* __restrict__ child_hash,
                                                        const fp_words* __restrict__ child_bal, uint32_t m,
                                                        uint32_t nc, const PoseidonTable* __restrict__ table,
                                                        fp_words* __restrict__ hashes, fp_words* __restrict__ bal) {
  __shared__ LdsTable tab;
  stage_table(table, &tab);
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  f29 s0 = f29_zero(), s1 = capacity_element(nc + 2);
  for (uint32_t c = 0; c < nc; c++) {
    f29 b = f29_add(load_hat_w(child_bal + (size_t)(2 * p) * nc + c), load_hat_w(child_bal + (size_t)(2 * p + 1) * nc + c));
    store_hat_w(bal + (size_t)p * nc + c, b);          // canonical sum
    s0 = f29_add(s0, b);                                // < 2 + 4
    poseidon_permute(s0, s1, &tab);
  }
  s0 = f29_add(s0, load_hat_w(child_hash + 2 * p));
  poseidon_permute(s0, s1, &tab);
  s0 = f29_add(s0, load_hat_w(child_hash + 2 * p + 1));
  poseidon_permute(s0, s1, &tab);
  store_hat_w(hashes + p, s0);
}

// ------------------------------------------------------------------ host side
hipError_t WitnessEngine::init(hipStream_t stream) {
  if (table_) return hipSuccess;
  uint32_t *d_rc = nullptr, *d_mds = nullptr;
  hipError_t e = hipMalloc(&table_, sizeof(PoseidonTable));
  if (e == hipSuccess) e = hipMalloc(&d_rc, sizeof(POSEIDON_RC));
  if (e == hipSuccess) e = hipMalloc(&d_mds, sizeof(POSEIDON_MDS));
  if (e == hipSuccess) e = hipMemcpyAsync(d_rc, POSEIDON_RC, sizeof(POSEIDON_RC), hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_mds, POSEIDON_MDS, sizeof(POSEIDON_MDS), hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) {
    poseidon_table_kernel<<<2, 128, 0, stream>>>(d_rc, d_mds, static_cast<PoseidonTable*>(table_));
    e = hipStreamSynchronize(stream);
  }
  if (d_rc) (void)hipFree(d_rc);
  if (d_mds) (void)hipFree(d_mds);
  return e;
}
void WitnessEngine::release() {
  if (table_) (void)hipFree(table_);
  table_ = nullptr;
}
hipError_t WitnessEngine::leaves(const fp_words* users, const fp_words* balances, size_t n, uint32_t nc,
                                 fp_words* hashes, hipStream_t stream) {
  if (!n) return hipSuccess;
  mst_leaves_kernel<<<(unsigned)((n + 127) / 128), 128, 0, stream>>>(users, balances, (uint32_t)n, nc,
                                                                    static_cast<const PoseidonTable*>(table_), hashes);
  return hipGetLastError();
}
hipError_t WitnessEngine::level(const fp_words* child_hash, const fp_words* child_bal, size_t m, uint32_t nc,
                                fp_words* hashes, fp_words* bal, hipStream_t stream) {
  if (!m) return hipSuccess;
  mst_level_kernel<<<(unsigned)((m + 127) / 128), 128, 0, stream>>>(child_hash, child_bal, (uint32_t)m, nc,
                                                                   static_cast<const PoseidonTable*>(table_), hashes, bal);
  return hipGetLastError();
}

}  // namespace sg
