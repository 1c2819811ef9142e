// Witness side of the prover: Summa's Merkle sum tree over
// Poseidon(t = 2, rate 1, R_F = 8, R_P = 56, x^5) on BN254 Fr -- the GPU counterpart of
// zk_prover/src/merkle_sum_tree/{node.rs:16-84, utils/build_tree.rs:5-78}:
//   leaf   = H(username, balance_0 .. balance_{NC-1})
//   middle = H(bal_l0 + bal_r0, .., hash_l, hash_r),  balances = element-wise sums
// H = halo2_gadgets' Pow5 sponge with ConstantLength<L>: state = [0, L * 2^64]; every input is
// added to state[0] and followed by one permutation; the output is state[0].
// One thread per hash; 472-ish Fr products per permutation, so the kernels are VALU-bound.
#include "witness.h"
#include "host_wait.h"
#include "side_prio.cuh"

#include "poseidon_constants.inc"

namespace sg {
SG_DEFINE_SIDE_PRIO_SETTER(witness_set_side_prio)

typedef Fr29 P;
struct PoseidonTable {  // 2^261-domain limbs, built once per context
  f29 rc[64][2];
  f29 mds[2][2];
};

__global__ void poseidon_table_kernel(const uint32_t* __restrict__ rc_words, const uint32_t* __restrict__ mds_words,
                                      PoseidonTable* out) {
  side_kernel_prio();
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
__device__ __forceinline__ void poseidon_permute(f29& s0, f29& s1, const LdsTable* t) {
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
  side_kernel_prio();
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
  side_kernel_prio();
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

// ------------------------------------------------------------------ circuit witness (advice columns) on the device
// `MstInclusionCircuit::synthesize` [REF zk_prover/src/circuits/merkle_sum_tree.rs:228-520] for a user of a device-resident
// Merkle sum tree: the three advice columns of the inclusion circuit in the reference's own floor plan.  The floor plan
// (which region starts at which row) depends only on <LEVELS, N_CURRENCIES, N_BYTES>; the host derives it once per
// proving key by replaying the floor planner (mst_inclusion.witness_program) and hands it over as a flat "program":
//   items     one per thread: a single cell copy, a range-check running sum, or a whole Poseidon hash region chain
//   absorbs   per absorbed word of a hash: rows of its "add input" and "permute state" regions, the word's source
// Every value a cell can hold is a node of the tree (a hash, a balance, a username) or a path bit, so a source is a
// "symbol" (kind, level, mode, lane) resolved against the tree arrays with the user's index -- no value depends on
// another cell, all items of all users run in one launch.  Sponge traces are recomputed here round by round (they
// are what the circuit constrains); the digests they end in are the tree's own nodes.
__device__ const uint32_t FR_ONE_M[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                         0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};   // 2^256 mod r: Montgomery form of 1
struct WitnessItem { uint32_t kind, col, row, sym, extra; };       // kind 0 cell, 1 range (extra = bytes), 2 hash (extra = first absorb | count << 20 | chip << 28)
struct WitnessAbsorb { uint32_t add_row, permute_row, sym; };
struct WitnessTree {
  const fp_words* users;      // 2^depth username field elements
  const fp_words* hashes;     // level-major node hashes
  const fp_words* balances;   // level-major node balances, nc per node
  uint32_t depth, nc;
  uint32_t level_offset[33];
};
// symbol: bits 0-3 kind (0 username, 1 node hash, 2 node balance, 3 path bit), 4-9 level, 10-12 mode, 13-19 lane
// modes of a node: 0 the path node (idx >> level), 1 its sibling, 2 child `lane & 1` of the sibling (one level down),
// 3 the left / right child (`lane & 1`) of the path's parent, at this level; for balances `lane` is the currency
__device__ __forceinline__ uint32_t sym_node(uint32_t sym, uint32_t idx, uint32_t* level_out) {
  const uint32_t level = (sym >> 4) & 63, mode = (sym >> 10) & 7, lane = (sym >> 13) & 127;
  const uint32_t at = idx >> level;
  uint32_t node = at, lv = level;
  if (mode == 1) node = at ^ 1;
  else if (mode == 2) { node = 2 * (at ^ 1) + (lane & 1); lv = level - 1; }
  else if (mode == 3) node = (at & ~1u) + (lane & 1);
  *level_out = lv;
  return node;
}
__device__ __forceinline__ void sym_words(const WitnessTree& t, uint32_t sym, uint32_t idx, uint32_t w[8]) {
  const uint32_t kind = sym & 15;
  if (kind == 0) {
    fp_words_load(t.users + (idx ^ ((sym >> 10) & 1)), w);
  } else if (kind == 1) {
    uint32_t lv;
    const uint32_t node = sym_node(sym, idx, &lv);
    fp_words_load(t.hashes + t.level_offset[lv] + node, w);
  } else if (kind == 2) {
    uint32_t lv;
    const uint32_t node = sym_node(sym & ~(127u << 13), idx, &lv);
    fp_words_load(t.balances + (size_t)(t.level_offset[lv] + node) * t.nc + ((sym >> 13) & 127), w);
  } else {
    const uint32_t bit = (idx >> ((sym >> 4) & 63)) & 1;
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = bit ? FR_ONE_M[k] : 0u;
  }
}
struct AdviceOut { fp_words* col[3]; };
__device__ __forceinline__ void put_state(const AdviceOut& o, uint32_t row, const f29& s0, const f29& s1) {
  store_hat_w(o.col[0] + row, s0);
  store_hat_w(o.col[1] + row, s1);
}
__global__ void __launch_bounds__(64) mst_inclusion_witness_kernel(const WitnessItem* __restrict__ items, uint32_t n_items,
                                                                  const WitnessAbsorb* __restrict__ absorbs, WitnessTree tree,
                                                                  const uint32_t* __restrict__ user_index, uint32_t n_users,
                                                                  const PoseidonTable* __restrict__ table, fp_words* advice,
                                                                  size_t rows, size_t user_stride) {
  side_kernel_prio();
  __shared__ LdsTable tab;
  stage_table(table, &tab);
  // hash items first in the program: whole waves of sponges, then the cheap cell copies
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, u = blockIdx.y;
  if (i >= n_items || u >= n_users) return;
  const uint32_t idx = user_index[u];
  AdviceOut out;
  for (int c = 0; c < 3; c++) out.col[c] = advice + (size_t)u * user_stride + (size_t)c * rows;
  const WitnessItem it = items[i];
  uint32_t w[8];
  if (it.kind == 0) {
    sym_words(tree, it.sym, idx, w);
    fp_words_store(out.col[it.col] + it.row, w);
  } else if (it.kind == 1) {
    // running sum of the byte decomposition: row i holds value >> 8 i (the value is a balance < 2^(8 bytes))
    sym_words(tree, it.sym, idx, w);
    f29 one_int = f29_zero();
    one_int.l[0] = 1;
    uint32_t v[9];
    f29_to_words(f29_cond_sub_p<P>(f29_mul<P>(f29_words_to_r261<P>(w), one_int)), v);   // the canonical integer
    v[8] = 0;
    for (uint32_t b = 0; b < it.extra; b++) {
      // integer -> memory (2^256) form: v * 2^517 * 2^-261, canonical
      uint32_t o[8];
      f29_to_words(f29_reduce_with<P>(f29_from_words<0>(v), P::r517), o);
      fp_words_store(out.col[0] + it.row + b, o);
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = (v[k] >> 8) | (v[k + 1] << 24);
    }
  } else {
    const uint32_t first = it.extra & 0xfffff, count = (it.extra >> 20) & 255;
    f29 s0 = f29_zero(), s1 = capacity_element(count);
    put_state(out, it.row, s0, s1);                       // "initial state" region
    for (uint32_t j = 0; j < count; j++) {
      const WitnessAbsorb ab = absorbs[first + j];
      put_state(out, ab.add_row, s0, s1);                 // "add input": state, the word, state + word
      sym_words(tree, ab.sym, idx, w);
      fp_words_store(out.col[0] + ab.add_row + 1, w);
      s0 = f29_add(s0, f29_words_to_r261<P>(w));
      put_state(out, ab.add_row + 2, s0, s1);
      uint32_t row = ab.permute_row, r = 0;               // "permute state": 37 rows
      for (int k = 0; k < 4; k++, r++, row++) {
        put_state(out, row, s0, s1);
        s0 = pow5(f29_add(s0, lds_f29(tab.rc, 2 * r)));
        s1 = pow5(f29_add(s1, lds_f29(tab.rc, 2 * r + 1)));
        mix(s0, s1, &tab);
      }
      for (int k = 0; k < 28; k++, r += 2, row++) {        // two partial rounds per row; a2 = the first s-box output
        put_state(out, row, s0, s1);
        s0 = pow5(f29_add(s0, lds_f29(tab.rc, 2 * r)));
        store_hat_w(out.col[2] + row, s0);
        s1 = f29_add(s1, lds_f29(tab.rc, 2 * r + 1));
        mix(s0, s1, &tab);
        s0 = pow5(f29_add(s0, lds_f29(tab.rc, 2 * r + 2)));
        s1 = f29_add(s1, lds_f29(tab.rc, 2 * r + 3));
        mix(s0, s1, &tab);
      }
      for (int k = 0; k < 4; k++, r++, row++) {
        put_state(out, row, s0, s1);
        s0 = pow5(f29_add(s0, lds_f29(tab.rc, 2 * r)));
        s1 = pow5(f29_add(s1, lds_f29(tab.rc, 2 * r + 1)));
        mix(s0, s1, &tab);
      }
      put_state(out, row, s0, s1);                        // offset 36: the state after the permutation
    }
  }
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
    e = host_wait_stream(stream);
  }
  retire_device_memory(d_rc);
  retire_device_memory(d_mds);
  return e;
}
void WitnessEngine::release() {
  if (table_) (void)hipFree(table_);
  table_ = nullptr;
}
hipError_t WitnessEngine::inclusion_witness(const uint32_t* d_program, uint32_t n_items, uint32_t n_absorbs, const fp_words* users,
                                            const fp_words* hashes, const fp_words* balances, uint32_t depth, uint32_t nc,
                                            const uint32_t* d_user_index, uint32_t n_users, fp_words* advice, size_t rows,
                                            hipStream_t stream) {
  if (!n_items || !n_users) return hipSuccess;
  WitnessTree t{users, hashes, balances, depth, nc, {}};
  uint32_t off = 0;
  for (uint32_t l = 0; l <= depth; l++) {
    t.level_offset[l] = off;
    off += 1u << (depth - l);
  }
  (void)n_absorbs;
  const WitnessItem* items = reinterpret_cast<const WitnessItem*>(d_program);
  const WitnessAbsorb* absorbs = reinterpret_cast<const WitnessAbsorb*>(d_program + 5 * (size_t)n_items);
  dim3 grid((n_items + 63) / 64, n_users);
  mst_inclusion_witness_kernel<<<grid, 64, 0, stream>>>(items, n_items, absorbs, t, d_user_index, n_users,
                                                       static_cast<const PoseidonTable*>(table_), advice, rows, 3 * rows);
  return hipGetLastError();
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
