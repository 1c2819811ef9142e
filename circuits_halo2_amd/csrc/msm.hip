// Pippenger bucket MSM over BN254 G1 for gfx950 -- the GPU side of halo2's `best_multiexp`
// (reached through ParamsKZG::commit / commit_lagrange from the reference's
// zk_prover/src/circuits/utils.rs:75,76,94-101,171-178).
//
//   result = sum_i s_i * P_i,  s_i: 32-B Montgomery Fr, P_i: 64-B affine Montgomery points.
//
// Pipeline (all on one stream; one host read-back of three counters to size the launches):
//   1 msm_digits     scalars -> canonical -> signed c-bit digits; one key per (window, scalar);
//                    bucket histogram (zero digits are skipped, as in halo2)
//   2 msm_scan       exclusive scans: bucket offsets and task offsets (a task = <= L
//                    consecutive entries of one bucket, so heavy buckets are split)
//   3 msm_scatter    counting sort of the point indices by (window, bucket)
//   4 msm_accumulate one thread per task: XYZZ accumulator += affine points (8M+2S each)
//     msm_merge      (only when a bucket had > L entries) same over partial sums
//   5 msm_reduce     sum_b b*B_b per window: per-thread running sums over G buckets, then a
//                    workgroup-wide suffix scan + tree reduction in LDS; repeated per level
//   6 host           Horner over the W window sums (c doublings each) + affine normalisation
//
// Signed digits halve the bucket count: digit d in [-2^(c-1), 2^(c-1)], bucket |d|, the point
// is negated on the fly when d < 0.  The group law is commutative, so the order in which a
// bucket's points are added (atomics make it non-deterministic) never changes the result bits.
#include "msm.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

#include "host_curve.h"

namespace sg {

static constexpr uint32_t EMPTY_KEY = 0xffffffffu;

// ------------------------------------------------------------------ 1: digits + histogram
__global__ void msm_digits(const fp_t* __restrict__ scalars, uint32_t n, uint32_t c, uint32_t W,
                           uint32_t* __restrict__ keys, uint32_t* __restrict__ counts) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fp_t s = fp_from_mont<FrP>(fp_load(scalars + i));
  const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
  uint32_t carry = 0;
  for (uint32_t j = 0; j < W; j++) {
    uint32_t v = (s.l[0] & mask) + carry;
    // s >>= c
#pragma unroll
    for (int k = 0; k < 7; k++) s.l[k] = (s.l[k] >> c) | (s.l[k + 1] << (32 - c));
    s.l[7] >>= c;
    uint32_t key;
    if (j + 1 < W && v > half) {
      v = (1u << c) - v;  // |d| of the negative digit, in [0, half)
      carry = 1;
      key = v ? (0x80000000u | (j * half + v - 1)) : EMPTY_KEY;
    } else {
      carry = 0;
      key = v ? (j * half + v - 1) : EMPTY_KEY;
    }
    keys[(size_t)j * n + i] = key;
    if (key != EMPTY_KEY) atomicAdd(&counts[key & 0x7fffffffu], 1u);
  }
}

// ------------------------------------------------------------------ 2: scans
// One workgroup.  cnt[NB] -> off[NB+1] (exclusive scan, optional), ntask[b] = ceil(cnt/L),
// toff[NB+1] (exclusive scan of ntask), meta = {sum cnt, sum ntask, max cnt}.
__global__ void __launch_bounds__(1024) msm_scan(const uint32_t* __restrict__ cnt, uint32_t NB, uint32_t log_L,
                                                 uint32_t* __restrict__ off, uint32_t* __restrict__ ntask,
                                                 uint32_t* __restrict__ toff, uint32_t* __restrict__ meta) {
  __shared__ uint32_t s_cnt[1024], s_tsk[1024], s_max[1024];
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t per = (NB + nthr - 1) / nthr;
  const uint32_t lo = min(tid * per, NB), hi = min(lo + per, NB);
  const uint32_t Lm1 = (1u << log_L) - 1;
  uint32_t a = 0, t = 0, m = 0;
  for (uint32_t b = lo; b < hi; b++) {
    uint32_t cval = cnt[b];
    a += cval;
    t += (cval + Lm1) >> log_L;
    m = max(m, cval);
  }
  s_cnt[tid] = a; s_tsk[tid] = t; s_max[tid] = m;
  __syncthreads();
  // Hillis-Steele inclusive scan over the per-thread totals
  for (uint32_t d = 1; d < nthr; d <<= 1) {
    uint32_t va = 0, vt = 0, vm = 0;
    if (tid >= d) { va = s_cnt[tid - d]; vt = s_tsk[tid - d]; vm = s_max[tid - d]; }
    __syncthreads();
    s_cnt[tid] += va; s_tsk[tid] += vt; s_max[tid] = max(s_max[tid], vm);
    __syncthreads();
  }
  uint32_t base_a = s_cnt[tid] - a, base_t = s_tsk[tid] - t;
  for (uint32_t b = lo; b < hi; b++) {
    uint32_t cval = cnt[b];
    uint32_t nt = (cval + Lm1) >> log_L;
    if (off) off[b] = base_a;
    ntask[b] = nt;
    toff[b] = base_t;
    base_a += cval;
    base_t += nt;
  }
  if (tid == nthr - 1) {
    if (off) off[NB] = s_cnt[tid];
    toff[NB] = s_tsk[tid];
    meta[0] = s_cnt[tid];
    meta[1] = s_tsk[tid];
    meta[2] = s_max[tid];
  }
}

// ------------------------------------------------------------------ 3: scatter
__global__ void msm_scatter(const uint32_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ off,
                            uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t key = keys[(size_t)blockIdx.y * n + i];
  if (key == EMPTY_KEY) return;
  uint32_t b = key & 0x7fffffffu;
  uint32_t pos = atomicAdd(&cursor[b], 1u);
  sorted[off[b] + pos] = i | (key & 0x80000000u);
}

// ------------------------------------------------------------------ 4: accumulate / merge
__device__ __forceinline__ uint32_t find_owner(const uint32_t* __restrict__ toff, uint32_t NB, uint32_t t) {
  uint32_t lo = 0, hi = NB;  // invariant: toff[lo] <= t < toff[hi]
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (toff[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256) msm_accumulate(const uint32_t* __restrict__ sorted,
                                                      const g1_affine* __restrict__ bases,
                                                      const uint32_t* __restrict__ off,
                                                      const uint32_t* __restrict__ cnt,
                                                      const uint32_t* __restrict__ toff, uint32_t NB, uint32_t log_L,
                                                      uint32_t ntasks, g1_xyzz* __restrict__ partial) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntasks) return;
  uint32_t b = find_owner(toff, NB, t);
  uint32_t seg = t - toff[b];
  uint32_t start = off[b] + (seg << log_L);
  uint32_t end = min(off[b] + cnt[b], start + (1u << log_L));
  g1_xyzz acc = xyzz_identity();
  uint32_t e = sorted[start];
  g1_affine p = affine_load(bases + (e & 0x7fffffffu));
  for (uint32_t k = start; k < end; k++) {
    uint32_t e_next = 0;
    g1_affine p_next;
    if (k + 1 < end) {  // prefetch the next point while this one is being added
      e_next = sorted[k + 1];
      p_next = affine_load(bases + (e_next & 0x7fffffffu));
    }
    if (e >> 31) p.y = fp_neg<FqP>(p.y);
    xyzz_madd(acc, p);
    e = e_next;
    p = p_next;
  }
  xyzz_store(partial + t, acc);
}

__global__ void __launch_bounds__(256) msm_merge(const g1_xyzz* __restrict__ in, const uint32_t* __restrict__ off,
                                                 const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ toff,
                                                 uint32_t NB, uint32_t log_L, uint32_t ntasks,
                                                 g1_xyzz* __restrict__ out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntasks) return;
  uint32_t b = find_owner(toff, NB, t);
  uint32_t seg = t - toff[b];
  uint32_t start = off[b] + (seg << log_L);
  uint32_t end = min(off[b] + cnt[b], start + (1u << log_L));
  g1_xyzz acc = xyzz_identity();
  for (uint32_t k = start; k < end; k++) xyzz_add(acc, xyzz_load(in + k));
  xyzz_store(out + t, acc);
}

// ------------------------------------------------------------------ 5: bucket reduction
// Items are (acc, run) pairs in bucket order; item t of a window stands for
// acc_t + (t * M) * run_t.  A workgroup of N items produces one item of the next level:
//   acc' = sum_t acc_t + M * sum_{t>=1} Suf_t,  run' = Suf_0,  Suf_t = sum_{u>=t} run_u.
__device__ void block_combine(g1_xyzz acc, g1_xyzz run, uint32_t log_M, g1_xyzz* sA, g1_xyzz* sR,
                              g1_xyzz* out_acc, g1_xyzz* out_run) {
  const uint32_t tid = threadIdx.x, N = blockDim.x;
  xyzz_store(&sR[tid], run);
  __syncthreads();
  for (uint32_t d = 1; d < N; d <<= 1) {
    g1_xyzz other = xyzz_identity();
    if (tid + d < N) other = xyzz_load(&sR[tid + d]);
    __syncthreads();
    xyzz_add(run, other);
    xyzz_store(&sR[tid], run);
    __syncthreads();
  }
  g1_xyzz total = xyzz_load(&sR[0]);
  __syncthreads();
  xyzz_store(&sA[tid], acc);
  if (tid == 0) xyzz_store(&sR[0], xyzz_identity());
  __syncthreads();
  // two tree reductions side by side: lower half of the threads folds sA, upper half sR
  const uint32_t halfN = N >> 1;
  g1_xyzz* arr = (tid < halfN) ? sA : sR;
  const uint32_t li = (tid < halfN) ? tid : tid - halfN;
  for (uint32_t s = halfN; s >= 1; s >>= 1) {
    if (li < s) {
      g1_xyzz a = xyzz_load(&arr[li]);
      xyzz_add(a, xyzz_load(&arr[li + s]));
      xyzz_store(&arr[li], a);
    }
    __syncthreads();
  }
  if (tid == 0) {
    g1_xyzz sufsum = xyzz_load(&sR[0]);
    for (uint32_t k = 0; k < log_M; k++) sufsum = xyzz_double(sufsum);
    g1_xyzz a = xyzz_load(&sA[0]);
    xyzz_add(a, sufsum);
    xyzz_store(out_acc, a);
    xyzz_store(out_run, total);
  }
}

// level 0: thread -> G = 2^log_G consecutive buckets of window blockIdx.y
__global__ void __launch_bounds__(256) msm_reduce_buckets(const g1_xyzz* __restrict__ partial,
                                                          const uint32_t* __restrict__ toff,
                                                          const uint32_t* __restrict__ ntask, uint32_t nbw,
                                                          uint32_t log_G, g1_xyzz* __restrict__ out_acc,
                                                          g1_xyzz* __restrict__ out_run) {
  extern __shared__ uint4 smem[];
  g1_xyzz* sA = reinterpret_cast<g1_xyzz*>(smem);
  g1_xyzz* sR = sA + blockDim.x;
  const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t G = 1u << log_G;
  g1_xyzz acc = xyzz_identity(), run = xyzz_identity();
  const uint32_t first = chunk << log_G;
  if (first < nbw) {
    const uint32_t wbase = blockIdx.y * nbw;
    for (uint32_t k = G; k-- > 0;) {
      uint32_t b = first + k;
      if (b < nbw && ntask[wbase + b]) xyzz_add(run, xyzz_load(partial + toff[wbase + b]));
      xyzz_add(acc, run);
    }
  }
  const uint32_t o = blockIdx.y * gridDim.x + blockIdx.x;
  block_combine(acc, run, log_G, sA, sR, out_acc + o, out_run + o);
}
// level >= 1: items from the previous level, `count` per window
__global__ void __launch_bounds__(256) msm_reduce_items(const g1_xyzz* __restrict__ in_acc,
                                                        const g1_xyzz* __restrict__ in_run, uint32_t count,
                                                        uint32_t log_M, g1_xyzz* __restrict__ out_acc,
                                                        g1_xyzz* __restrict__ out_run) {
  extern __shared__ uint4 smem[];
  g1_xyzz* sA = reinterpret_cast<g1_xyzz*>(smem);
  g1_xyzz* sR = sA + blockDim.x;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  g1_xyzz acc = xyzz_identity(), run = xyzz_identity();
  if (t < count) {
    acc = xyzz_load(in_acc + blockIdx.y * count + t);
    run = xyzz_load(in_run + blockIdx.y * count + t);
  }
  const uint32_t o = blockIdx.y * gridDim.x + blockIdx.x;
  block_combine(acc, run, log_M, sA, sR, out_acc + o, out_run + o);
}

// out[i] = scalars[i] * G  (ParamsKZG::setup's fixed-base products; also used to build
// synthetic bases for benchmarks).  One thread per scalar, double-and-add, Jacobian-free
// XYZZ, result normalised on the device.
__global__ void __launch_bounds__(256) g1_fixed_base_mul(const fp_t* __restrict__ scalars, uint32_t n,
                                                         g1_affine* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fp_t s = fp_from_mont<FrP>(fp_load(scalars + i));
  g1_affine gen;
  gen.x = fp_one<FqP>();
  gen.y = fp_dbl<FqP>(gen.x);
  g1_xyzz acc = xyzz_identity();
  for (int limb = 7; limb >= 0; limb--) {
    uint32_t w = s.l[7];
#pragma unroll
    for (int k = 7; k > 0; k--) s.l[k] = s.l[k - 1];
    s.l[0] = 0;
    for (int bit = 31; bit >= 0; bit--) {
      acc = xyzz_double(acc);
      if ((w >> bit) & 1) xyzz_madd(acc, gen);
    }
    (void)limb;
  }
  g1_affine r;
  if (xyzz_is_identity(acc)) {
    r.x = fp_zero<FqP>();
    r.y = fp_zero<FqP>();
  } else {
    fp_t iz = fp_inv<FqP>(acc.zzz);             // 1/ZZZ
    fp_t t = fp_mul<FqP>(acc.zz, iz);           // ZZ/ZZZ = 1/Z
    r.x = fp_mul<FqP>(acc.x, fp_sqr<FqP>(t));   // X/ZZ
    r.y = fp_mul<FqP>(acc.y, iz);               // Y/ZZZ
  }
  fp_store(&out[i].x, r.x);
  fp_store(&out[i].y, r.y);
}

// ------------------------------------------------------------------ host driver
#define SG_TRY(x)                      \
  do {                                 \
    hipError_t _e = (x);               \
    if (_e != hipSuccess) return _e;   \
  } while (0)

MsmEngine::~MsmEngine() { release(); }

void MsmEngine::release() {
  keys_.release(); sorted_.release(); counts_.release(); off_.release(); cursor_.release(); meta_.release();
  for (int i = 0; i < 2; i++) {
    ntask_[i].release(); toff_[i].release(); partial_[i].release(); red_acc_[i].release(); red_run_[i].release();
  }
  if (h_meta_) (void)hipHostFree(h_meta_);
  if (h_win_) (void)hipHostFree(h_win_);
  h_meta_ = nullptr;
  h_win_ = nullptr;
}

uint32_t MsmEngine::window_bits_for(size_t n) const {
  if (cfg_.window_bits) return std::min<uint32_t>(16, std::max<uint32_t>(2, cfg_.window_bits));
  uint32_t lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) lg++;
  int c = (int)lg - 4;
  return (uint32_t)std::min(16, std::max(4, c));
}

hipError_t MsmEngine::init() {
  SG_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_reduce_buckets),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  SG_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_reduce_items),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  return hipSuccess;
}

hipError_t MsmEngine::run(const fp_t* d_scalars, const g1_affine* d_bases, size_t n, hipStream_t stream,
                          uint8_t out_affine[64], MsmTimings* tm) {
  if (tm) *tm = MsmTimings{};
  if (n == 0) {
    std::memset(out_affine, 0, 64);
    return hipSuccess;
  }
  if (n >= (1ull << 31)) return hipErrorInvalidValue;
  const uint32_t c = window_bits_for(n);
  const uint32_t W = (255 + c - 1) / c;  // W*c >= 255: the top window never needs a carry out
  const uint32_t nbw = 1u << (c - 1);
  const uint32_t NB = W * nbw;
  const uint32_t log_L = cfg_.log_seg;
  const size_t entries = (size_t)W * n;

  // workspace (grown on demand, kept across calls)
  SG_TRY(keys_.reserve(entries));
  SG_TRY(sorted_.reserve(entries));
  SG_TRY(counts_.reserve((size_t)NB + 1));
  SG_TRY(off_.reserve((size_t)NB + 1));
  SG_TRY(cursor_.reserve((size_t)NB + 1));
  for (int i = 0; i < 2; i++) {
    SG_TRY(ntask_[i].reserve((size_t)NB + 1));
    SG_TRY(toff_[i].reserve((size_t)NB + 1));
  }
  SG_TRY(meta_.reserve(16));
  if (!h_meta_) SG_TRY(hipHostMalloc(&h_meta_, 16 * sizeof(uint32_t)));
  if (!h_win_) SG_TRY(hipHostMalloc(&h_win_, 64 * sizeof(g1_xyzz)));

  hipEvent_t ev[5];
  if (tm) {
    for (auto& e : ev) SG_TRY(hipEventCreate(&e));
    SG_TRY(hipEventRecord(ev[0], stream));
  }
  auto drop_events = [&]() {
    if (tm) {
      for (auto& e : ev) (void)hipEventDestroy(e);
    }
  };

  SG_TRY(hipMemsetAsync(counts_.p, 0, sizeof(uint32_t) * (NB + 1), stream));
  SG_TRY(hipMemsetAsync(cursor_.p, 0, sizeof(uint32_t) * (NB + 1), stream));
  msm_digits<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d_scalars, (uint32_t)n, c, W, keys_.p, counts_.p);
  if (tm) SG_TRY(hipEventRecord(ev[1], stream));
  msm_scan<<<1, 1024, 0, stream>>>(counts_.p, NB, log_L, off_.p, ntask_[0].p, toff_[0].p, meta_.p);
  SG_TRY(hipMemcpyAsync(h_meta_, meta_.p, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  msm_scatter<<<dim3((unsigned)((n + 255) / 256), W), 256, 0, stream>>>(keys_.p, (uint32_t)n, off_.p, cursor_.p,
                                                                        sorted_.p);
  if (tm) SG_TRY(hipEventRecord(ev[2], stream));
  SG_TRY(hipStreamSynchronize(stream));
  const uint32_t ntasks = h_meta_[1], max_cnt = h_meta_[2];
  if (!ntasks) {  // every digit was zero
    std::memset(out_affine, 0, 64);
    drop_events();
    return hipSuccess;
  }

  // bucket b owns cur[toff_[lvl][b] .. +ntask_[lvl][b])
  SG_TRY(partial_[0].reserve(ntasks));
  msm_accumulate<<<(ntasks + 255) / 256, 256, 0, stream>>>(sorted_.p, d_bases, off_.p, counts_.p, toff_[0].p, NB, log_L,
                                                           ntasks, partial_[0].p);
  const g1_xyzz* cur = partial_[0].p;
  int lvl = 0, pbuf = 0;
  // heavy buckets: fold their partial sums until every bucket owns at most one
  for (uint32_t max_items = (max_cnt + (1u << log_L) - 1) >> log_L; max_items > 1;
       max_items = (max_items + (1u << log_L) - 1) >> log_L) {
    const int nxt = 1 - lvl;
    msm_scan<<<1, 1024, 0, stream>>>(ntask_[lvl].p, NB, log_L, nullptr, ntask_[nxt].p, toff_[nxt].p, meta_.p);
    SG_TRY(hipMemcpyAsync(h_meta_, meta_.p, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    SG_TRY(hipStreamSynchronize(stream));
    const uint32_t nt2 = h_meta_[1];
    SG_TRY(partial_[1 - pbuf].reserve(nt2));
    msm_merge<<<(nt2 + 255) / 256, 256, 0, stream>>>(cur, toff_[lvl].p, ntask_[lvl].p, toff_[nxt].p, NB, log_L, nt2,
                                                     partial_[1 - pbuf].p);
    pbuf = 1 - pbuf;
    cur = partial_[pbuf].p;
    lvl = nxt;
  }
  if (tm) SG_TRY(hipEventRecord(ev[3], stream));

  // bucket reduction levels
  const uint32_t log_G = std::min<uint32_t>(cfg_.log_red_chunk, c - 1);
  uint32_t items = nbw >> log_G;  // chunks per window at level 0 (a power of two)
  uint32_t threads = std::min<uint32_t>(256, std::max<uint32_t>(64, items));
  uint32_t blocks = (items + threads - 1) / threads;
  for (int i = 0; i < 2; i++) {
    SG_TRY(red_acc_[i].reserve((size_t)W * blocks));
    SG_TRY(red_run_[i].reserve((size_t)W * blocks));
  }
  size_t lds = (size_t)threads * 2 * sizeof(g1_xyzz);
  msm_reduce_buckets<<<dim3(blocks, W), threads, lds, stream>>>(cur, toff_[lvl].p, ntask_[lvl].p, nbw, log_G,
                                                                 red_acc_[0].p, red_run_[0].p);
  uint32_t log_M = log_G;
  int src = 0;
  while (blocks > 1) {
    // one item of this level spans `threads` items of the previous one
    uint32_t lt = 0;
    while ((1u << lt) < threads) lt++;
    log_M += lt;
    items = blocks;
    threads = 64;
    while (threads < items && threads < 256) threads <<= 1;
    blocks = (items + threads - 1) / threads;
    lds = (size_t)threads * 2 * sizeof(g1_xyzz);
    msm_reduce_items<<<dim3(blocks, W), threads, lds, stream>>>(red_acc_[src].p, red_run_[src].p, items, log_M,
                                                                 red_acc_[1 - src].p, red_run_[1 - src].p);
    src = 1 - src;
  }
  if (tm) SG_TRY(hipEventRecord(ev[4], stream));
  SG_TRY(hipMemcpyAsync(h_win_, red_acc_[src].p, sizeof(g1_xyzz) * W, hipMemcpyDeviceToHost, stream));
  SG_TRY(hipStreamSynchronize(stream));

  // Horner over the window sums, high to low
  using namespace host;
  Jac total = Jac::identity();
  for (int j = (int)W - 1; j >= 0; j--) {
    for (uint32_t k = 0; k < c; k++) total = jac_double(total);
    Fq x, y, zz, zzz;
    std::memcpy(x.v, &h_win_[j].x, 32);
    std::memcpy(y.v, &h_win_[j].y, 32);
    std::memcpy(zz.v, &h_win_[j].zz, 32);
    std::memcpy(zzz.v, &h_win_[j].zzz, 32);
    total = jac_add(total, jac_from_xyzz(x, y, zz, zzz));
  }
  jac_to_affine_bytes(total, out_affine);

  if (tm) {
    float ms;
    (void)hipEventElapsedTime(&ms, ev[0], ev[1]); tm->digits_ms = ms;
    (void)hipEventElapsedTime(&ms, ev[1], ev[2]); tm->sort_ms = ms;
    (void)hipEventElapsedTime(&ms, ev[2], ev[3]); tm->accumulate_ms = ms;
    (void)hipEventElapsedTime(&ms, ev[3], ev[4]); tm->reduce_ms = ms;
    (void)hipEventElapsedTime(&ms, ev[0], ev[4]); tm->total_ms = ms;
    tm->window_bits = c;
    tm->windows = W;
    tm->tasks = ntasks;
    tm->max_bucket = max_cnt;
    drop_events();
  }
  return hipSuccess;
}

hipError_t fixed_base_mul(const fp_t* d_scalars, size_t n, g1_affine* d_out, hipStream_t stream) {
  if (!n) return hipSuccess;
  g1_fixed_base_mul<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d_scalars, (uint32_t)n, d_out);
  return hipGetLastError();
}

}  // namespace sg
