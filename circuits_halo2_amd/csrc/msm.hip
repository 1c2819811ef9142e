// Pippenger bucket MSM over BN254 G1 for gfx950 -- the GPU side of halo2's `best_multiexp`
// (reached through ParamsKZG::commit / commit_lagrange from the reference's
// zk_prover/src/circuits/utils.rs:75,76,94-101,171-178).
//
//   result = sum_i s_i * P_i,  s_i: 32-B Montgomery Fr, P_i: 64-B affine Montgomery points.
//
// Pipeline (all on one stream; one host read-back of three counters to size the launches):
//   1 msm_digits      scalars -> canonical -> W signed c-bit digits (int16 rows, one per window)
//   2 msm_hist        LDS-staged window buckets: workgroup (chunk p, window j) histograms its
//                     chunk's digits in LDS (2^(c-1) counters, up to 128 KiB); zero digits are
//                     skipped, as in halo2
//     msm_hist_prefix per-bucket prefix over chunks -> bucket counts
//   3 msm_scan_*      exclusive scans: bucket offsets and task offsets (a task = <= L
//                     consecutive entries of one bucket, so heavy buckets are split)
//     msm_scatter     counting sort of the point indices by (window, bucket); slots are handed
//                     out by LDS atomics on per-workgroup cursors (no global atomics in the sort)
//   4 msm_accumulate  one lane per task: XYZZ accumulator += affine points (8M+2S each); persistent waves that
//                     take tickets of 64 tasks from one global counter
//     msm_merge       (only when a bucket had > L entries) same over partial sums
//   5 msm_reduce_*    sum_b b*B_b per window: per-thread running sums over G buckets, then a
//                     workgroup-wide suffix scan + tree reduction in LDS; repeated per level
//   6 host            Horner over the W window sums (c doublings each) + affine normalisation
//
// Signed digits halve the bucket count: digit d in [-2^(c-1), 2^(c-1)], bucket |d|, the point
// is negated on the fly when d < 0.  The group law is commutative, so the order in which a
// bucket's points are added (LDS atomics make it non-deterministic) never changes the result bits.
#include "msm.h"
#include "side_prio.cuh"
#include "host_wait.h"

#include <algorithm>
#include <atomic>
#include <functional>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <vector>

#include "host_curve.h"

namespace sg {
SG_DEFINE_SIDE_PRIO_SETTER(msm_set_side_prio)


// ------------------------------------------------------------------ 1: signed digits
// Windows have individual widths (WindowPlan): W-1 signed windows of c or c-1 bits and an
// unsigned top window of at most c-1 bits, widths summing to exactly 254, so every window
// spreads its points over (almost) the same number of buckets -- a leftover-bits top window
// would put n / 2^t points in each of its 2^t buckets.
// dig[j*n + i] = digit j of scalar i as int16.  Adding K = sum_{j<W-1} 2^(o_j + w_j - 1)
// once makes every window's digit independent of its neighbours:
//   d_j = (((s + K) >> o_j) & (2^w_j - 1)) - 2^(w_j - 1)   in [-2^(w_j-1), 2^(w_j-1)).
// blockIdx.y = m selects the scalar vector of a fused batch (BatchPtrs); its digit rows are
// dig[(m*W + j)*n + i].
__global__ void msm_digits(BatchPtrs bp, uint32_t n, WindowPlan wp, int16_t* __restrict__ dig) {
  side_kernel_prio();
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const fp_words* __restrict__ scalars = bp.scalars[blockIdx.y];
  dig += (size_t)blockIdx.y * wp.W * n;
  words8 s;
  {
    // canonical scalar = s~ * 2^-256 = s~ * 2^5 * 2^-261
    f29 k = f29_zero();
    k.l[0] = 32;
    f29 v = f29_load_r256<Fr29>(scalars + i);                  // any 256-bit word value: bound < 6
    if ((bp.diff_mask >> blockIdx.y) & 1ull) {
      // difference form: the scalar of row i is s[i] - s[i+1] (s[n] = 0), against the prefix-summed basis
      if (i + 1 < n) v = f29_sub<Fr29, 2>(v, f29_load_r256<Fr29>(scalars + i + 1));   // + 8r: bound < 14
    }
    f29_to_words(f29_cond_sub_p<Fr29>(f29_mul<Fr29>(v, k)), s.l);
  }
  const uint32_t W = wp.W;
  // s += K (K < 2^254, s < 2^254: no overflow out of 256 bits)
  {
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t o = 0;
    for (uint32_t j = 0; j + 1 < W; j++) {
      uint32_t bit = o + wp.width[j] - 1;
      uint32_t m = 1u << (bit & 31);
      uint32_t q = bit >> 5;
#pragma unroll
      for (int t = 0; t < 8; t++) k[t] |= (q == (uint32_t)t) ? m : 0u;
      o += wp.width[j];
    }
    uint32_t carry = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      uint64_t t = (uint64_t)s.l[q] + k[q] + carry;
      s.l[q] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
  }
  for (uint32_t j = 0; j < W; j++) {
    const uint32_t w = wp.width[j];
    uint32_t v = s.l[0] & ((1u << w) - 1);
#pragma unroll
    for (int q = 0; q < 7; q++) s.l[q] = (s.l[q] >> w) | (s.l[q + 1] << (32 - w));
    s.l[7] >>= w;
    int32_t d = (j + 1 < W) ? (int32_t)v - (int32_t)(1u << (w - 1)) : (int32_t)v;
    dig[(size_t)j * n + i] = (int16_t)d;
  }
}

// ------------------------------------------------------------------ 2: LDS-staged histogram
// grid (W, P): workgroup (j, p) counts the digits of scalar chunk p for window j in an LDS
// histogram of 2^(c-1) buckets, then stores it to hist[(j*P + p)*nbw + b].  The window is the
// fast grid index so that (workgroups being dealt round-robin to the 8 XCDs) the chunks of a
// window share an XCD's L2; measured neutral for msm_scatter's 4-byte scattered stores
// (WRITE_SIZE stays ~8x the useful bytes), kept because it costs nothing.
// `shift` > 0 histograms coarse bins (bucket >> shift) for the two-pass sort; nbw = bins per row.
__global__ void __launch_bounds__(1024) msm_hist(const int16_t* __restrict__ dig, uint32_t n, uint32_t chunk,
                                                 uint32_t nbw, uint32_t shift, uint32_t* __restrict__ hist) {
  side_kernel_prio();
  extern __shared__ uint32_t s_cnt[];
  const uint32_t j = blockIdx.x, p = blockIdx.y, P = gridDim.y;
  for (uint32_t b = threadIdx.x; b < nbw; b += blockDim.x) s_cnt[b] = 0;
  __syncthreads();
  const uint32_t lo = p * chunk, hi = min(n, lo + chunk);
  const int16_t* row = dig + (size_t)j * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int32_t d = row[i];
    if (d) atomicAdd(&s_cnt[((uint32_t)(d < 0 ? -d : d) - 1u) >> shift], 1u);
  }
  __syncthreads();
  uint32_t* out = hist + ((size_t)j * P + p) * nbw;
  for (uint32_t b = threadIdx.x; b < nbw; b += blockDim.x) out[b] = s_cnt[b];
}
// per bucket: exclusive prefix over the P chunks (in place) and the bucket total.  Block of
// 32 buckets x 8 chunk groups: every thread sums its group's share of the column, the group
// bases come from LDS, then a second sweep writes the prefixes (a column is P strided loads;
// one thread per bucket made this the slowest kernel of a small MSM).
static constexpr uint32_t HP_BUCKETS = 32, HP_GROUPS = 8;
// Round 5 (`fe` != nullptr, coarse bins of the two-pass sort, NB <= FE_MAX_BINS): the workgroup that finishes LAST (a counter in
// device memory, as msm_scan_sums does for its block sums) also scans the bin totals -- coff[g] = entries before bin g and
// tbase[g] = task slots before bin g, a bin of F buckets and E entries owning F + (E >> log_L) slots (an upper bound of its
// sum_f ceil(c_f / L) tasks) -- which used to be a launch of its own (msm_scan_small / msm_scan_sums + msm_scan_write).
static constexpr uint32_t FE_MAX_BINS = 4096;
struct FrontEndScan {
  uint32_t* done;    // counter of finished workgroups (zero between launches), nullptr: no scan here
  uint32_t* coff;    // [NB + 1]
  uint32_t* tbase;   // [NB + 1]
  uint32_t F, log_L;
};
__global__ void __launch_bounds__(256) msm_hist_prefix(uint32_t* __restrict__ hist, uint32_t P, uint32_t nbw, uint32_t NB,
                                                       uint32_t* __restrict__ counts, FrontEndScan fe) {
  side_kernel_prio();
  __shared__ uint32_t s_sum[HP_GROUPS][HP_BUCKETS];
  __shared__ uint32_t s_last;
  const uint32_t bx = threadIdx.x % HP_BUCKETS, gy = threadIdx.x / HP_BUCKETS;
  const uint32_t g = blockIdx.x * HP_BUCKETS + bx;  // global bucket id = j*nbw + b
  const bool live = g < NB;
  const uint32_t j = live ? g / nbw : 0, b = live ? g - j * nbw : 0;
  uint32_t* col = hist + (size_t)j * P * nbw + b;
  const uint32_t per = (P + HP_GROUPS - 1) / HP_GROUPS;
  const uint32_t lo = min(gy * per, P), hi = min(lo + per, P);
  uint32_t sum = 0;
  if (live)
    for (uint32_t p = lo; p < hi; p++) sum += col[(size_t)p * nbw];
  s_sum[gy][bx] = sum;
  __syncthreads();
  uint32_t run = 0;
  for (uint32_t q = 0; q < gy; q++) run += s_sum[q][bx];
  if (live) {
    for (uint32_t p = lo; p < hi; p++) {
      uint32_t v = col[(size_t)p * nbw];
      col[(size_t)p * nbw] = run;
      run += v;
    }
    if (gy == HP_GROUPS - 1) {
      if (fe.done) {   // a returning exchange: performed at the memory side once the value is back (no release fence needed below)
        const uint32_t was = atomicExch(&counts[g], run);
        asm volatile("" ::"v"(was));
      } else {
        counts[g] = run;
      }
    }
  }
  if (!fe.done) return;
  __syncthreads();                       // (every thread of the workgroup reaches this: the early return above is gone)
  if (threadIdx.x == 0) s_last = atomicAdd(fe.done, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  // the last workgroup: exclusive scans over the NB <= 4096 bin totals, 16 per thread (atomic reads of what the other
  // workgroups' exchanges left)
  __shared__ uint32_t s_a[256], s_t[256];
  const uint32_t tid = threadIdx.x, per_t = (NB + 255) / 256, b0 = min(tid * per_t, NB), b1 = min(b0 + per_t, NB);
  uint32_t vals[FE_MAX_BINS / 256];
  uint32_t a = 0, t = 0;
#pragma unroll
  for (uint32_t q = 0; q < FE_MAX_BINS / 256; q++) vals[q] = (b0 + q < b1) ? atomicAdd(&counts[b0 + q], 0u) : 0u;   // all in flight together
#pragma unroll
  for (uint32_t q = 0; q < FE_MAX_BINS / 256; q++) {
    if (b0 + q < b1) {
      a += vals[q];
      t += fe.F + (vals[q] >> fe.log_L);
    }
  }
  s_a[tid] = a; s_t[tid] = t;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    uint32_t xa = 0, xt = 0;
    if (tid >= d) { xa = s_a[tid - d]; xt = s_t[tid - d]; }
    __syncthreads();
    s_a[tid] += xa; s_t[tid] += xt;
    __syncthreads();
  }
  uint32_t ra = s_a[tid] - a, rt = s_t[tid] - t;
#pragma unroll
  for (uint32_t q = 0; q < FE_MAX_BINS / 256; q++) {
    if (b0 + q < b1) {
      fe.coff[b0 + q] = ra;
      fe.tbase[b0 + q] = rt;
      ra += vals[q];
      rt += fe.F + (vals[q] >> fe.log_L);
    }
  }
  if (tid == 255) {
    fe.coff[NB] = s_a[255];
    fe.tbase[NB] = s_t[255];
    atomicExch(fe.done, 0u);             // ready for the next launch on this stream
  }
}

// ------------------------------------------------------------------ 3: scans (multi-block)
// cnt[NB] -> off[NB+1] (exclusive scan, optional), ntask[b] = ceil(cnt[b]/L), toff[NB+1]
// (exclusive scan of ntask), meta = {sum cnt, sum ntask, max cnt}.  2048 buckets per block.
static constexpr uint32_t SCAN_ITEMS = 8, SCAN_THREADS = 256, SCAN_BLOCK = SCAN_ITEMS * SCAN_THREADS;

// Round 4: the scan of the (<= 1024) block sums is done by whichever workgroup of this launch finishes LAST (a counter in
// device memory, zero between launches: the last arrival resets it; no workgroup ever waits for another) -- the separate
// one-workgroup launch of rounds 1-3 is gone, one ~5.5 us launch less per scan, two scans per MSM job.
// The totals land in off[NB] (optional), toff[NB], meta[0..2] and, when `host_meta` is given, straight in page-locked host
// memory the device can write (the host reads them after an event, no copy kernel in between).
static constexpr uint32_t SCAN_DONE = 12;         // word of meta_ that counts the finished workgroups of msm_scan_sums
__global__ void __launch_bounds__(256) msm_scan_sums(const uint32_t* __restrict__ cnt, uint32_t NB, uint32_t log_L,
                                                     uint32_t* __restrict__ bsum, uint32_t* __restrict__ meta,
                                                     uint32_t* __restrict__ off, uint32_t* __restrict__ toff,
                                                     volatile uint32_t* host_meta) {
  side_kernel_prio();
  __shared__ uint32_t s_a[SCAN_THREADS], s_t[SCAN_THREADS], s_m[SCAN_THREADS];
  __shared__ uint32_t s_last;
  const uint32_t Lm1 = (1u << log_L) - 1;
  uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
  uint32_t a = 0, t = 0, m = 0;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_ITEMS; k++) {
    uint32_t v = (base + k < NB) ? cnt[base + k] : 0;
    a += v;
    t += (v + Lm1) >> log_L;
    m = max(m, v);
  }
  s_a[threadIdx.x] = a; s_t[threadIdx.x] = t; s_m[threadIdx.x] = m;
  __syncthreads();
  for (uint32_t s = SCAN_THREADS / 2; s >= 1; s >>= 1) {
    if (threadIdx.x < s) {
      s_a[threadIdx.x] += s_a[threadIdx.x + s];
      s_t[threadIdx.x] += s_t[threadIdx.x + s];
      s_m[threadIdx.x] = max(s_m[threadIdx.x], s_m[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    bsum[3 * blockIdx.x] = s_a[0];
    bsum[3 * blockIdx.x + 1] = s_t[0];
    bsum[3 * blockIdx.x + 2] = s_m[0];   // the largest count of the block
    __threadfence();                     // the sums above are visible device-wide before this workgroup counts as done
    s_last = atomicAdd(meta + SCAN_DONE, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // the last workgroup: exclusive scan of the nblk <= 1024 block sums (4 per thread), totals out
  const uint32_t nblk = gridDim.x, tid = threadIdx.x;
  volatile uint32_t* vb = bsum;          // written by other workgroups of this launch: not through a cached non-coherent load
  uint32_t va[4], vt[4], vm = 0;
  a = 0; t = 0;
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    const uint32_t b = 4 * tid + k;
    va[k] = b < nblk ? vb[3 * b] : 0u;
    vt[k] = b < nblk ? vb[3 * b + 1] : 0u;
    vm = max(vm, b < nblk ? vb[3 * b + 2] : 0u);
    a += va[k];
    t += vt[k];
  }
  __syncthreads();
  s_a[tid] = a; s_t[tid] = t; s_m[tid] = vm;
  __syncthreads();
  for (uint32_t d = 1; d < SCAN_THREADS; d <<= 1) {
    uint32_t xa = 0, xt = 0, xm = 0;
    if (tid >= d) { xa = s_a[tid - d]; xt = s_t[tid - d]; xm = s_m[tid - d]; }
    __syncthreads();
    s_a[tid] += xa; s_t[tid] += xt; s_m[tid] = max(s_m[tid], xm);
    __syncthreads();
  }
  uint32_t ra = s_a[tid] - a, rt = s_t[tid] - t;
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    const uint32_t b = 4 * tid + k;
    if (b < nblk) {
      bsum[3 * b] = ra;
      bsum[3 * b + 1] = rt;
    }
    ra += va[k];
    rt += vt[k];
  }
  if (tid == SCAN_THREADS - 1) {
    if (off) off[NB] = s_a[tid];
    toff[NB] = s_t[tid];
    meta[0] = s_a[tid];
    meta[1] = s_t[tid];
    meta[2] = s_m[tid];
    meta[SCAN_DONE] = 0;                 // ready for the next launch on this stream
    if (host_meta) {
      host_meta[0] = s_a[tid];
      host_meta[1] = s_t[tid];
      host_meta[2] = s_m[tid];
      __threadfence_system();
    }
  }
}
__global__ void __launch_bounds__(256) msm_scan_write(const uint32_t* __restrict__ cnt, uint32_t NB, uint32_t log_L,
                                                      const uint32_t* __restrict__ bsum, uint32_t* __restrict__ off,
                                                      uint32_t* __restrict__ ntask, uint32_t* __restrict__ toff) {
  side_kernel_prio();
  __shared__ uint32_t s_a[SCAN_THREADS], s_t[SCAN_THREADS];
  const uint32_t Lm1 = (1u << log_L) - 1, tid = threadIdx.x;
  uint32_t base = blockIdx.x * SCAN_BLOCK + tid * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t a = 0, t = 0;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_ITEMS; k++) {
    v[k] = (base + k < NB) ? cnt[base + k] : 0;
    a += v[k];
    t += (v[k] + Lm1) >> log_L;
  }
  s_a[tid] = a; s_t[tid] = t;
  __syncthreads();
  for (uint32_t d = 1; d < SCAN_THREADS; d <<= 1) {
    uint32_t va = 0, vt = 0;
    if (tid >= d) { va = s_a[tid - d]; vt = s_t[tid - d]; }
    __syncthreads();
    s_a[tid] += va; s_t[tid] += vt;
    __syncthreads();
  }
  uint32_t ra = bsum[3 * blockIdx.x] + s_a[tid] - a, rt = bsum[3 * blockIdx.x + 1] + s_t[tid] - t;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_ITEMS; k++) {
    if (base + k < NB) {
      uint32_t nt = (v[k] + Lm1) >> log_L;
      if (off) off[base + k] = ra;
      ntask[base + k] = nt;
      toff[base + k] = rt;
      ra += v[k];
      rt += nt;
    }
  }
}

// ------------------------------------------------------------------ 3b: LDS-cursor scatter
// workgroup (p, j): cursors = bucket offset + this chunk's prefix, kept in LDS; every point
// of the chunk takes the next slot of its bucket with an LDS atomic.
__global__ void __launch_bounds__(1024) msm_scatter(const int16_t* __restrict__ dig, uint32_t n, uint32_t chunk,
                                                    uint32_t nbw, const uint32_t* __restrict__ hist,
                                                    const uint32_t* __restrict__ off, uint32_t collapse_W,
                                                    uint32_t n_tab, uint32_t* __restrict__ sorted) {
  side_kernel_prio();
  extern __shared__ uint32_t s_cur[];
  const uint32_t j = blockIdx.x, p = blockIdx.y, P = gridDim.y;
  // blockIdx.z = round r of R: only the buckets [r, r+1) * nbw / R are placed.  Every round re-reads
  // the (2-byte, coalesced) digits, but the 4-byte scattered stores of the workgroups in flight stay
  // inside 1/R of the output, so sectors fill up in L2 before they are written back.
  const uint32_t span = nbw / gridDim.z, b_lo = blockIdx.z * span;
  const uint32_t* pre = hist + ((size_t)j * P + p) * nbw + b_lo;
  // fixed-base mode (collapse_W = windows per MSM): the W windows of an MSM share ONE bucket set and an
  // entry names row (window, i) of the precomputed table  2^offset_w * P_i
  const uint32_t* ob = off + (size_t)(collapse_W ? j / collapse_W : j) * nbw + b_lo;
  const uint32_t base_idx = collapse_W ? (j % collapse_W) * n_tab : 0u;
  for (uint32_t b = threadIdx.x; b < span; b += blockDim.x) s_cur[b] = ob[b] + pre[b];
  __syncthreads();
  const uint32_t lo = p * chunk, hi = min(n, lo + chunk);
  const int16_t* row = dig + (size_t)j * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int32_t d = row[i];
    uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u - b_lo;   // d == 0 wraps to a huge value
    if (b < span) {
      uint32_t pos = atomicAdd(&s_cur[b], 1u);
      sorted[pos] = (base_idx + i) | (d < 0 ? 0x80000000u : 0u);
    }
  }
}

// ------------------------------------------------------------------ 3c: two-pass sort
// msm_scatter's 4-byte stores land all over the output (a workgroup holds chunk / nbw entries per
// bucket), and every one of them costs a 32-byte write to HBM.  The two-pass sort only ever writes
// runs: pass 1 partitions the digits into B coarse bins per bucket set (bin = bucket >> shift) through
// an LDS tile sort, so that a wave stores consecutive addresses; pass 2 gives each coarse bin to
// one workgroup, which counts its 2^shift buckets in LDS, emits the bucket counts and writes the
// bin's entries in bucket order, again from LDS.
static constexpr uint32_t SORT_TILE = 8192;  // entries per LDS tile (both passes)

// pass 1: grid (rows, P); row j / chunk p as in msm_hist.  chist holds the per-chunk exclusive
// prefixes (msm_hist_prefix), coff the bin offsets.  Output: (entry, bucket) pairs grouped by bin.
__global__ void __launch_bounds__(1024) msm_partition(const int16_t* __restrict__ dig, uint32_t n, uint32_t chunk,
                                                      uint32_t B, uint32_t shift, const uint32_t* __restrict__ chist,
                                                      const uint32_t* __restrict__ coff, uint32_t collapse_W,
                                                      uint32_t n_tab, uint32_t* __restrict__ part_entry,
                                                      uint16_t* __restrict__ part_fine) {
  side_kernel_prio();
  extern __shared__ uint32_t s_mem[];
  uint32_t* s_cnt = s_mem;                 // [B] tile counts, then tile bases
  uint32_t* s_base = s_cnt + B;            // [B]
  uint32_t* s_gcur = s_base + B;           // [B] global cursors of this (row, chunk)
  uint32_t* s_entry = s_gcur + B;          // [SORT_TILE]
  uint16_t* s_fine = reinterpret_cast<uint16_t*>(s_entry + SORT_TILE);  // [SORT_TILE]
  const uint32_t j = blockIdx.x, p = blockIdx.y, P = gridDim.y, tid = threadIdx.x;
  const uint32_t set = collapse_W ? j / collapse_W : j;
  const uint32_t base_idx = collapse_W ? (j % collapse_W) * n_tab : 0u;
  for (uint32_t b = tid; b < B; b += blockDim.x) s_gcur[b] = coff[(size_t)set * B + b] + chist[((size_t)j * P + p) * B + b];
  const uint32_t lo = p * chunk, hi = min(n, lo + chunk);
  const int16_t* row = dig + (size_t)j * n;
  constexpr uint32_t PER = SORT_TILE / 1024;
  for (uint32_t t0 = lo; t0 < hi; t0 += SORT_TILE) {
    for (uint32_t b = tid; b < B; b += blockDim.x) s_cnt[b] = 0;
    __syncthreads();
    int32_t d[PER];
    uint32_t rank[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t i = t0 + k * 1024 + tid;
      d[k] = i < hi ? (int32_t)row[i] : 0;
      if (d[k]) rank[k] = atomicAdd(&s_cnt[((uint32_t)(d[k] < 0 ? -d[k] : d[k]) - 1u) >> shift], 1u);
    }
    __syncthreads();
    // exclusive scan of the B tile counts (B <= 1024): s_base
    {
      uint32_t v = tid < B ? s_cnt[tid] : 0;
      if (tid < B) s_base[tid] = v;
      __syncthreads();
      for (uint32_t dd = 1; dd < B; dd <<= 1) {
        uint32_t u = (tid < B && tid >= dd) ? s_base[tid - dd] : 0;
        __syncthreads();
        if (tid < B) s_base[tid] += u;
        __syncthreads();
      }
      if (tid < B) s_base[tid] -= v;
      __syncthreads();
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      if (d[k]) {
        const uint32_t fine = (uint32_t)(d[k] < 0 ? -d[k] : d[k]) - 1u;
        const uint32_t slot = s_base[fine >> shift] + rank[k];
        s_entry[slot] = (base_idx + t0 + k * 1024 + tid) | (d[k] < 0 ? 0x80000000u : 0u);
        s_fine[slot] = (uint16_t)fine;
      }
    }
    __syncthreads();
    const uint32_t total = s_base[B - 1] + s_cnt[B - 1];
    for (uint32_t slot = tid; slot < total; slot += blockDim.x) {
      const uint32_t fine = s_fine[slot], bin = fine >> shift;
      const uint32_t g = s_gcur[bin] + (slot - s_base[bin]);
      part_entry[g] = s_entry[slot];
      part_fine[g] = (uint16_t)fine;
    }
    __syncthreads();
    for (uint32_t b = tid; b < B; b += blockDim.x) s_gcur[b] += s_cnt[b];
    __syncthreads();
  }
}

// pass 2: one workgroup per (set, coarse bin): F = 2^shift buckets.  counts[set*nbw + bin*F + f] and
// the bin's slice of `sorted` in bucket order.
__global__ void __launch_bounds__(512) msm_fine_sort(const uint32_t* __restrict__ part_entry,
                                                     const uint16_t* __restrict__ part_fine,
                                                     const uint32_t* __restrict__ coff,
                                                     const uint32_t* __restrict__ ccnt, uint32_t B, uint32_t shift,
                                                     uint32_t nbw, uint32_t* __restrict__ counts,
                                                     uint32_t* __restrict__ sorted) {
  side_kernel_prio();
  extern __shared__ uint32_t s_mem[];
  const uint32_t F = 1u << shift, tid = threadIdx.x, nthr = blockDim.x;
  uint32_t* s_cnt = s_mem;        // [F] counts, then cursors
  uint32_t* s_ofs = s_cnt + F;    // [F] exclusive offsets
  uint32_t* s_part = s_ofs + F;   // [nthr] scan scratch
  uint32_t* s_out = s_part + nthr;  // [SORT_TILE]
  const uint32_t start = coff[blockIdx.x], E = ccnt[blockIdx.x];
  const uint32_t set = blockIdx.x / B, bin = blockIdx.x - set * B;
  for (uint32_t f = tid; f < F; f += nthr) s_cnt[f] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < E; i += nthr) atomicAdd(&s_cnt[part_fine[start + i] & (F - 1)], 1u);
  __syncthreads();
  // exclusive scan over F: contiguous share per thread + Hillis-Steele over the shares
  const uint32_t per = (F + nthr - 1) / nthr, f0 = min(tid * per, F), f1 = min(f0 + per, F);
  uint32_t a = 0;
  for (uint32_t f = f0; f < f1; f++) a += s_cnt[f];
  s_part[tid] = a;
  __syncthreads();
  for (uint32_t dd = 1; dd < nthr; dd <<= 1) {
    uint32_t u = tid >= dd ? s_part[tid - dd] : 0;
    __syncthreads();
    s_part[tid] += u;
    __syncthreads();
  }
  uint32_t run = s_part[tid] - a;
  uint32_t* cout = counts + (size_t)set * nbw + (size_t)bin * F;
  for (uint32_t f = f0; f < f1; f++) {
    const uint32_t c = s_cnt[f];
    cout[f] = c;
    s_ofs[f] = run;
    run += c;
  }
  __syncthreads();
  for (uint32_t f = tid; f < F; f += nthr) s_cnt[f] = s_ofs[f];   // cursors
  __syncthreads();
  if (E <= SORT_TILE) {
    for (uint32_t i = tid; i < E; i += nthr) {
      const uint32_t pos = atomicAdd(&s_cnt[part_fine[start + i] & (F - 1)], 1u);
      s_out[pos] = part_entry[start + i];
    }
    __syncthreads();
    for (uint32_t i = tid; i < E; i += nthr) sorted[start + i] = s_out[i];
  } else {  // oversized bin (skewed scalars): place directly; the region belongs to this workgroup alone
    for (uint32_t i = tid; i < E; i += nthr) {
      const uint32_t pos = atomicAdd(&s_cnt[part_fine[start + i] & (F - 1)], 1u);
      sorted[start + pos] = part_entry[start + i];
    }
  }
}

// Round 5: pass 2 that also closes the front end.  A workgroup owns a coarse bin whose entry offset (coff) and task-slot base
// (tbase) are known from the coarse scan, so everything the bucket-level scans produced is local to it: off[b] = start +
// exclusive count prefix, ntask[b] = ceil(c / L), toff[b] = tbase + exclusive task prefix (the task index space has gaps at
// the end of every bin: consumers only ever index partial[toff[b] + seg]).  What is global -- the number of tasks, the
// largest bucket, the histogram of task lengths that orders the tasks longest first -- leaves the workgroup as fire-and-forget
// atomics into one of FE_REPL replicas (thousands of workgroups adding to ONE word per length cost a millisecond of serialised
// atomics) and is summed by the NEXT kernel of the job (msm_task_scatter_reserve), where the kernel boundary has made it
// complete: no workgroup waits for a return value or for another workgroup, and there is no release fence -- in this kernel a
// fence writes back an L2 that has just been filled with sorted entries, once per workgroup: a millisecond
// (profiles/r05_sweeps/frontend.txt has all three measurements).  msm_scan_sums, msm_scan_write, msm_task_hist and msm_task_scan
// are gone from the job's chain.
//   fe words: [3] finished workgroups of msm_hist_prefix, [FE_CURSOR + k] the scatter's running position inside the tasks of
//   length k (zero at its start), and two SETS of replicas, used by alternate jobs of the engine (a job's sort zeroes the other
//   set for the next job): per replica r of set p at FE_SET + p * FE_SET_WORDS + r * FE_ROW: [0] tasks, [1] largest count,
//   [2 + k] tasks of (clamped) length k
static constexpr uint32_t TASK_BINS_FE = 257;     // = TASK_BINS (defined with the task ordering below)
static constexpr uint32_t FE_REPL = 64;
static constexpr uint32_t FE_CURSOR = 8, FE_ROW = TASK_BINS_FE + 2, FE_SET_WORDS = FE_REPL * FE_ROW, FE_SET = FE_CURSOR + TASK_BINS_FE,
                          FE_WORDS = FE_SET + 2 * FE_SET_WORDS;
struct FrontEndOut {
  uint32_t* off;
  uint32_t* ntask;
  uint32_t* toff;
  const uint32_t* tbase;
  uint32_t* fe;
  uint32_t log_L, parity;
};
__global__ void __launch_bounds__(512) msm_fine_sort_fused(const uint32_t* __restrict__ part_entry,
                                                           const uint16_t* __restrict__ part_fine,
                                                           const uint32_t* __restrict__ coff,
                                                           const uint32_t* __restrict__ ccnt, uint32_t B, uint32_t shift,
                                                           uint32_t nbw, uint32_t* __restrict__ counts,
                                                           uint32_t* __restrict__ sorted, FrontEndOut o) {
  side_kernel_prio();
  extern __shared__ uint32_t s_mem[];
  __shared__ uint32_t s_th[TASK_BINS_FE];
  __shared__ uint32_t s_red[2];   // tasks of the bin, largest count
  const uint32_t F = 1u << shift, tid = threadIdx.x, nthr = blockDim.x;
  uint32_t* s_cnt = s_mem;        // [F] counts, then cursors
  uint32_t* s_ofs = s_cnt + F;    // [F] exclusive offsets
  uint32_t* s_part = s_ofs + F;   // [nthr] scan scratch
  uint32_t* s_out = s_part + nthr;  // [SORT_TILE]
  const uint32_t start = coff[blockIdx.x], E = ccnt[blockIdx.x];
  const uint32_t set = blockIdx.x / B, bin = blockIdx.x - set * B;
  const uint32_t Lm1 = (1u << o.log_L) - 1, full_bin = min(1u << o.log_L, TASK_BINS_FE - 1);
  // housekeeping for the jobs to come: the other replica set and the scatter's cursors back to zero (whoever used them last
  // has finished: same stream)
  {
    uint32_t* other = o.fe + FE_SET + (o.parity ^ 1u) * FE_SET_WORDS;
    for (uint32_t row = blockIdx.x; row < FE_REPL; row += gridDim.x)
      for (uint32_t k = tid; k < FE_ROW; k += nthr) other[row * FE_ROW + k] = 0;
    if (blockIdx.x == 0)
      for (uint32_t k = tid; k < TASK_BINS_FE; k += nthr) o.fe[FE_CURSOR + k] = 0;
  }
  for (uint32_t f = tid; f < F; f += nthr) s_cnt[f] = 0;
  for (uint32_t k = tid; k < TASK_BINS_FE; k += nthr) s_th[k] = 0;
  if (tid < 2) s_red[tid] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < E; i += nthr) atomicAdd(&s_cnt[part_fine[start + i] & (F - 1)], 1u);
  __syncthreads();
  // exclusive scans over F (entries and tasks): contiguous share per thread + ONE Hillis-Steele over the shares, the two sums
  // packed in a 64-bit word
  const uint32_t per = (F + nthr - 1) / nthr, f0 = min(tid * per, F), f1 = min(f0 + per, F);
  uint32_t a = 0, t = 0, mx = 0;
  for (uint32_t f = f0; f < f1; f++) {
    const uint32_t c = s_cnt[f];
    a += c;
    t += (c + Lm1) >> o.log_L;
    mx = max(mx, c);
    // task lengths: nfull tasks of exactly L entries and at most one shorter one
    const uint32_t nfull = c >> o.log_L, rem = c - (nfull << o.log_L);
    if (nfull) atomicAdd(&s_th[full_bin], nfull);
    if (rem) atomicAdd(&s_th[min(rem, TASK_BINS_FE - 1)], 1u);
  }
  uint64_t* s_part64 = reinterpret_cast<uint64_t*>(s_part);   // [nthr] 64-bit: the scratch area is followed by s_out, which is not in use yet
  s_part64[tid] = ((uint64_t)t << 32) | a;
  __syncthreads();
  for (uint32_t dd = 1; dd < nthr; dd <<= 1) {
    uint64_t u = tid >= dd ? s_part64[tid - dd] : 0;
    __syncthreads();
    s_part64[tid] += u;
    __syncthreads();
  }
  uint32_t run = (uint32_t)s_part64[tid] - a;
  uint32_t trun = o.tbase[blockIdx.x] + (uint32_t)(s_part64[tid] >> 32) - t;
  if (t) atomicAdd(&s_red[0], t);
  if (mx) atomicMax(&s_red[1], mx);
  const size_t b_first = (size_t)set * nbw + (size_t)bin * F;
  uint32_t* cout = counts + b_first;
  for (uint32_t f = f0; f < f1; f++) {
    const uint32_t c = s_cnt[f], nt = (c + Lm1) >> o.log_L;
    cout[f] = c;
    s_ofs[f] = run;
    o.off[b_first + f] = start + run;
    o.ntask[b_first + f] = nt;
    o.toff[b_first + f] = trun;
    run += c;
    trun += nt;
  }
  __syncthreads();
  // the bin's share of the global figures: fire and forget
  {
    uint32_t* mine = o.fe + FE_SET + o.parity * FE_SET_WORDS + (blockIdx.x % FE_REPL) * FE_ROW;
    for (uint32_t k = tid; k <= full_bin; k += nthr)
      if (s_th[k]) atomicAdd(mine + 2 + k, s_th[k]);
    if (tid == 0) {
      if (s_red[0]) atomicAdd(mine, s_red[0]);
      if (s_red[1]) atomicMax(mine + 1, s_red[1]);
    }
  }
  for (uint32_t f = tid; f < F; f += nthr) s_cnt[f] = s_ofs[f];   // cursors
  __syncthreads();
  if (E <= SORT_TILE) {
    for (uint32_t i = tid; i < E; i += nthr) {
      const uint32_t pos = atomicAdd(&s_cnt[part_fine[start + i] & (F - 1)], 1u);
      s_out[pos] = part_entry[start + i];
    }
    __syncthreads();
    for (uint32_t i = tid; i < E; i += nthr) sorted[start + i] = s_out[i];
  } else {  // oversized bin (skewed scalars): place directly; the region belongs to this workgroup alone
    for (uint32_t i = tid; i < E; i += nthr) {
      const uint32_t pos = atomicAdd(&s_cnt[part_fine[start + i] & (F - 1)], 1u);
      sorted[start + pos] = part_entry[start + i];
    }
  }
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

// ---- task ordering: longest tasks first, equal lengths side by side, so the 64 lanes of a
// wave run the same number of additions (bucket sizes are Poisson-distributed: without this
// a wave waits for its largest bucket, ~30 % of the lanes' time idle).
static constexpr uint32_t ACC_TICKET = 8;          // word of meta_ that holds msm_accumulate's task counter
static constexpr uint32_t TASK_BINS = 257;        // task length clamped to 256
static_assert(TASK_BINS == TASK_BINS_FE, "one histogram of task lengths");
// buckets per workgroup in the ordering passes: ~128 workgroups, 256 .. 8192 buckets each
static inline uint32_t task_block_for(uint32_t NB, uint32_t nbins) {
  const uint32_t max_blk = std::min<uint32_t>(128, (32 * 1024) / nbins);  // msm_task_scan: nbins * nblk <= 32 Ki
  uint32_t tb = 256;
  while ((NB + tb - 1) / tb > max_blk) tb <<= 1;
  return tb;
}

// thist[bin * nblk + blk] = number of tasks of (clamped) length `bin` in block blk
__global__ void __launch_bounds__(256) msm_task_hist(const uint32_t* __restrict__ cnt, uint32_t NB,
                                                     uint32_t log_L, uint32_t task_block,
                                                     uint32_t* __restrict__ thist) {
  side_kernel_prio();
  __shared__ uint32_t s_h[TASK_BINS];
  for (uint32_t k = threadIdx.x; k < TASK_BINS; k += blockDim.x) s_h[k] = 0;
  __syncthreads();
  const uint32_t full_bin = min(1u << log_L, TASK_BINS - 1);
  for (uint32_t q = threadIdx.x; q < task_block; q += blockDim.x) {
    uint32_t b = blockIdx.x * task_block + q;
    if (b < NB) {
      // a bucket is nfull tasks of exactly L entries plus at most one shorter task
      uint32_t cv = cnt[b], nfull = cv >> log_L, rem = cv - (nfull << log_L);
      if (nfull) atomicAdd(&s_h[full_bin], nfull);
      if (rem) atomicAdd(&s_h[min(rem, TASK_BINS - 1)], 1u);
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < TASK_BINS; k += blockDim.x) thist[k * gridDim.x + blockIdx.x] = s_h[k];
}
// one workgroup: exclusive scan of thist in DESCENDING bin order (bin-major, block-minor) over the nbins = L + 1 bins in
// use; total = nbins * nblk <= 32 Ki entries.  256 lanes with a handful of registers, two passes over the entries (the
// second one hits L2): the former 1024 lanes x 98 registers needed a whole CU's register file at once and, beside a
// running accumulation, waited for that accumulation to END -- a millisecond on the path of the next job.
__global__ void __launch_bounds__(256) msm_task_scan(uint32_t* __restrict__ thist, uint32_t nblk, uint32_t nbins,
                                                     uint32_t* __restrict__ ticket) {
  side_kernel_prio();
  __shared__ uint32_t s_sum[256];
  if (threadIdx.x == 0) *ticket = 0;   // msm_accumulate's task counter (next launch on this stream)
  const uint32_t total = nbins * nblk, tid = threadIdx.x;
  const uint32_t per = (total + 255) / 256;
  const uint32_t lo = min(tid * per, total), hi = min(lo + per, total);
  // position q in scan order <-> entry (nbins-1 - q / nblk) * nblk + q % nblk
  const uint32_t row0 = lo / nblk, col0 = lo - row0 * nblk;
  uint32_t a = 0;
  {
    uint32_t row = row0, col = col0;
#pragma unroll 8
    for (uint32_t q = lo; q < hi; q++) {
      a += thist[(nbins - 1 - row) * nblk + col];
      if (++col == nblk) { col = 0; row++; }
    }
  }
  s_sum[tid] = a;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    uint32_t u = tid >= d ? s_sum[tid - d] : 0;
    __syncthreads();
    s_sum[tid] += u;
    __syncthreads();
  }
  uint32_t run = s_sum[tid] - a;
  {
    uint32_t row = row0, col = col0;
    for (uint32_t q = lo; q < hi; q++) {
      const uint32_t idx = (nbins - 1 - row) * nblk + col;
      const uint32_t v = thist[idx];
      thist[idx] = run;
      run += v;
      if (++col == nblk) { col = 0; row++; }
    }
  }
}
// order[pos] = (bucket, segment) of the task that runs as thread `pos`
__global__ void __launch_bounds__(256) msm_task_scatter(const uint32_t* __restrict__ cnt, uint32_t NB,
                                                        uint32_t log_L, uint32_t task_block,
                                                        const uint32_t* __restrict__ thist,
                                                        uint2* __restrict__ order) {
  side_kernel_prio();
  __shared__ uint32_t s_c[TASK_BINS];
  for (uint32_t k = threadIdx.x; k < TASK_BINS; k += blockDim.x) s_c[k] = thist[k * gridDim.x + blockIdx.x];
  __syncthreads();
  const uint32_t full_bin = min(1u << log_L, TASK_BINS - 1);
  for (uint32_t q = threadIdx.x; q < task_block; q += blockDim.x) {
    uint32_t b = blockIdx.x * task_block + q;
    if (b < NB) {
      uint32_t cv = cnt[b], nfull = cv >> log_L, rem = cv - (nfull << log_L);
      if (nfull) {
        uint32_t pos = atomicAdd(&s_c[full_bin], nfull);
        for (uint32_t seg = 0; seg < nfull; seg++) order[pos + seg] = make_uint2(b, seg);
      }
      if (rem) {
        uint32_t pos = atomicAdd(&s_c[min(rem, TASK_BINS - 1)], 1u);
        order[pos] = make_uint2(b, nfull);
      }
    }
  }
}

// the same after msm_fine_sort_fused: every workgroup first sums the replicas of the task-length histogram the sort left (the
// kernel boundary made them complete) into the positions at which each length starts in `order` (descending lengths), counts
// its own tasks per length, reserves their positions with one atomic per length on the running cursors, and places them --
// msm_task_hist and msm_task_scan are not needed.  Workgroup 0 also hands the job's totals to the accumulation and to the host.
// (Tasks of one length come in the order the reservations happen to be served: which lane runs which task is free, the sums
// are the same.)
struct FrontEndTotals {
  uint32_t* fe;
  const uint32_t* coff;           // [NBc]: entries of the job
  const uint32_t* tbase;          // [NBc]: task slots of the job
  uint32_t* off;                  // off[NB] <- entries
  uint32_t* toff;                 // toff[NB] <- task slots
  uint32_t* meta;                 // [0] entries, [1] tasks, [2] largest count; [ticket_word] <- 0
  volatile uint32_t* host_meta;   // the same three for the host (mapped page-locked memory)
  uint32_t NBc, parity, ticket_word;
};
__global__ void __launch_bounds__(256) msm_task_scatter_reserve(const uint32_t* __restrict__ cnt, uint32_t NB,
                                                                uint32_t log_L, uint32_t task_block, FrontEndTotals ft,
                                                                uint2* __restrict__ order) {
  side_kernel_prio();
  __shared__ uint32_t s_c[TASK_BINS], s_tot[TASK_BINS], s_red[2];
  const uint32_t tid = threadIdx.x;
  const uint32_t full_bin = min(1u << log_L, TASK_BINS - 1), nbins = full_bin + 1;
  const uint32_t* set = ft.fe + FE_SET + ft.parity * FE_SET_WORDS;
  for (uint32_t k = tid; k < TASK_BINS; k += blockDim.x) { s_c[k] = 0; s_tot[k] = 0; }
  if (tid < 2) s_red[tid] = 0;
  __syncthreads();
  for (uint32_t idx = tid; idx < nbins * FE_REPL; idx += blockDim.x) {
    const uint32_t k = idx % nbins, r = idx / nbins;
    const uint32_t v = set[r * FE_ROW + 2 + k];
    if (v) atomicAdd(&s_tot[k], v);
  }
  if (blockIdx.x == 0 && tid < FE_REPL) {
    const uint32_t tk = set[tid * FE_ROW], mxr = set[tid * FE_ROW + 1];
    if (tk) atomicAdd(&s_red[0], tk);
    if (mxr) atomicMax(&s_red[1], mxr);
  }
  for (uint32_t q = tid; q < task_block; q += blockDim.x) {
    uint32_t b = blockIdx.x * task_block + q;
    if (b < NB) {
      uint32_t cv = cnt[b], nfull = cv >> log_L, rem = cv - (nfull << log_L);
      if (nfull) atomicAdd(&s_c[full_bin], nfull);
      if (rem) atomicAdd(&s_c[min(rem, TASK_BINS - 1)], 1u);
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) {
    const uint32_t entries = ft.coff[ft.NBc], tasks = s_red[0], largest = s_red[1];
    ft.off[NB] = entries;
    ft.toff[NB] = ft.tbase[ft.NBc];
    ft.meta[0] = entries;
    ft.meta[1] = tasks;
    ft.meta[2] = largest;
    ft.meta[ft.ticket_word] = 0;          // msm_accumulate's task counter (the next launch on this stream)
    if (ft.host_meta) {
      ft.host_meta[0] = entries;
      ft.host_meta[1] = tasks;
      ft.host_meta[2] = largest;
      __threadfence_system();
    }
  }
  // position of this workgroup's tasks of length k: every longer task first, then what other workgroups reserved before
  for (uint32_t k = tid; k < nbins; k += blockDim.x) {
    uint32_t base = 0;
    for (uint32_t k2 = k + 1; k2 < nbins; k2++) base += s_tot[k2];
    const uint32_t mine = s_c[k];
    s_c[k] = base + (mine ? atomicAdd(ft.fe + FE_CURSOR + k, mine) : 0u);
  }
  __syncthreads();
  for (uint32_t q = tid; q < task_block; q += blockDim.x) {
    uint32_t b = blockIdx.x * task_block + q;
    if (b < NB) {
      uint32_t cv = cnt[b], nfull = cv >> log_L, rem = cv - (nfull << log_L);
      if (nfull) {
        uint32_t pos = atomicAdd(&s_c[full_bin], nfull);
        for (uint32_t seg = 0; seg < nfull; seg++) order[pos + seg] = make_uint2(b, seg);
      }
      if (rem) {
        uint32_t pos = atomicAdd(&s_c[min(rem, TASK_BINS - 1)], 1u);
        order[pos] = make_uint2(b, nfull);
      }
    }
  }
}

// Persistent: the grid is a fixed number of waves per SIMD (MsmConfig::acc_waves; three fill the register file at 161
// registers), and every wave takes tickets of 64 consecutive tasks from a counter until the task list (longest first, so
// the 64 lanes of a ticket have equal work) is used up: the end of the launch is balanced by construction instead of by
// the order in which the hardware happens to retire workgroups (2^20: 1.16 -> 1.12 ms; profiles/r03_sweeps/persistent_accumulate.txt).
// Sizing the launch to leave room for the kernels of other streams (two waves per SIMD) was measured too and does not pay:
// the kernels beside it still crawl (same file), and the accumulation alone loses 4 %.
__global__ void __launch_bounds__(256) msm_accumulate(const uint32_t* __restrict__ sorted, BatchPtrs bp,
                                                      uint32_t buckets_per_msm,
                                                      const uint32_t* __restrict__ off,
                                                      const uint32_t* __restrict__ cnt,
                                                      const uint32_t* __restrict__ toff,
                                                      const uint2* __restrict__ order, uint32_t log_L,
                                                      const uint32_t* __restrict__ meta, uint32_t* __restrict__ ticket,
                                                      xyzz29_mem* __restrict__ partial, uint64_t* __restrict__ trace) {
  // launched before the host has read the counters back (the read overlaps this kernel); the exact task count is meta[1]
  const uint32_t ntasks = meta[1], lane = threadIdx.x & 63u;
  // debug (msm.acc_trace): when every wave starts and leaves (wall_clock64 ticks)
  const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (trace && lane == 0) trace[2 * wave_id] = wall_clock64();
  for (;;) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(ticket, 64u);
    base = __builtin_amdgcn_readfirstlane(base);
    if (base >= ntasks) break;          // every wave gets here: the counter only grows
    const uint32_t t = base + lane;
    if (t < ntasks) {
      const uint2 o = order[t];
      const uint32_t b = o.x, seg = o.y;
      const g1_affine_mem* __restrict__ bases = bp.bases[b / buckets_per_msm];
      uint32_t start = off[b] + (seg << log_L);
      uint32_t end = min(off[b] + cnt[b], start + (1u << log_L));
      xyzz29 acc = xyzz29_identity();
      uint32_t e = sorted[start];
      g1_affine_mem raw = bases[e & 0x7fffffffu];
      for (uint32_t k = start; k < end; k++) {
        uint32_t e_next = 0;
        g1_affine_mem raw_next = raw;
        if (k + 1 < end) {  // prefetch the next point while this one is being added
          e_next = sorted[k + 1];
          raw_next = bases[e_next & 0x7fffffffu];
        }
        affine29 p = affine29_load(&raw);
        if (e >> 31) affine29_negate(p);
        xyzz29_madd(acc, p);
        e = e_next;
        raw = raw_next;
      }
      xyzz29_store(partial + toff[b] + seg, acc);
    }
  }
  if (trace && lane == 0) trace[2 * wave_id + 1] = wall_clock64();
}

// Every kernel below is written for LOGICAL threads of Q lanes: Q = 1 is one lane per point
// operation, Q = 4 the quad-cooperative addition (xyzz29_add_quad; all 4 lanes hold the same
// values).  lt = logical thread, role = lane within the quad.
template <int Q>
__device__ __forceinline__ void add_q(xyzz29& acc, const xyzz29& q, uint32_t role) {
  if (Q == 4) xyzz29_add_quad(acc, q, role);
  else xyzz29_add(acc, q);   // (inlined at every call site: one out-of-line copy per kernel was measured and is slower, docs/history.md section 4.11)
}
template <int Q>
__global__ void __launch_bounds__(256) msm_merge(const xyzz29_mem* __restrict__ in, const uint32_t* __restrict__ off,
                                                 const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ toff,
                                                 uint32_t NB, uint32_t log_L, const uint32_t* __restrict__ meta,
                                                 xyzz29_mem* __restrict__ out) {
  side_kernel_prio();
  // the grid covers a host-side upper bound; the exact task count of this level is meta[1]
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / Q, role = threadIdx.x % Q;
  if (t >= meta[1]) return;
  uint32_t b = find_owner(toff, NB, t);
  uint32_t seg = t - toff[b];
  uint32_t start = off[b] + (seg << log_L);
  uint32_t end = min(off[b] + cnt[b], start + (1u << log_L));
  xyzz29 acc = xyzz29_identity();
  for (uint32_t k = start; k < end; k++) add_q<Q>(acc, xyzz29_load(in + k), role);
  if (role == 0) xyzz29_store(out + t, acc);
}

// ------------------------------------------------------------------ 5: bucket reduction
// Window sum = sum_b (b+1) * B_b.  No doublings run on the GPU: every level only adds, and
// the power-of-two weights are applied by the host tail, where a dependent chain of point
// doublings costs a fraction of a microsecond per step instead of several on one GPU lane.
//
// level 0: thread t of window j owns G = 2^log_G consecutive buckets: run_t = sum B,
//          acc_t = sum (k+1) B_{first+k} (running sums).  A workgroup w of N threads emits
//             A_w = sum_t acc_t,  S_w = sum_t t * run_t (= sum_{t>=1} Suf_t),  R_w = sum_t run_t
//          so that  window = sum_w [ A_w + G * S_w + G * N * w * R_w ].
// level 1: one workgroup per window:  A = sum A_w,  S = sum S_w,  T = sum_w w * R_w.
// host:    window = A + 2^log_G * (S + 2^log_N * T).
struct ReduceOut {
  xyzz29_mem* a;
  xyzz29_mem* s;
  xyzz29_mem* r;
};
// tree-sum of arr[0..len) (len a power of two) by logical threads li = 0..len/2-1 of a group;
// every thread of the workgroup must call it (barriers inside)
template <int Q>
__device__ __forceinline__ void tree_sum(xyzz29_mem* arr, uint32_t len, uint32_t li, uint32_t role, bool member) {
  for (uint32_t s = len >> 1; s >= 1; s >>= 1) {
    if (member && li < s) {
      xyzz29 a = xyzz29_load(&arr[li]);
      add_q<Q>(a, xyzz29_load(&arr[li + s]), role);
      if (role == 0) xyzz29_store(&arr[li], a);
    }
    __syncthreads();
  }
}
// in-place suffix scan of arr[0..N) by N logical threads (Hillis-Steele)
template <int Q>
__device__ __forceinline__ xyzz29 suffix_scan(xyzz29_mem* arr, xyzz29 mine, uint32_t N, uint32_t lt, uint32_t role) {
  if (role == 0) xyzz29_store(&arr[lt], mine);
  __syncthreads();
  for (uint32_t d = 1; d < N; d <<= 1) {
    xyzz29 other = xyzz29_identity();
    if (lt + d < N) other = xyzz29_load(&arr[lt + d]);
    __syncthreads();
    add_q<Q>(mine, other, role);
    if (role == 0) xyzz29_store(&arr[lt], mine);
    __syncthreads();
  }
  return mine;
}

// level 0: grid (blocks, W), N = blockDim.x / Q logical threads (power of two >= 16)
template <int Q>
__device__ __forceinline__ void reduce_buckets_body(const xyzz29_mem* __restrict__ partial, const uint32_t* __restrict__ toff,
                                                    const uint32_t* __restrict__ ntask, uint32_t nbw, uint32_t log_G,
                                                    ReduceOut out) {
  side_kernel_prio();
  extern __shared__ uint4 smem[];
  const uint32_t N = blockDim.x / Q, lt = threadIdx.x / Q, role = threadIdx.x % Q;
  xyzz29_mem* sA = reinterpret_cast<xyzz29_mem*>(smem);
  xyzz29_mem* sR = sA + N;
  const uint32_t chunk = blockIdx.x * N + lt;
  const uint32_t G = 1u << log_G;
  xyzz29 acc = xyzz29_identity(), run = xyzz29_identity();
  const uint32_t first = chunk << log_G;
  if (first < nbw) {
    const uint32_t wbase = blockIdx.y * nbw;
    auto fetch = [&](uint32_t k) {
      const uint32_t b = first + k;
      xyzz29 v = xyzz29_identity();   // (not "c ? load : identity": two temporaries behind a pointer phi stay in scratch)
      if (b < nbw && ntask[wbase + b]) v = xyzz29_load(partial + toff[wbase + b]);
      return v;
    };
    xyzz29 nxt = fetch(G - 1);
    for (uint32_t k = G; k-- > 0;) {
      const xyzz29 cur = nxt;
      if (k) nxt = fetch(k - 1);  // in flight while the two additions below run
      add_q<Q>(run, cur, role);
      add_q<Q>(acc, run, role);
    }
  }
  const uint32_t o = blockIdx.y * gridDim.x + blockIdx.x;
  if (role == 0) xyzz29_store(&sA[lt], acc);
  suffix_scan<Q>(sR, run, N, lt, role);            // sR[t] = Suf_t
  if (threadIdx.x == 0) {
    xyzz29_store(out.r + o, xyzz29_load(&sR[0]));
    xyzz29_store(&sR[0], xyzz29_identity());  // S sums t >= 1 only
  }
  __syncthreads();
  // two tree sums side by side: lower half of the logical threads folds sA, upper half sR
  const uint32_t halfN = N >> 1;
  xyzz29_mem* arr = (lt < halfN) ? sA : sR;
  tree_sum<Q>(arr, N, (lt < halfN) ? lt : lt - halfN, role, true);
  if (threadIdx.x == 0) xyzz29_store(out.a + o, xyzz29_load(&sA[0]));
  if (lt == halfN && role == 0) xyzz29_store(out.s + o, xyzz29_load(&sR[0]));
}
template <int Q>
__global__ void __launch_bounds__(256) msm_reduce_buckets(const xyzz29_mem* __restrict__ partial,
                                                          const uint32_t* __restrict__ toff,
                                                          const uint32_t* __restrict__ ntask, uint32_t nbw,
                                                          uint32_t log_G, ReduceOut out) {
  reduce_buckets_body<Q>(partial, toff, ntask, nbw, log_G, out);
}
// the same within 168 registers (60 values live in scratch, +5 %): a wave of it fits beside two waves of msm_accumulate on
// a SIMD, so the reduction of one job runs under the accumulation of the next instead of after it
template <int Q>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
msm_reduce_buckets_lean(const xyzz29_mem* __restrict__ partial, const uint32_t* __restrict__ toff,
                        const uint32_t* __restrict__ ntask, uint32_t nbw, uint32_t log_G, ReduceOut out) {
  reduce_buckets_body<Q>(partial, toff, ntask, nbw, log_G, out);
}
// level 1: grid (1, W), blockDim = 3 * T1 * Q (T1 a power of two >= count): group 0 scans/folds
// the R items, group 1 folds A, group 2 folds S
template <int Q>
__global__ void __launch_bounds__(768) msm_reduce_items(ReduceOut in, uint32_t count, uint32_t T1, ReduceOut out) {
  side_kernel_prio();
  extern __shared__ uint4 smem[];
  xyzz29_mem* sR = reinterpret_cast<xyzz29_mem*>(smem);
  xyzz29_mem* sA = sR + T1;
  xyzz29_mem* sS = sA + T1;
  const uint32_t lt = threadIdx.x / Q, role = threadIdx.x % Q;
  const uint32_t g = lt / T1, li = lt - g * T1;
  const uint32_t base = blockIdx.y * count;
  xyzz29 v = xyzz29_identity();
  if (li < count) v = xyzz29_load((g == 0 ? in.r : g == 1 ? in.a : in.s) + base + li);
  if (role == 0) {
    if (g == 1) xyzz29_store(&sA[li], v);
    if (g == 2) xyzz29_store(&sS[li], v);
    // suffix scan of R (group 0 works, everyone keeps the barriers)
    if (g == 0) xyzz29_store(&sR[li], v);
  }
  __syncthreads();
  for (uint32_t d = 1; d < T1; d <<= 1) {
    xyzz29 other = xyzz29_identity();
    if (g == 0 && li + d < T1) other = xyzz29_load(&sR[li + d]);
    __syncthreads();
    if (g == 0) {
      add_q<Q>(v, other, role);
      if (role == 0) xyzz29_store(&sR[li], v);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) xyzz29_store(&sR[0], xyzz29_identity());  // T sums w >= 1 only
  __syncthreads();
  xyzz29_mem* arr = g == 0 ? sR : g == 1 ? sA : sS;
  tree_sum<Q>(arr, T1, li, role, true);
  if (li == 0 && role == 0) xyzz29_store((g == 0 ? out.r : g == 1 ? out.a : out.s) + blockIdx.y, xyzz29_load(&arr[0]));
}
// ------------------------------------------------------------------ 5b: 2-D bucket reduction (few bucket sets)
// The scan-based reduction above is a chain of ~40 dependent additions whatever the size.  With the bucket index
// split as b = hi * C + lo (C = 2^ceil(bits/2) columns, Rr = nbw / C rows),
//     sum_b (b + 1) B_b = C * sum_hi hi * R_hi + sum_lo lo * C_lo + U,   R_hi / C_lo row / column sums, U the total,
// and each small weighted sum by bits,  sum_x x * V_x = sum_j 2^j * (sum of the V_x with bit j of x set),
// everything on the device is a PLAIN sum: lines (rows and columns) first, then one masked sum per bit --
// two launches of ~9 dependent quad-cooperative additions each.  The powers of two are applied by the host tail,
// which already places terms at bit offsets; it receives bits + 1 points per bucket set, so this path is for
// jobs with few sets (fixed-base commits of up to 4 polynomials).
struct Reduce2dShape {
  uint32_t log_cols, log_rows;  // nbw = 2^(log_rows + log_cols)
};
// sum of up to 256 XYZZ values by one workgroup of 256 lanes: Q = 4: 64 quads (quad-cooperative additions, for
// few sets: latency), Q = 1: 256 lanes, one addition each (many sets: throughput); `get(e)` yields element e
template <int Q, typename F>
__device__ __forceinline__ xyzz29 wg_sum(uint32_t n, F get, xyzz29_mem* lds) {
  constexpr uint32_t NL = 256 / Q;  // logical threads
  const uint32_t lt = threadIdx.x / Q, role = threadIdx.x % Q;
  xyzz29 acc = xyzz29_identity();
  // (round 5: loading the next element while the addition runs was measured and is slower -- 366 -> 406 us of line sums per proof:
  // 185 registers instead of 149, and the loads of a quad's two to four elements are back to back anyway)
  for (uint32_t e = lt; e < n; e += NL) add_q<Q>(acc, get(e), role);
  if (role == 0) xyzz29_store(&lds[lt], acc);
  __syncthreads();
  tree_sum<Q>(lds, NL, lt, role, true);
  return xyzz29_load(&lds[0]);
}
// the same with `get(e, acc, role)` adding element e -- possibly several terms -- into the running sum itself
template <int Q, typename F>
__device__ __forceinline__ xyzz29 wg_sum_into(uint32_t n, F get, xyzz29_mem* lds) {
  constexpr uint32_t NL = 256 / Q;  // logical threads
  const uint32_t lt = threadIdx.x / Q, role = threadIdx.x % Q;
  xyzz29 acc = xyzz29_identity();
  for (uint32_t e = lt; e < n; e += NL) get(e, acc, role);
  if (role == 0) xyzz29_store(&lds[lt], acc);
  __syncthreads();
  tree_sum<Q>(lds, NL, lt, role, true);
  return xyzz29_load(&lds[0]);
}
// A bucket of a fixed-base job owns several partial sums (one per accumulation task: ~64 entries in tasks of 16).  The
// line sums below used to add them on the way -- twice, once in the row pass and once in the column pass, and with
// quad-cooperative additions (2.5 lanes' worth of issue slots each): 2 x 4.5 quad-additions per bucket where the sums
// themselves need 2.  This pass adds them ONCE, one logical thread per bucket, and leaves one value per bucket in
// bucket order, so the line sums read plain coalesced arrays.  Q = 4 (a quad per bucket) while the job is small enough
// for every bucket to get its quad at once (one bucket set: 2^15 quads), Q = 1 beyond.  grid: NB logical threads.
template <int Q>
__global__ void __launch_bounds__(256) msm_fold_buckets(const xyzz29_mem* __restrict__ partial, const uint32_t* __restrict__ toff,
                                                        const uint32_t* __restrict__ ntask, uint32_t NB,
                                                        xyzz29_mem* __restrict__ folded) {
  side_kernel_prio();
  const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) / Q, role = threadIdx.x % Q;
  if (b >= NB) return;
  const uint32_t nt = ntask[b], t0 = toff[b];
  xyzz29 acc = xyzz29_identity();
  if (nt) {
    acc = xyzz29_load(partial + t0);
    xyzz29 nxt = acc;
    if (nt > 1) nxt = xyzz29_load(partial + t0 + 1);
    for (uint32_t t = 1; t < nt; t++) {
      const xyzz29 cur = nxt;
      if (t + 1 < nt) nxt = xyzz29_load(partial + t0 + t + 1);   // in flight while the addition below runs
      add_q<Q>(acc, cur, role);
    }
  }
  if (role == 0) xyzz29_store(folded + b, acc);
}
// grid (rows + cols, sets): line sums over folded buckets (one value per bucket, bucket order).  lines[set * (rows + cols) + L]
template <int Q>
__global__ void __launch_bounds__(256) msm_reduce2d_lines_folded(const xyzz29_mem* __restrict__ folded, Reduce2dShape sh,
                                                                 xyzz29_mem* __restrict__ lines) {
  side_kernel_prio();
  __shared__ xyzz29_mem lds[256 / Q];
  const uint32_t rows = 1u << sh.log_rows, cols = 1u << sh.log_cols, L = blockIdx.x, set = blockIdx.y;
  const xyzz29_mem* src = folded + ((size_t)set << (sh.log_rows + sh.log_cols));
  const bool is_row = L < rows;
  const uint32_t n = is_row ? cols : rows;
  xyzz29 sum = wg_sum<Q>(n, [&](uint32_t e) {
    return xyzz29_load(src + (is_row ? (L << sh.log_cols) + e : (e << sh.log_cols) + (L - rows)));
  }, lds);
  if (threadIdx.x == 0) xyzz29_store(lines + (size_t)set * (rows + cols) + L, sum);
}
// (the one-launch form: line sums that add a bucket's partial sums themselves; msm.red2d_prefold = 0)
// grid (rows + cols, sets): line sums.  lines[set * (rows + cols) + L]
template <int Q>
__global__ void __launch_bounds__(256) msm_reduce2d_lines(const xyzz29_mem* __restrict__ partial,
                                                          const uint32_t* __restrict__ toff,
                                                          const uint32_t* __restrict__ ntask, Reduce2dShape sh,
                                                          xyzz29_mem* __restrict__ lines) {
  side_kernel_prio();
  __shared__ xyzz29_mem lds[256 / Q];
  const uint32_t rows = 1u << sh.log_rows, cols = 1u << sh.log_cols, L = blockIdx.x, set = blockIdx.y;
  const uint32_t base = set << (sh.log_rows + sh.log_cols);
  const bool is_row = L < rows;
  const uint32_t n = is_row ? cols : rows;
  // a bucket owns ntask[b] partial sums (one per accumulation task; several when the heavy-bucket merge was folded in
  // here: the line sums add them on the way, which costs the few extra additions of a merge round without its launches)
  auto get = [&](uint32_t e, xyzz29& acc, uint32_t role) {
    const uint32_t b = base + (is_row ? (L << sh.log_cols) + e : (e << sh.log_cols) + (L - rows));
    const uint32_t nt = ntask[b], t0 = toff[b];
    for (uint32_t t = 0; t < nt; t++) add_q<Q>(acc, xyzz29_load(partial + t0 + t), role);
  };
  xyzz29 sum = wg_sum_into<Q>(n, get, lds);
  if (threadIdx.x == 0) xyzz29_store(lines + (size_t)set * (rows + cols) + L, sum);
}
// grid (log_rows + log_cols + 1, sets): WG j < log_cols: columns with bit j of lo set; next log_rows: rows with
// bit j' of hi set; last: all columns (= the total).  out[set * (bits + 1) + j]
template <int Q>
__global__ void __launch_bounds__(256) msm_reduce2d_bits(const xyzz29_mem* __restrict__ lines, Reduce2dShape sh,
                                                         xyzz29_mem* __restrict__ out) {
  side_kernel_prio();
  __shared__ xyzz29_mem lds[256 / Q];
  const uint32_t rows = 1u << sh.log_rows, cols = 1u << sh.log_cols, j = blockIdx.x, set = blockIdx.y;
  const uint32_t bits = sh.log_rows + sh.log_cols;
  const xyzz29_mem* ln = lines + (size_t)set * (rows + cols);
  xyzz29 sum;
  if (j < sh.log_cols) {
    sum = wg_sum<Q>(cols, [&](uint32_t e) {
      xyzz29 v = xyzz29_identity();
      if ((e >> j) & 1) v = xyzz29_load(ln + rows + e);
      return v;
    }, lds);
  } else if (j < bits) {
    const uint32_t jr = j - sh.log_cols;
    sum = wg_sum<Q>(rows, [&](uint32_t e) {
      xyzz29 v = xyzz29_identity();
      if ((e >> jr) & 1) v = xyzz29_load(ln + e);
      return v;
    }, lds);
  } else {
    sum = wg_sum<Q>(cols, [&](uint32_t e) { return xyzz29_load(ln + rows + e); }, lds);
  }
  if (threadIdx.x == 0) xyzz29_store(out + (size_t)set * (bits + 1) + j, sum);
}
// many sets: the powers of two on the device as well -- thread t doubles term t t times (<= 14 doublings), then
// a tree sum: one point per set.  grid (sets), 32 threads
__global__ void __launch_bounds__(32) msm_reduce2d_combine(const xyzz29_mem* __restrict__ terms, uint32_t bits,
                                                           xyzz29_mem* __restrict__ out) {
  side_kernel_prio();
  __shared__ xyzz29_mem lds[32];
  const uint32_t t = threadIdx.x, set = blockIdx.x;
  xyzz29 p = xyzz29_identity();
  if (t <= bits) {
    p = xyzz29_load(terms + (size_t)set * (bits + 1) + t);
    if (t < bits)
      for (uint32_t i = 0; i < t; i++) p = xyzz29_double(p);   // term t < bits has weight 2^t; term `bits` (the total) 1
  }
  xyzz29_store(&lds[t], p);
  __syncthreads();
  tree_sum<1>(lds, 32, t, 0, true);
  if (t == 0) xyzz29_store(out + set, xyzz29_load(&lds[0]));
}
// canonical words of `count` XYZZ points for the host tail (32 words each)
// (`out` is page-locked host memory mapped into the device: the words land where the host tail reads them)
__global__ void msm_export_points(const xyzz29_mem* __restrict__ in, uint32_t count, uint32_t* __restrict__ out) {
  side_kernel_prio();
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= count) return;
  uint32_t w[32];
  xyzz29_to_words(xyzz29_load(in + q), w);
  uint4* o = reinterpret_cast<uint4*>(out + 32 * q);
  for (int i = 0; i < 8; i++) o[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
  __threadfence_system();
}

// per-window (A, S, T) -> canonical 8 x u32 Montgomery-2^256 words (X, Y, ZZ, ZZZ each) for the
// host tail; out[(3*j + which)*32 ..]
__global__ void msm_export_windows(ReduceOut in, uint32_t W, uint32_t has_t, uint32_t* __restrict__ out) {
  side_kernel_prio();
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= 3 * W) return;
  uint32_t j = q / 3, which = q - 3 * j;
  uint32_t w[32];
  if (which == 2 && !has_t) {
    for (int i = 0; i < 32; i++) w[i] = 0;
  } else {
    xyzz29_to_words(xyzz29_load((which == 0 ? in.a : which == 1 ? in.s : in.r) + j), w);
  }
  uint4* o = reinterpret_cast<uint4*>(out + 32 * q);
  for (int i = 0; i < 8; i++) o[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
  __threadfence_system();
}

// out[i] = scalars[i] * G  (ParamsKZG::setup's fixed-base products; also used to build
// synthetic bases for benchmarks).  One thread per scalar, double-and-add in XYZZ, result
// normalised on the device.
__global__ void __launch_bounds__(256) g1_fixed_base_mul(const fp_words* __restrict__ scalars, uint32_t n,
                                                         g1_affine_mem* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  typedef Fq29 P;
  words8 s;
  {
    f29 k = f29_zero();
    k.l[0] = 32;
    f29_to_words(f29_cond_sub_p<Fr29>(f29_mul<Fr29>(f29_load_r256<Fr29>(scalars + i), k)), s.l);
  }
  affine29 gen;
  {
    // G = (1, 2): Montgomery-2^256 words of 1 and 2, then the usual shifted load
    uint32_t w[16];
    f29 one256 = f29_const<P>(P::r256);             // limbs of 2^256 mod q == words of 1~
    f29_to_words(one256, w);
    f29 two = f29_cond_sub_p<P>(f29_normalize(f29_add(one256, one256)));
    f29_to_words(two, w + 8);
    gen = affine29_from_words(w);
  }
  xyzz29 acc = xyzz29_identity();
  for (int limb = 7; limb >= 0; limb--) {
    uint32_t w = s.l[7];
#pragma unroll
    for (int k = 7; k > 0; k--) s.l[k] = s.l[k - 1];
    s.l[0] = 0;
    for (int bit = 31; bit >= 0; bit--) {
      acc = xyzz29_double(acc);
      if ((w >> bit) & 1) xyzz29_madd(acc, gen);
    }
  }
  uint32_t ow[16];
  if (xyzz29_is_identity(acc)) {
    for (int k = 0; k < 16; k++) ow[k] = 0;
  } else {
    f29 iz = f29_inv<P>(acc.zzz);                            // 1/ZZZ
    f29 t = f29_mul<P>(acc.zz, iz);                          // ZZ/ZZZ = 1/Z
    f29 ax = f29_mul<P>(acc.x, f29_sqr<P>(t));               // X/ZZ
    f29 ay = f29_mul<P>(acc.y, iz);                          // Y/ZZZ
    f29_to_words(f29_reduce_with<P>(ax, P::r256), ow);
    f29_to_words(f29_reduce_with<P>(ay, P::r256), ow + 8);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) out[i].q[k] = make_uint4(ow[4 * k], ow[4 * k + 1], ow[4 * k + 2], ow[4 * k + 3]);
}

// ------------------------------------------------------------------ N5: FFT over G1
// ParamsKZG::downsize / g_to_lagrange: the same radix-2 butterfly as
// best_fft with group elements: a' = a + w*b, b' = a - w*b, w*b a 254-bit scalar
// multiplication.  Set-up time only (once per SRS), so: one thread per butterfly per stage,
// points in XYZZ limb form in global memory, twiddles computed on the fly.
__device__ inline xyzz29 xyzz29_scalar_mul(const xyzz29& p, const uint32_t k[8]) {
  xyzz29 acc = xyzz29_identity();
  for (int limb = 7; limb >= 0; limb--) {
    const uint32_t w = k[limb];
    for (int bit = 31; bit >= 0; bit--) {
      acc = xyzz29_double(acc);
      if ((w >> bit) & 1) xyzz29_add(acc, p);
    }
  }
  return acc;
}
// canonical integer words of x^ (2^261-domain Fr)
__device__ __forceinline__ void fr29_to_integer_words(const f29& x, uint32_t w[8]) {
  f29 one = f29_zero();
  one.l[0] = 1;
  f29_to_words(f29_cond_sub_p<Fr29>(f29_mul<Fr29>(x, one)), w);
}
__global__ void g1fft_load(const g1_affine_mem* __restrict__ in, uint32_t log_n, xyzz29_mem* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> log_n) return;
  g1_affine_mem raw = in[i];
  affine29 q = affine29_load(&raw);
  xyzz29 p = xyzz29_identity();
  xyzz29_madd(p, q);  // identity + q: reduces the lazy coordinates, sets ZZ = ZZZ = 1
  uint32_t r = log_n ? (__brev(i) >> (32 - log_n)) : 0;
  xyzz29_store(out + r, p);
}
__global__ void __launch_bounds__(128) g1fft_stage(xyzz29_mem* __restrict__ a, uint32_t log_n, uint32_t s, words8 omega) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >> (log_n - 1)) return;
  const uint32_t h = 1u << s, j = q & (h - 1), blk = q >> s;
  const uint32_t i0 = (blk << (s + 1)) + j, i1 = i0 + h;
  xyzz29 u = xyzz29_load(a + i0), v = xyzz29_load(a + i1);
  if (j) {
    uint32_t k[8];
    fr29_to_integer_words(f29_pow_u64<Fr29>(f29_words_to_r261<Fr29>(omega.l), (uint64_t)j << (log_n - s - 1)), k);
    v = xyzz29_scalar_mul(v, k);
  }
  xyzz29 sum = u;
  xyzz29_add(sum, v);
  if (!xyzz29_is_identity(v)) v.y = f29_sub<Fq29, 1>(f29_zero(), v.y);  // -v: Y < 4 -> 4p - Y
  xyzz29_add(u, v);
  xyzz29_store(a + i0, sum);
  xyzz29_store(a + i1, u);
}
// out[i] = scale * a[i], affine
__global__ void __launch_bounds__(128) g1fft_store(const xyzz29_mem* __restrict__ a, uint32_t log_n, words8 scale,
                                                   uint32_t has_scale, g1_affine_mem* __restrict__ out) {
  typedef Fq29 P;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> log_n) return;
  xyzz29 p = xyzz29_load(a + i);
  if (has_scale) {
    uint32_t k[8];
    fr29_to_integer_words(f29_words_to_r261<Fr29>(scale.l), k);
    p = xyzz29_scalar_mul(p, k);
  }
  uint32_t ow[16];
  if (xyzz29_is_identity(p)) {
    for (int k = 0; k < 16; k++) ow[k] = 0;
  } else {
    f29 iz = f29_inv<P>(p.zzz);
    f29 t = f29_mul<P>(p.zz, iz);
    f29_to_words(f29_reduce_with<P>(f29_mul<P>(p.x, f29_sqr<P>(t)), P::r256), ow);
    f29_to_words(f29_reduce_with<P>(f29_mul<P>(p.y, iz), P::r256), ow + 8);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) out[i].q[k] = make_uint4(ow[4 * k], ow[4 * k + 1], ow[4 * k + 2], ow[4 * k + 3]);
}

hipError_t g1_fft(const g1_affine_mem* d_in, g1_affine_mem* d_out, uint32_t log_n, const words8& omega,
                  const words8* scale, xyzz29_mem* d_work, hipStream_t stream) {
  const uint32_t n = 1u << log_n;
  g1fft_load<<<(n + 127) / 128, 128, 0, stream>>>(d_in, log_n, d_work);
  for (uint32_t s = 0; s < log_n; s++)
    g1fft_stage<<<(n / 2 + 127) / 128, 128, 0, stream>>>(d_work, log_n, s, omega);
  words8 sc = scale ? *scale : omega;
  g1fft_store<<<(n + 127) / 128, 128, 0, stream>>>(d_work, log_n, sc, scale ? 1u : 0u, d_out);
  return hipGetLastError();
}

// ------------------------------------------------------------------ prefix sums of a basis (difference-form commits)
// Blocked scan over points: every thread runs through PFX_CHUNK consecutive elements (inclusive running sums in place,
// chunk total to the next level), the totals are scanned the same way recursively, then the offsets are added on the
// way down; the last step also normalises to affine.  Once per SRS.
static constexpr uint32_t PFX_CHUNK = 32;
template <bool AFFINE>
__global__ void __launch_bounds__(128) g1_prefix_chunks(const g1_affine_mem* __restrict__ in, xyzz29_mem* __restrict__ run, uint32_t n,
                                                       xyzz29_mem* __restrict__ totals) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t first = t * PFX_CHUNK;
  if (first >= n) return;
  const uint32_t last = min(n, first + PFX_CHUNK);
  xyzz29 acc = xyzz29_identity();
  for (uint32_t i = first; i < last; i++) {
    if (AFFINE) {
      g1_affine_mem raw = in[i];
      xyzz29_madd(acc, affine29_load(&raw));
    } else {
      xyzz29_add(acc, xyzz29_load(run + i));
    }
    xyzz29_store(run + i, acc);
  }
  if (totals) xyzz29_store(totals + t, acc);
}
// run[i] += upper[i / PFX_CHUNK - 1]  (upper = inclusive prefix sums of the chunk totals)
__global__ void __launch_bounds__(128) g1_prefix_apply(xyzz29_mem* __restrict__ run, uint32_t n, const xyzz29_mem* __restrict__ upper) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || i < PFX_CHUNK) return;
  xyzz29 v = xyzz29_load(run + i);
  xyzz29_add(v, xyzz29_load(upper + i / PFX_CHUNK - 1));
  xyzz29_store(run + i, v);
}
__global__ void __launch_bounds__(128) g1_prefix_store(const xyzz29_mem* __restrict__ run, uint32_t n, const xyzz29_mem* __restrict__ upper,
                                                      g1_affine_mem* __restrict__ out) {
  typedef Fq29 P;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  xyzz29 p = xyzz29_load(run + i);
  if (upper && i >= PFX_CHUNK) xyzz29_add(p, xyzz29_load(upper + i / PFX_CHUNK - 1));
  uint32_t ow[16];
  if (xyzz29_is_identity(p)) {
    for (int k = 0; k < 16; k++) ow[k] = 0;
  } else {
    f29 iz = f29_inv<P>(p.zzz);
    f29 t = f29_mul<P>(p.zz, iz);
    f29_to_words(f29_reduce_with<P>(f29_mul<P>(p.x, f29_sqr<P>(t)), P::r256), ow);
    f29_to_words(f29_reduce_with<P>(f29_mul<P>(p.y, iz), P::r256), ow + 8);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) out[i].q[k] = make_uint4(ow[4 * k], ow[4 * k + 1], ow[4 * k + 2], ow[4 * k + 3]);
}
hipError_t g1_prefix_sums(const g1_affine_mem* d_in, size_t n, g1_affine_mem* d_out, hipStream_t stream) {
  if (!n) return hipSuccess;
  if (n >= (1ull << 31) || d_in == d_out) return hipErrorInvalidValue;
  // level sizes: n, ceil(n / 32), ... down to one chunk
  std::vector<size_t> size{n};
  while (size.back() > PFX_CHUNK) size.push_back((size.back() + PFX_CHUNK - 1) / PFX_CHUNK);
  size_t total = 0;
  for (size_t v : size) total += v;
  xyzz29_mem* work = nullptr;
  hipError_t e = hipMalloc(&work, total * sizeof(xyzz29_mem));
  if (e != hipSuccess) return e;
  std::vector<xyzz29_mem*> lvl(size.size());
  lvl[0] = work;
  for (size_t l = 1; l < size.size(); l++) lvl[l] = lvl[l - 1] + size[l - 1];
  auto grid = [](size_t threads) { return (unsigned)((threads + 127) / 128); };
  for (size_t l = 0; l < size.size(); l++) {   // up: running sums per chunk, totals to the next level
    const size_t chunks = (size[l] + PFX_CHUNK - 1) / PFX_CHUNK;
    xyzz29_mem* totals = l + 1 < size.size() ? lvl[l + 1] : nullptr;
    if (l == 0) g1_prefix_chunks<true><<<grid(chunks), 128, 0, stream>>>(d_in, lvl[0], (uint32_t)size[0], totals);
    else g1_prefix_chunks<false><<<grid(chunks), 128, 0, stream>>>(nullptr, lvl[l], (uint32_t)size[l], totals);
  }
  for (size_t l = size.size() - 1; l-- > 1;)   // down: levels size-2 .. 1 become global prefix sums
    g1_prefix_apply<<<grid(size[l]), 128, 0, stream>>>(lvl[l], (uint32_t)size[l], lvl[l + 1]);
  g1_prefix_store<<<grid(n), 128, 0, stream>>>(lvl[0], (uint32_t)n, size.size() > 1 ? lvl[1] : nullptr, d_out);
  e = hipGetLastError();
  if (e == hipSuccess) e = host_wait_stream(stream);
  (void)hipFree(work);
  return e;
}

// ------------------------------------------------------------------ fixed-base window table
// next[i] = 2^doublings * prev[i], affine (one inversion per point: this runs once per SRS)
__global__ void __launch_bounds__(128) msm_table_step(const g1_affine_mem* __restrict__ prev, uint32_t n,
                                                      uint32_t doublings, g1_affine_mem* __restrict__ next) {
  typedef Fq29 P;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  g1_affine_mem raw = prev[i];
  uint32_t ow[16];
  bool ident = true;
#pragma unroll
  for (int k = 0; k < 4; k++) ident = ident && !(raw.q[k].x | raw.q[k].y | raw.q[k].z | raw.q[k].w);
  if (ident || doublings == 0) {
    next[i] = raw;
    return;
  }
  affine29 q = affine29_load(&raw);
  xyzz29 acc = xyzz29_double_affine(q);
  for (uint32_t d = 1; d < doublings; d++) acc = xyzz29_double(acc);
  if (xyzz29_is_identity(acc)) {
    for (int k = 0; k < 16; k++) ow[k] = 0;
  } else {
    f29 iz = f29_inv<P>(acc.zzz);
    f29 t = f29_mul<P>(acc.zz, iz);
    f29_to_words(f29_reduce_with<P>(f29_mul<P>(acc.x, f29_sqr<P>(t)), P::r256), ow);
    f29_to_words(f29_reduce_with<P>(f29_mul<P>(acc.y, iz), P::r256), ow + 8);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) next[i].q[k] = make_uint4(ow[4 * k], ow[4 * k + 1], ow[4 * k + 2], ow[4 * k + 3]);
}

// ------------------------------------------------------------------ host driver
#define SG_TRY(x)                      \
  do {                                 \
    hipError_t _e = (x);               \
    if (_e != hipSuccess) return _e;   \
  } while (0)

MsmEngine::~MsmEngine() { release(); }

// Accumulations of different jobs never share the device: each one alone keeps the vector ALUs busy, and two polite ones
// side by side fill the register file that politeness leaves to the other kernels.  Every accumulation launch waits for
// the event recorded after the previous one (whichever engine / stream launched it).
// (process-wide state: the library serves ONE device per process -- sg_init binds it, one process per GPU -- so "of the
// process" is "of the device")
static std::mutex g_acc_chain_mu;
static std::atomic<int> g_jobs_in_flight{0};   // jobs of this process between their first kernel and the end of their host tail
static hipEvent_t g_acc_chain_last = nullptr;   // recorded after the most recent accumulation launch of the process
// wait for the previous accumulation, launch, record -- under ONE hold of the mutex: taken separately around the launch, two
// lanes could both wait on the same predecessor and then run their accumulations side by side, which is what the chain is for
// Launch log (parameter "msm.acc_log", profiling only): one record per msm_accumulate launch, appended under the chain's mutex,
// i.e. in the order in which the chained launches run on the device -- the i-th msm_accumulate of a kernel trace ordered by
// start time IS the i-th record, so a profile attributes every launch to its job exactly (tools/proof_budget.py)
static std::atomic<int> g_acc_log_on{0};
static std::vector<AccLaunchRecord> g_acc_log;
void msm_acc_log_enable(bool on) {
  std::lock_guard<std::mutex> lk(g_acc_chain_mu);
  g_acc_log_on.store(on ? 1 : 0);
  if (on) g_acc_log.clear();
}
size_t msm_acc_log_read(AccLaunchRecord* out, size_t cap) {
  std::lock_guard<std::mutex> lk(g_acc_chain_mu);
  const size_t m = std::min(cap, g_acc_log.size());
  if (out && m) std::memcpy(out, g_acc_log.data(), m * sizeof(AccLaunchRecord));
  return g_acc_log.size();
}
hipError_t MsmEngine::chained_accumulate(hipStream_t stream, hipEvent_t after_wait, const std::function<void()>& launch) {
  std::unique_lock<std::mutex> lk(g_acc_chain_mu, std::defer_lock);
  if (cfg_.acc_chain || g_acc_log_on.load()) lk.lock();
  if (cfg_.acc_chain) {
    if (g_acc_chain_last) SG_TRY(hipStreamWaitEvent(stream, g_acc_chain_last, 0));
  }
  if (g_acc_log_on.load()) {
    const Job& j = job_;
    g_acc_log.push_back(AccLaunchRecord{(uint64_t)j.entries, (uint32_t)j.n, j.M, j.acc_threads, j.fixed ? 1u : 0u,
                                        (uint32_t)g_jobs_in_flight.load(), 1u << j.log_L});
  }
  if (after_wait) SG_TRY(hipEventRecord(after_wait, stream));   // timing mode: the accumulation's own start, behind the chain wait
  launch();
  SG_TRY(hipGetLastError());
  if (cfg_.acc_chain) {
    // two events per engine, alternating: the previous record of this engine may still be the one another stream waits on
    chain_slot_ ^= 1;
    if (!ev_chain_[chain_slot_]) SG_TRY(hipEventCreateWithFlags(&ev_chain_[chain_slot_], hipEventDisableTiming));
    SG_TRY(hipEventRecord(ev_chain_[chain_slot_], stream));
    g_acc_chain_last = ev_chain_[chain_slot_];
  }
  return hipSuccess;
}
void MsmEngine::release() {
  mark_in_flight(false);
  {
    std::lock_guard<std::mutex> lk(g_acc_chain_mu);
    for (auto& e : ev_chain_) {
      if (e && g_acc_chain_last == e) g_acc_chain_last = nullptr;
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
  }
  fe_.release(); tbase_.release(); win_words_.release(); trace_.release(); part_entry_.release(); part_fine_.release(); ccnt_.release(); coff_.release(); dig_.release(); thist_.release(); order_.release(); sorted_.release(); counts_.release(); off_.release(); hist_.release(); bsum_.release(); meta_.release();
  for (int i = 0; i < 2; i++) {
    ntask_[i].release(); toff_[i].release(); partial_[i].release(); red_a_[i].release(); red_s_[i].release(); red_r_[i].release();
  }
  if (ev_acc_) (void)hipEventDestroy(ev_acc_);
  ev_acc_ = nullptr;
  if (ev_meta_) (void)hipEventDestroy(ev_meta_);
  if (ev_done_) (void)hipEventDestroy(ev_done_);
  if (ev_tiny_) (void)hipEventDestroy(ev_tiny_);
  ev_meta_ = ev_done_ = ev_tiny_ = nullptr;
  if (h_meta_) (void)hipHostFree(h_meta_);
  if (h_win_) (void)hipHostFree(h_win_);
  if (h_tiny_) (void)hipHostFree(h_tiny_);
  h_meta_ = nullptr;
  h_win_ = nullptr;
  h_tiny_ = d_tiny_ = nullptr;
}

// Window size, from the measured sweeps (profiles/r01_sweeps/sweep_c2.txt): a single MSM is
// partly latency-bound (few, deep tasks) and prefers larger windows (log2 n - 2); a fused batch
// is throughput-bound and prefers the work-optimal log2 n - 4.
uint32_t MsmEngine::window_bits_for(size_t n, bool fused) const {
  if (cfg_.window_bits) return std::min<uint32_t>(16, std::max<uint32_t>(4, cfg_.window_bits));
  uint32_t lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) lg++;
  int c = (int)lg - (fused ? 4 : 2);
  return (uint32_t)std::min(16, std::max(4, c));
}

hipError_t MsmEngine::init() {
  {
    int dev = 0, cus = 0;
    SG_TRY(hipGetDevice(&dev));
    SG_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus_ = cus > 0 ? (uint32_t)cus : 256u;
  }
  SG_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_reduce_buckets<1>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  SG_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_reduce_buckets_lean<1>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  SG_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_reduce_items<1>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
  SG_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_hist), hipFuncAttributeMaxDynamicSharedMemorySize,
                             128 * 1024));
  SG_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                             128 * 1024));
  SG_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_partition), hipFuncAttributeMaxDynamicSharedMemorySize,
                             96 * 1024));
  SG_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_fine_sort), hipFuncAttributeMaxDynamicSharedMemorySize,
                             128 * 1024));
  return hipSuccess;
}

// the same three scans for NB <= 4 Ki buckets in ONE workgroup (a small MSM is a chain of ~25 launches with
// a ~5 us floor each; this replaces four of them -- memset + three kernels -- per scan): thread t owns
// buckets [t*per, (t+1)*per)
static constexpr uint32_t SCAN_SMALL_PER = 4;   // x 1024 threads = 4 Ki buckets (beyond that the strided stores cost more than the launches saved)
__global__ void __launch_bounds__(1024) msm_scan_small(const uint32_t* __restrict__ cnt, uint32_t NB, uint32_t log_L,
                                                       uint32_t* __restrict__ off, uint32_t* __restrict__ ntask,
                                                       uint32_t* __restrict__ toff, uint32_t* __restrict__ meta,
                                                       volatile uint32_t* host_meta) {
  side_kernel_prio();
  __shared__ uint32_t s_a[1024], s_t[1024], s_m[1024];
  const uint32_t Lm1 = (1u << log_L) - 1, tid = threadIdx.x;
  const uint32_t per = (NB + 1023) / 1024, lo = min(tid * per, NB);
  uint32_t v[SCAN_SMALL_PER];                      // all loads of a thread in flight together
#pragma unroll
  for (uint32_t k = 0; k < SCAN_SMALL_PER; k++) v[k] = (k < per && lo + k < NB) ? cnt[lo + k] : 0u;
  uint32_t a = 0, t = 0, m = 0;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_SMALL_PER; k++) {
    a += v[k];
    t += (v[k] + Lm1) >> log_L;
    m = max(m, v[k]);
  }
  s_a[tid] = a; s_t[tid] = t; s_m[tid] = m;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t va = 0, vt = 0, vm = 0;
    if (tid >= d) { va = s_a[tid - d]; vt = s_t[tid - d]; vm = s_m[tid - d]; }
    __syncthreads();
    s_a[tid] += va; s_t[tid] += vt; s_m[tid] = max(s_m[tid], vm);
    __syncthreads();
  }
  uint32_t ra = s_a[tid] - a, rt = s_t[tid] - t;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_SMALL_PER; k++) {
    if (k < per && lo + k < NB) {
      const uint32_t nt = (v[k] + Lm1) >> log_L;
      if (off) off[lo + k] = ra;
      ntask[lo + k] = nt;
      toff[lo + k] = rt;
      ra += v[k];
      rt += nt;
    }
  }
  if (tid == 1023) {
    if (off) off[NB] = s_a[tid];
    toff[NB] = s_t[tid];
    meta[0] = s_a[tid];
    meta[1] = s_t[tid];
    meta[2] = s_m[tid];
    if (host_meta) {
      host_meta[0] = s_a[tid];
      host_meta[1] = s_t[tid];
      host_meta[2] = s_m[tid];
      __threadfence_system();
    }
  }
}

// exclusive scans over NB buckets (one launch for small NB, else two); host_meta: see msm_scan_sums
static hipError_t launch_scan(const uint32_t* cnt, uint32_t NB, uint32_t log_L, uint32_t* off, uint32_t* ntask,
                              uint32_t* toff, uint32_t* bsum, uint32_t* meta, hipStream_t stream, uint32_t* host_meta = nullptr) {
  if (NB <= SCAN_SMALL_PER * 1024) {
    msm_scan_small<<<1, 1024, 0, stream>>>(cnt, NB, log_L, off, ntask, toff, meta, host_meta);
    return hipGetLastError();
  }
  const uint32_t nblk = (NB + SCAN_BLOCK - 1) / SCAN_BLOCK;
  if (nblk > 1024) return hipErrorInvalidValue;
  msm_scan_sums<<<nblk, SCAN_THREADS, 0, stream>>>(cnt, NB, log_L, bsum, meta, off, toff, host_meta);   // + the scan of the block sums
  msm_scan_write<<<nblk, SCAN_THREADS, 0, stream>>>(cnt, NB, log_L, bsum, off, ntask, toff);
  return hipGetLastError();
}

// ---- phase 1: everything up to the counting sort; ends with an async copy of the counters
size_t MsmEngine::max_fused(size_t n) const {
  if (n == 0) return MAX_FUSED;
  const uint32_t c = window_bits_for(n, true);
  const uint32_t W = (255 + c - 1) / c;
  const size_t nb = (size_t)W << (c - 1);
  // the scans handle 2^21 buckets (1024 blocks x 2048); cap the fused work space (cfg: log_fuse_entries)
  const size_t by_buckets = ((size_t)1 << 21) / nb;
  const size_t by_entries = ((size_t)1 << cfg_.log_fuse_entries) / std::max<size_t>(1, (size_t)W * n);
  return std::max<size_t>(1, std::min<size_t>(std::min(by_buckets, by_entries), MAX_FUSED));
}

// window widths: W-1 signed windows + an unsigned top window, 254 bits in total
WindowPlan make_window_plan(uint32_t c) {
  WindowPlan wp{};
  const uint32_t W1 = wp.W = (255 + c - 1) / c;  // W*c >= 255: the top window never carries out
  for (uint32_t q = 0; q + 1 < W1; q++) wp.width[q] = (uint8_t)c;
  wp.width[W1 - 1] = (uint8_t)(c - 1);
  for (uint32_t k = 0, slack = W1 * c - 255; k < slack; k++) wp.width[W1 - 2 - k] -= 1;
  return wp;
}

uint32_t fixed_window_bits_for(size_t n) {
  uint32_t lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) lg++;
  return std::min<uint32_t>(16, std::max<uint32_t>(4, lg));  // measured: tools/sweep_fixed_c.py (k = 11 .. 17)
}

// table rows: row w = 2^(offset of window w) * bases, W x n affine points
hipError_t build_window_table(const g1_affine_mem* d_bases, size_t n, uint32_t c, FixedTable* out, hipStream_t stream) {
  if (n == 0 || n >= (1ull << 31) || c < 4 || c > 16) return hipErrorInvalidValue;
  FixedTable t;
  t.c = c;
  t.n = n;
  t.wp = make_window_plan(c);
  if ((size_t)t.wp.W * n >= (1ull << 31)) return hipErrorInvalidValue;
  SG_TRY(hipMalloc(&t.table, sizeof(g1_affine_mem) * n * t.wp.W));
  hipError_t e = hipMemcpyAsync(t.table, d_bases, sizeof(g1_affine_mem) * n, hipMemcpyDeviceToDevice, stream);
  for (uint32_t w = 1; w < t.wp.W && e == hipSuccess; w++) {
    msm_table_step<<<(unsigned)((n + 127) / 128), 128, 0, stream>>>(t.table + (size_t)(w - 1) * n, (uint32_t)n,
                                                                    t.wp.width[w - 1], t.table + (size_t)w * n);
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    (void)hipFree(t.table);
    return e;
  }
  *out = t;
  return hipSuccess;
}

hipError_t MsmEngine::enqueue_front_fixed(const fp_words* const* d_scalars, const FixedTable& tab, size_t M, size_t n,
                                          hipStream_t stream, uint8_t* out_affine, MsmTimings* tm,
                                          const g1_affine_mem* const* tables, uint64_t diff_mask) {
  if (n > tab.n || !tab.table) return hipErrorInvalidValue;
  if (diff_mask && n != tab.n) return hipErrorInvalidValue;   // s[n] = 0 closes the telescoping sum only at full length
  const g1_affine_mem* bs[MAX_FUSED];
  // `tables` (optional): one window table per MSM, all built with tab's plan (same c and n), e.g. g and g_lagrange
  for (size_t m = 0; m < M && m < MAX_FUSED; m++) bs[m] = tables ? tables[m] : tab.table;
  fixed_ = &tab;
  diff_mask_ = diff_mask;
  hipError_t e = enqueue_front_fused(d_scalars, bs, M, n, stream, out_affine, tm);
  fixed_ = nullptr;
  diff_mask_ = 0;
  return e;
}
size_t MsmEngine::max_fused_fixed(const FixedTable& tab, size_t n) const {
  if (n == 0) return MAX_FUSED;
  const size_t by_buckets = ((size_t)1 << 21) >> (tab.c - 1);
  const size_t by_entries = ((size_t)1 << cfg_.log_fuse_entries_fixed) / std::max<size_t>(1, (size_t)tab.wp.W * n);
  return std::max<size_t>(1, std::min<size_t>(std::min(by_buckets, by_entries), MAX_FUSED));
}

hipError_t MsmEngine::enqueue_front(const fp_words* d_scalars, const g1_affine_mem* d_bases, size_t n,
                                    hipStream_t stream, uint8_t* out_affine, MsmTimings* tm) {
  const fp_words* sc[1] = {d_scalars};
  const g1_affine_mem* bs[1] = {d_bases};
  return enqueue_front_fused(sc, bs, 1, n, stream, out_affine, tm);
}

// M independent MSMs of the same length n as ONE job: every kernel covers M*W windows, so the
// latency-bound phases (scans, bucket reduction, host round trips) are paid once per batch
hipError_t MsmEngine::enqueue_front_fused(const fp_words* const* d_scalars, const g1_affine_mem* const* d_bases,
                                          size_t M, size_t n, hipStream_t stream, uint8_t* out_affine,
                                          MsmTimings* tm) {
  // in flight from its first kernel on: an accumulation launched while ANOTHER job is still sorting must already leave
  // room for that sort (counted from the accumulation launch only, the first accumulation after a pause took the whole
  // register file and the other callers' front-ends waited a millisecond behind it)
  mark_in_flight(true);
  const hipError_t e = enqueue_front_fused_impl(d_scalars, d_bases, M, n, stream, out_affine, tm);
  if (e != hipSuccess) mark_in_flight(false);
  return e;
}
hipError_t MsmEngine::enqueue_front_fused_impl(const fp_words* const* d_scalars, const g1_affine_mem* const* d_bases,
                                               size_t M, size_t n, hipStream_t stream, uint8_t* out_affine,
                                               MsmTimings* tm) {
  Job& j = job_;
  j = Job{};
  j.M = (uint32_t)M; j.n = n; j.stream = stream; j.out = out_affine; j.tm = tm;
  if (tm) *tm = MsmTimings{};
  if (n == 0 || M == 0) {
    j.trivial = true;
    return hipSuccess;
  }
  if (n >= (1ull << 31) || M > MAX_FUSED) return hipErrorInvalidValue;
  for (size_t m = 0; m < M; m++) {
    j.bp.scalars[m] = d_scalars[m];
    j.bp.bases[m] = d_bases[m];
  }
  j.bp.diff_mask = diff_mask_;
  // fixed-base mode: the job's windows all land in one bucket set per MSM (see msm_scatter)
  j.fixed = fixed_ != nullptr;
  j.n_tab = j.fixed ? (uint32_t)fixed_->n : 0;
  const uint32_t c = j.c = j.fixed ? fixed_->c : window_bits_for(n, M > 1);
  j.wp = j.fixed ? fixed_->wp : make_window_plan(c);
  const uint32_t W1 = j.wp.W;
  const uint32_t W = W1 * (uint32_t)M;             // digit rows of the whole fused job
  const uint32_t nbw = j.nbw = 1u << (c - 1);
  const uint32_t NB = j.NB = (j.fixed ? (uint32_t)M : W) * nbw;
  if (NB > (1u << 21)) return hipErrorInvalidValue;
  const size_t entries = (size_t)W * n;
  j.entries = entries;
  if (entries >= ((size_t)1 << 32)) return hipErrorInvalidValue;  // bucket offsets are 32-bit (n <= 2^27 at c = 16)
  // task length: deep enough to amortise, shallow enough that the longest dependent chain of
  // additions stays a small multiple of the per-lane share of the work
  j.log_L = cfg_.log_seg;
  if (!j.log_L) {
    const size_t share = 2 * entries / (256 * 4 * 64 * 4);  // entries per resident lane, x2
    // small jobs are pure latency chains: shorter tasks (more lanes, more merging) win -- measured at k = 11 .. 17
    // (tools/sweep_seg_batch.sh, time_fixed_phases.py: below ~12 M entries the chip is not full and long tasks only
    // lengthen the chain: 6.3 M entries, L = 64 -> 16: 1.25 -> 1.02 ms)
    const size_t depth = entries / NB;  // mean entries per bucket
    if (entries < ((size_t)1 << 16)) j.log_L = 2;
    else if (entries < ((size_t)1 << 19)) j.log_L = 3;
    else if (depth < 40 && entries >= ((size_t)1 << 20)) {
      // large jobs with shallow buckets (arbitrary bases: ~n / 2^(c-1) per bucket): a task is a whole bucket, and L
      // only has to exceed the largest bucket so that no merge round is needed (k = 18: L = 16 -> 64: 0.99 -> 0.88 ms)
      // (uniform scalars: the largest of NB Poisson(depth) buckets is ~ depth + 6 sqrt(depth), 66 at depth 32)
      size_t need = depth + 8;
      for (size_t r = 1; r * r <= 64 * depth; r++) need = depth + 8 + r;  // + 8 sqrt(depth)
      j.log_L = 6;
      while (j.log_L < 8 && (((size_t)1 << j.log_L) < share || ((size_t)1 << j.log_L) < need)) j.log_L++;
    }
    else if (entries <= (size_t)7 << 20) j.log_L = 4;
    else if (entries <= (size_t)12 << 20) j.log_L = 5;
    else {
      j.log_L = 4;
      while (j.log_L < 8 && ((size_t)1 << j.log_L) < share) j.log_L++;
    }
  }
  // two-pass sort (msm_partition / msm_fine_sort) for everything but small jobs: B coarse bins per
  // bucket set, sized for ~4 Ki entries per bin (half an LDS tile, so Poisson tails still fit)
  const uint32_t sets = j.fixed ? (uint32_t)M : W;
  const size_t set_entries = entries / sets;
  const bool two_pass = cfg_.two_pass == 2 || (cfg_.two_pass == 1 && entries >= ((size_t)1 << 19));  // measured crossover
  uint32_t B = 1, shift = c - 1;
  if (two_pass) {
    while (B < 1024 && B < nbw && (set_entries / B > 4096 || (nbw / B) > 8192)) B <<= 1;
    shift = 0;
    while ((nbw >> shift) > B) shift++;
  }
  // chunking of the scalars for the LDS-staged counting sort: W * P workgroups
  const uint32_t target_wgs = two_pass ? 1024 : (nbw * 4 > 64 * 1024) ? 256 : 512;
  uint32_t P = std::max<uint32_t>(1, target_wgs / W);
  const uint32_t chunk = (uint32_t)std::max<size_t>(two_pass ? SORT_TILE : 1024, (n + P - 1) / P);
  P = (uint32_t)((n + chunk - 1) / chunk);

  // workspace (grown on demand, kept across calls)
  SG_TRY(dig_.reserve(entries));
  SG_TRY(sorted_.reserve(entries));
  SG_TRY(hist_.reserve((size_t)W * P * (two_pass ? B : nbw)));
  if (two_pass) {
    SG_TRY(part_entry_.reserve(entries));
    SG_TRY(part_fine_.reserve(entries));
    SG_TRY(ccnt_.reserve((size_t)sets * B + 1));
    SG_TRY(coff_.reserve((size_t)sets * B + 1));
  }
  SG_TRY(bsum_.reserve(3 * 1024));
  SG_TRY(counts_.reserve((size_t)NB + 1));
  SG_TRY(off_.reserve((size_t)NB + 1));
  for (int i = 0; i < 2; i++) {
    SG_TRY(ntask_[i].reserve((size_t)NB + 1));
    SG_TRY(toff_[i].reserve((size_t)NB + 1));
  }
  {
    const uint32_t* before = meta_.p;
    SG_TRY(meta_.reserve(16));
    if (meta_.p != before) SG_TRY(hipMemsetAsync(meta_.p, 0, 16 * sizeof(uint32_t), stream));   // SCAN_DONE starts at zero (msm_scan_sums keeps it there)
  }
  // the counters and the window sums reach the host through page-locked memory the kernels write directly (mapped,
  // coherent): no copy kernels between the producing kernel and the event the host waits for
  if (!h_meta_) {
    SG_TRY(hipHostMalloc(&h_meta_, 16 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
    SG_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_hmeta_), h_meta_, 0));
  }
  {
    const size_t need = std::max<size_t>((size_t)W * 96, (size_t)W * 32 * 17);   // (A, S, T) per window, or bits + 1 points per set
    if (h_win_cap_ < need) {
      if (h_win_) (void)hipHostFree(h_win_);   // (the previous job has finished: finish() waited for its event)
      h_win_ = nullptr;
      SG_TRY(hipHostMalloc(&h_win_, need * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
      SG_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_hwin_), h_win_, 0));
      h_win_cap_ = need;
    }
  }
  if (!ev_meta_) SG_TRY(hipEventCreateWithFlags(&ev_meta_, hipEventDisableTiming));
  if (!ev_done_) SG_TRY(hipEventCreateWithFlags(&ev_done_, hipEventDisableTiming));
  if (tm) {
    for (auto& e : j.ev) SG_TRY(hipEventCreate(&e));
    SG_TRY(hipEventRecord(j.ev[0], stream));
  }

  msm_digits<<<dim3((unsigned)((n + 255) / 256), (unsigned)M), 256, 0, stream>>>(j.bp, (uint32_t)n, j.wp, dig_.p);
  if (tm) SG_TRY(hipEventRecord(j.ev[1], stream));
  if (two_pass) {
    const uint32_t NBc = sets * B;
    // round 5: the scans and the task-length histogram ride on the sort's own kernels (msm_hist_prefix's and
    // msm_fine_sort_fused's last workgroups): five launches fewer per job
    // ... for a job that has the device to itself (a blocking MSM 1.73 -> 1.70 ms, a proof's commitment jobs 26 launches
    // fewer); with other jobs in flight the separate small kernels slip in beside the running accumulation more easily than one
    // heavier sort pass does (three MSMs in flight: 770 -> 745 M points/s with the fused form, profiles/r05_sweeps/frontend.txt):
    // 1 = by that rule, 2 = always, 0 = never
    j.fe = NBc <= FE_MAX_BINS && (cfg_.fused_frontend == 2 || (cfg_.fused_frontend == 1 && !others_in_flight()));
    FrontEndScan fe_scan{nullptr, nullptr, nullptr, 0u, 0u};
    if (j.fe) {
      const uint32_t* before = fe_.p;
      SG_TRY(fe_.reserve(FE_WORDS));
      if (fe_.p != before) SG_TRY(hipMemsetAsync(fe_.p, 0, fe_.cap * sizeof(uint32_t), stream));   // counters and histogram start at zero; every job leaves them there
      SG_TRY(tbase_.reserve((size_t)NBc + 1));
      fe_scan = FrontEndScan{fe_.p + 3, coff_.p, tbase_.p, 1u << shift, j.log_L};
    }
    msm_hist<<<dim3(W, P), 1024, B * sizeof(uint32_t), stream>>>(dig_.p, (uint32_t)n, chunk, B, shift, hist_.p);
    msm_hist_prefix<<<(NBc + HP_BUCKETS - 1) / HP_BUCKETS, HP_BUCKETS * HP_GROUPS, 0, stream>>>(
        hist_.p, j.fixed ? W1 * P : P, B, NBc, ccnt_.p, fe_scan);
    // bin offsets (the task outputs of this scan are scratch)
    if (!j.fe) SG_TRY(launch_scan(ccnt_.p, NBc, j.log_L, coff_.p, ntask_[1].p, toff_[1].p, bsum_.p, meta_.p, stream));
    msm_partition<<<dim3(W, P), 1024, (3 * B + SORT_TILE) * sizeof(uint32_t) + SORT_TILE * sizeof(uint16_t), stream>>>(
        dig_.p, (uint32_t)n, chunk, B, shift, hist_.p, coff_.p, j.fixed ? W1 : 0u, j.n_tab, part_entry_.p, part_fine_.p);
    const uint32_t F = 1u << shift, fs_threads = 512;
    if (j.fe) {
      fe_parity_ ^= 1u;
      j.fe_parity = fe_parity_;
      j.NBc = NBc;
      msm_fine_sort_fused<<<NBc, fs_threads, (2 * F + 2 * fs_threads + SORT_TILE) * sizeof(uint32_t), stream>>>(
          part_entry_.p, part_fine_.p, coff_.p, ccnt_.p, B, shift, nbw, counts_.p, sorted_.p,
          FrontEndOut{off_.p, ntask_[0].p, toff_[0].p, tbase_.p, fe_.p, j.log_L, j.fe_parity});
    } else {
      msm_fine_sort<<<NBc, fs_threads, (2 * F + fs_threads + SORT_TILE) * sizeof(uint32_t), stream>>>(
          part_entry_.p, part_fine_.p, coff_.p, ccnt_.p, B, shift, nbw, counts_.p, sorted_.p);
      SG_TRY(launch_scan(counts_.p, NB, j.log_L, off_.p, ntask_[0].p, toff_[0].p, bsum_.p, meta_.p, stream, d_hmeta_));
    }
    if (!j.fe) SG_TRY(hipEventRecord(ev_meta_, stream));   // (fused front end: the totals come with the task order, enqueue_back)
  } else {
    msm_hist<<<dim3(W, P), 1024, nbw * sizeof(uint32_t), stream>>>(dig_.p, (uint32_t)n, chunk, nbw, 0, hist_.p);
    msm_hist_prefix<<<(NB + HP_BUCKETS - 1) / HP_BUCKETS, HP_BUCKETS * HP_GROUPS, 0, stream>>>(
        hist_.p, j.fixed ? W1 * P : P, nbw, NB, counts_.p, FrontEndScan{nullptr, nullptr, nullptr, 0u, 0u});
    SG_TRY(launch_scan(counts_.p, NB, j.log_L, off_.p, ntask_[0].p, toff_[0].p, bsum_.p, meta_.p, stream, d_hmeta_));
    SG_TRY(hipEventRecord(ev_meta_, stream));
    uint32_t log_R = std::min<uint32_t>(cfg_.log_scatter_rounds, c - 1);
    msm_scatter<<<dim3(W, P, 1u << log_R), 1024, (nbw >> log_R) * sizeof(uint32_t), stream>>>(
        dig_.p, (uint32_t)n, chunk, nbw, hist_.p, off_.p, j.fixed ? W1 : 0u, j.n_tab, sorted_.p);
  }
  if (tm) SG_TRY(hipEventRecord(j.ev[2], stream));
  return hipGetLastError();
}

// ---- phase 2: needs the task count on the host; enqueues accumulate .. export + result copy
// jobs of this process between their first kernel and the end of their host tail (all engines, all lanes)
// (g_jobs_in_flight is defined with the accumulation chain, above)
// a caller that is about to run several jobs side by side on its own engines (the chunked host-pointer MSM) declares them
// "in flight" for the length of the call, so that the first of them already launches politely
void msm_hold_in_flight(bool on) { g_jobs_in_flight.fetch_add(on ? 1 : -1); }
void MsmEngine::mark_in_flight(bool on) {
  if (on == counted_) return;
  counted_ = on;
  g_jobs_in_flight.fetch_add(on ? 1 : -1);
}
bool MsmEngine::others_in_flight() const { return g_jobs_in_flight.load() > (counted_ ? 1 : 0); }
hipError_t MsmEngine::enqueue_back() {
  mark_in_flight(true);
  const hipError_t e = enqueue_back_impl();
  if (e != hipSuccess) mark_in_flight(false);
  return e;
}
hipError_t MsmEngine::finish() {
  const hipError_t e = finish_impl();
  mark_in_flight(false);
  return e;
}
hipError_t MsmEngine::enqueue_back_impl() {
  Job& j = job_;
  if (j.trivial) return hipSuccess;
  hipStream_t stream = j.stream;
  const uint32_t Wm = j.fixed ? 1u : j.wp.W;  // bucket sets ("windows") per MSM
  const uint32_t NB = j.NB, W = Wm * j.M, nbw = j.nbw, log_L = j.log_L;
  // tasks: sum_b ceil(cnt_b / L) <= (#non-empty buckets) + entries / L -- enough to size the task tables and the
  // accumulation launch without the counters; the host reads them (for the merge rounds) while that launch runs
  const uint32_t ntasks_ub = (uint32_t)(std::min<size_t>(NB, j.entries) + (j.entries >> log_L));
  // bucket b owns cur[toff_[lvl][b] .. +ntask_[lvl][b])
  // (fused front end: a bucket's task slots start at its coarse bin's base, with gaps at the end of every bin)
  SG_TRY(partial_[0].reserve(j.fe ? (size_t)NB + (j.entries >> log_L) + 1 : (size_t)ntasks_ub));
  {
    const uint32_t nbins = std::min<uint32_t>(1u << log_L, TASK_BINS - 1) + 1;  // task lengths 0 .. L
    const uint32_t tb = task_block_for(NB, nbins), tblk = (NB + tb - 1) / tb;
    SG_TRY(order_.reserve(ntasks_ub));
    if (j.fe) {
      // (no scan over bins x workgroups to keep small any more: more, smaller workgroups -- the pass is latency, not work)
      const uint32_t tb = NB >= (1u << 16) ? 512u : 256u, tblk = (NB + tb - 1) / tb;
      msm_task_scatter_reserve<<<tblk, 256, 0, stream>>>(
          counts_.p, NB, log_L, tb,
          FrontEndTotals{fe_.p, coff_.p, tbase_.p, off_.p, toff_[0].p, meta_.p, d_hmeta_, j.NBc, j.fe_parity, ACC_TICKET}, order_.p);
      SG_TRY(hipEventRecord(ev_meta_, stream));
    } else {
      SG_TRY(thist_.reserve((size_t)TASK_BINS * tblk));
      msm_task_hist<<<tblk, 256, 0, stream>>>(counts_.p, NB, log_L, tb, thist_.p);
      msm_task_scan<<<1, 256, 0, stream>>>(thist_.p, tblk, nbins, meta_.p + ACC_TICKET);
      msm_task_scatter<<<tblk, 256, 0, stream>>>(counts_.p, NB, log_L, tb, thist_.p, order_.p);
    }
  }
  const uint32_t at = cfg_.acc_threads ? cfg_.acc_threads : 128;  // measured: 128 beats 256 by 5 % at 2^20 (finer-grained tail), 64 loses in fixed mode
  // persistent launch: `waves` per SIMD on every CU (3 fill the register file)
  // ... three fill the register file (a job that has the device to itself); two leave a third of it to the kernels of other
  // streams, which run at wave priority 3 (side_kernel_prio): the other jobs in flight, a proof's transforms under its commitments
  // Round 5, fixed-base jobs: three waves when the job is several ROUNDS of tasks on a two-wave launch (the five dense quotient
  // pieces of a proof: 330 K tasks on 131 072 lanes) -- the counters show the two-wave launch issuing 61 % of the time where
  // three waves issue 87 %, and nothing runs beside that job (the evaluations wait for its challenge); a job of ONE round
  // (a single polynomial: W, W') keeps two: its time is the length of one task, which a third wave per SIMD only stretches
  // (k = 17 proof, phase 4 2.03 -> 1.90 ms, phase 6 1.20 -> 1.26 with three waves everywhere; profiles/r05_sweeps/accumulate_waves_fixed.txt)
  const uint32_t lanes2 = cus_ * 4u * 64u * 2u;
  const uint32_t waves_fixed_auto = ntasks_ub >= 3u * lanes2 ? 3u : 2u;
  const uint32_t waves = j.fixed ? (cfg_.acc_waves_fixed ? cfg_.acc_waves_fixed : waves_fixed_auto)
                                 : (cfg_.acc_waves ? cfg_.acc_waves : (others_in_flight() ? 2 : 3));
  const uint32_t wg_all = (ntasks_ub + at - 1) / at;
  const uint32_t wg = waves >= 8 ? wg_all : std::min<uint32_t>(wg_all, cus_ * (waves * 4 * 64 / at));
  j.acc_threads = wg * at;
  if (cfg_.acc_trace) SG_TRY(trace_.reserve((size_t)2 * wg * at / 64));
  SG_TRY(chained_accumulate(stream, j.tm ? j.ev[5] : nullptr, [&]() {
    msm_accumulate<<<wg, at, 0, stream>>>(sorted_.p, j.bp, Wm * nbw, off_.p, counts_.p, toff_[0].p, order_.p, log_L, meta_.p,
                                          meta_.p + ACC_TICKET, partial_[0].p, cfg_.acc_trace ? trace_.p : nullptr);
  }));
  SG_TRY(host_wait_event(ev_meta_));
  const volatile uint32_t* hm = h_meta_;   // written by the device (msm_scan_sums / msm_scan_small), complete with the event
  const uint32_t ntasks = j.ntasks = hm[1], max_cnt = j.max_cnt = hm[2];
  if (!ntasks) {  // every digit was zero (the launches above found nothing to do)
    j.all_zero = true;
    return hipSuccess;
  }
  const xyzz29_mem* cur = partial_[0].p;
  // quad-cooperative additions pay off while the reduction is a latency chain (few buckets in total);
  // with many windows it is throughput-bound and one lane per addition is the efficient shape
  const bool quad = cfg_.quad == 2 || (cfg_.quad == 1 && NB <= (1u << 18));  // measured crossover: tools/small_batches2.sh
  int lvl = 0, pbuf = 0;
  uint32_t items_ub = ntasks;  // upper bound of the number of partial sums alive at this level
  // 2-D reduction (below) or the scan-based one?  Decided here because the 2-D line sums can add a bucket's few partial sums
  // themselves: up to `fold` of them per bucket need no merge round (its launches -- three scans and the merge -- cost more
  // than the extra additions inside a launch that runs anyway)
  j.red2d = (cfg_.red2d && j.c >= 5) ? ((W <= cfg_.red2d_max_sets && Wm <= 4) ? 1u : (cfg_.red2d >= 2 ? 2u : 0u)) : 0u;
  const uint32_t fold = j.red2d ? cfg_.red2d_fold : 1u;
  // heavy buckets: fold their partial sums until every bucket owns at most `fold`
  for (uint32_t max_items = (max_cnt + (1u << log_L) - 1) >> log_L; max_items > fold;
       max_items = (max_items + (1u << log_L) - 1) >> log_L) {
    const int nxt = 1 - lvl;
    SG_TRY(launch_scan(ntask_[lvl].p, NB, log_L, nullptr, ntask_[nxt].p, toff_[nxt].p, bsum_.p, meta_.p, stream));
    // no host round trip: sum_b ceil(t_b / L) <= (#non-empty buckets) + items / L
    const uint32_t nt2 = std::min(NB, items_ub) + (items_ub >> log_L);
    items_ub = nt2;
    SG_TRY(partial_[1 - pbuf].reserve(nt2));
    if (quad && nt2 <= cfg_.merge_quad_tasks)
      msm_merge<4><<<(nt2 + 63) / 64, 256, 0, stream>>>(cur, toff_[lvl].p, ntask_[lvl].p, toff_[nxt].p, NB, log_L, meta_.p,
                                                        partial_[1 - pbuf].p);
    else
      msm_merge<1><<<(nt2 + 255) / 256, 256, 0, stream>>>(cur, toff_[lvl].p, ntask_[lvl].p, toff_[nxt].p, NB, log_L, meta_.p,
                                                          partial_[1 - pbuf].p);
    pbuf = 1 - pbuf;
    cur = partial_[pbuf].p;
    lvl = nxt;
  }
  if (j.tm) SG_TRY(hipEventRecord(j.ev[3], stream));
  if (tail_stream_) {
    // the latency-bound tail runs on a high-priority stream so that its few workgroups are
    // dispatched ahead of the next MSM's accumulation (batches ping-pong between two engines)
    if (!ev_acc_) SG_TRY(hipEventCreateWithFlags(&ev_acc_, hipEventDisableTiming));
    SG_TRY(hipEventRecord(ev_acc_, stream));
    SG_TRY(hipStreamWaitEvent(tail_stream_, ev_acc_, 0));
    stream = tail_stream_;
  }

  // 2-D reduction (rows / columns / bits): plain sums only, ~9 dependent additions per launch instead of a chain
  // of ~40.  Few sets: quad-cooperative additions and the powers of two on the host (bits + 1 points per set);
  // many sets: one lane per addition and a third launch that applies the powers of two (one point per set).
  // Measured (profiles/r01_sweeps): a clear win for up to 4 sets (k = 17 single commit: reduction 190 -> 90 us); with
  // many sets the tree sums waste lanes and the scan-based path below is faster, so the device-weights variant only
  // runs when forced (msm.red2d = 2).
  if (j.red2d) {
    const uint32_t bits = j.c - 1, sets = W;
    Reduce2dShape sh{(bits + 1) / 2, bits / 2};
    const uint32_t rows = 1u << sh.log_rows, cols = 1u << sh.log_cols;
    SG_TRY(red_a_[0].reserve((size_t)sets * (rows + cols)));
    SG_TRY(red_a_[1].reserve((size_t)sets * (bits + 1)));
    if (cfg_.red2d_prefold) {
      // partial sums -> one value per bucket (once), then line sums over plain arrays
      SG_TRY(red_r_[0].reserve(NB));
      if (quad && NB <= cfg_.prefold_quad_buckets)
        msm_fold_buckets<4><<<(NB + 63) / 64, 256, 0, stream>>>(cur, toff_[lvl].p, ntask_[lvl].p, NB, red_r_[0].p);
      else
        msm_fold_buckets<1><<<(NB + 255) / 256, 256, 0, stream>>>(cur, toff_[lvl].p, ntask_[lvl].p, NB, red_r_[0].p);
      if (quad) msm_reduce2d_lines_folded<4><<<dim3(rows + cols, sets), 256, 0, stream>>>(red_r_[0].p, sh, red_a_[0].p);
      else msm_reduce2d_lines_folded<1><<<dim3(rows + cols, sets), 256, 0, stream>>>(red_r_[0].p, sh, red_a_[0].p);
    } else if (quad) {
      msm_reduce2d_lines<4><<<dim3(rows + cols, sets), 256, 0, stream>>>(cur, toff_[lvl].p, ntask_[lvl].p, sh, red_a_[0].p);
    } else {
      msm_reduce2d_lines<1><<<dim3(rows + cols, sets), 256, 0, stream>>>(cur, toff_[lvl].p, ntask_[lvl].p, sh, red_a_[0].p);
    }
    if (quad) msm_reduce2d_bits<4><<<dim3(bits + 1, sets), 256, 0, stream>>>(red_a_[0].p, sh, red_a_[1].p);
    else msm_reduce2d_bits<1><<<dim3(bits + 1, sets), 256, 0, stream>>>(red_a_[0].p, sh, red_a_[1].p);
    const xyzz29_mem* fin = red_a_[1].p;
    uint32_t count = sets * (bits + 1);
    if (j.red2d == 2) {
      SG_TRY(red_s_[0].reserve(sets));
      msm_reduce2d_combine<<<sets, 32, 0, stream>>>(red_a_[1].p, bits, red_s_[0].p);
      fin = red_s_[0].p;
      count = sets;
    }
    if (j.tm) SG_TRY(hipEventRecord(j.ev[4], stream));
    msm_export_points<<<(count + 63) / 64, 64, 0, stream>>>(fin, count, d_hwin_);
    SG_TRY(hipEventRecord(ev_done_, stream));
    return hipGetLastError();
  }

  // bucket reduction: level 0 over the buckets, level 1 over the workgroup items.  Both are chains of
  // dependent point additions with most of the chip idle, so by default a point addition is spread
  // over the 4 lanes of a quad (cfg.quad; see `quad` above): 64 logical threads per workgroup.
  const uint32_t max_threads = quad ? 64u : cfg_.red_threads;       // logical threads per workgroup
  const uint32_t max_blocks = quad ? 64u : 256u;                    // level 1 holds 3 * T1 * Q <= 768 lanes
  // G buckets per logical thread: 8 for the largest windows, 4 below (depth vs. work, measured)
  // ... and 16 when other jobs are in flight: the reduction then runs under another job's accumulation, where what counts is
  // the instructions it issues (running sums are 2 additions per bucket, the scan and tree steps come per thread: 88 instead
  // of 116 wave-additions per 2048 buckets), not the length of its own chain (alone: 0.29 -> 0.39 ms; three MSMs in
  // flight: +0.7 % points/s, profiles/r04_sweeps/reduce_chunk_pipelined.txt)
  const uint32_t auto_log_G = nbw >= (1u << 14) ? ((!quad && others_in_flight()) ? 4u : 3u) : 2u;
  j.log_G = std::min<uint32_t>(cfg_.log_red_chunk ? cfg_.log_red_chunk : auto_log_G, j.c - 1);
  while ((nbw >> j.log_G) > max_threads * max_blocks) j.log_G++;
  const uint32_t items = nbw >> j.log_G;  // chunks per window at level 0 (a power of two)
  const uint32_t threads = std::min<uint32_t>(max_threads, std::max<uint32_t>(16, items));
  const uint32_t blocks = j.blocks = (items + threads - 1) / threads;
  if (blocks > max_blocks) return hipErrorInvalidValue;
  j.log_N = 0;
  while ((1u << j.log_N) < threads) j.log_N++;
  for (int i = 0; i < 2; i++) {
    SG_TRY(red_a_[i].reserve((size_t)W * blocks));
    SG_TRY(red_s_[i].reserve((size_t)W * blocks));
    SG_TRY(red_r_[i].reserve((size_t)W * blocks));
  }
  ReduceOut lvl0{red_a_[0].p, red_s_[0].p, red_r_[0].p}, lvl1{red_a_[1].p, red_s_[1].p, red_r_[1].p};
  if (quad)
    msm_reduce_buckets<4><<<dim3(blocks, W), threads * 4, (size_t)threads * 2 * sizeof(xyzz29_mem), stream>>>(
        cur, toff_[lvl].p, ntask_[lvl].p, nbw, j.log_G, lvl0);
  else if (cfg_.red_lean == 2 || (cfg_.red_lean == 1 && others_in_flight()))
    msm_reduce_buckets_lean<1><<<dim3(blocks, W), threads, (size_t)threads * 2 * sizeof(xyzz29_mem), stream>>>(
        cur, toff_[lvl].p, ntask_[lvl].p, nbw, j.log_G, lvl0);
  else
    msm_reduce_buckets<1><<<dim3(blocks, W), threads, (size_t)threads * 2 * sizeof(xyzz29_mem), stream>>>(
        cur, toff_[lvl].p, ntask_[lvl].p, nbw, j.log_G, lvl0);
  ReduceOut fin = lvl0;
  if (blocks > 1) {
    uint32_t T1 = 16;
    while (T1 < blocks) T1 <<= 1;
    // level 1 is a handful of items per window whatever the job: always a latency chain, so its additions are
    // quad-cooperative whenever the workgroup fits (3 * T1 * 4 lanes)
    if (quad || T1 <= 64)
      msm_reduce_items<4><<<dim3(1, W), 3 * T1 * 4, (size_t)3 * T1 * sizeof(xyzz29_mem), stream>>>(lvl0, blocks, T1, lvl1);
    else
      msm_reduce_items<1><<<dim3(1, W), 3 * T1, (size_t)3 * T1 * sizeof(xyzz29_mem), stream>>>(lvl0, blocks, T1, lvl1);
    fin = lvl1;
  }
  if (j.tm) SG_TRY(hipEventRecord(j.ev[4], stream));
  msm_export_windows<<<(3 * W + 63) / 64, 64, 0, stream>>>(fin, W, blocks > 1 ? 1u : 0u, d_hwin_);
  SG_TRY(hipEventRecord(ev_done_, stream));
  return hipGetLastError();
}

// ---- phase 3: wait for the window sums; host tail
hipError_t MsmEngine::finish_impl() {
  Job& j = job_;
  auto drop_events = [&]() {
    if (j.tm) {
      for (auto& e : j.ev) (void)hipEventDestroy(e);
    }
  };
  if (j.trivial) {
    std::memset(j.out, 0, 64 * std::max<size_t>(1, j.M));
    return hipSuccess;
  }
  if (j.all_zero) {
    std::memset(j.out, 0, 64 * (size_t)j.M);
    drop_events();
    return hipSuccess;
  }
  SG_TRY(host_wait_event(ev_done_));
  // window_j = A + 2^log_G (S + 2^log_N T); then Horner over the windows, high to low
  using namespace host;
  auto point_at = [&](uint32_t q) {
    Fq x, y, zz, zzz;
    std::memcpy(x.v, h_win_ + 32 * q, 32);
    std::memcpy(y.v, h_win_ + 32 * q + 8, 32);
    std::memcpy(zz.v, h_win_ + 32 * q + 16, 32);
    std::memcpy(zzz.v, h_win_ + 32 * q + 24, 32);
    return jac_from_xyzz(x, y, zz, zzz);
  };
  // window_w = A + 2^log_G (S + 2^log_N T) and result = sum_w 2^(offset_w) window_w: all 3W terms
  // are placed at their bit offsets and folded by ONE double-and-add sweep from the top bit
  // (254 + log_G + log_N doublings instead of W * (width + log_G + log_N))
  uint32_t offs[64];
  const uint32_t Wm = j.fixed ? 1u : j.wp.W;  // fixed-base: the table rows already carry the window offsets
  {
    uint32_t o = 0;
    for (uint32_t w = 0; w < Wm; w++) {
      offs[w] = o;
      o += j.wp.width[w];
    }
  }
  constexpr uint32_t MAXBIT = 254 + 16 + 16;
  Jac totals[MAX_FUSED];
  // terms per window: legacy (A, S, T) at offsets (0, log_G, log_G + log_N); 2-D with host weights: term t < bits
  // at offset t and the total at 0; 2-D with device weights: one term at 0
  const uint32_t bits2d = j.c - 1;
  const uint32_t per_win = j.red2d == 1 ? bits2d + 1 : j.red2d == 2 ? 1u : 3u;
  for (uint32_t m = 0; m < j.M; m++) {
    int head[MAXBIT + 1];
    int next[3 * 64];
    for (auto& h : head) h = -1;
    uint32_t top = 0;
    for (uint32_t w = 0; w < Wm; w++) {
      for (uint32_t which = 0; which < per_win; which++) {
        uint32_t rel;
        if (j.red2d == 1) rel = which < bits2d ? which : 0u;
        else if (j.red2d == 2) rel = 0;
        else rel = (which >= 1 ? j.log_G : 0) + (which == 2 ? j.log_N : 0);
        const uint32_t bit = offs[w] + rel;
        const int id = (int)(per_win * w + which);
        next[id] = head[bit];
        head[bit] = id;
        top = std::max(top, bit);
      }
    }
    Jac total = Jac::identity();
    for (int bit = (int)top; bit >= 0; bit--) {
      total = jac_double(total);
      for (int id = head[bit]; id >= 0; id = next[id]) total = jac_add(total, point_at(per_win * m * Wm + (uint32_t)id));
    }
    totals[m] = total;
  }
  // affine normalisation of the M results with ONE field inversion (Montgomery's trick): an inversion
  // is ~13 us on the host, as much as the rest of a fixed-base tail
  {
    Fq prefix[MAX_FUSED];
    Fq run = Fq::one();
    for (uint32_t m = 0; m < j.M; m++) {
      prefix[m] = run;
      if (!totals[m].is_identity()) run = run * totals[m].z;
    }
    Fq inv = run.inv();
    for (uint32_t m = j.M; m-- > 0;) {
      uint8_t* out = j.out + 64 * m;
      if (totals[m].is_identity()) {
        std::memset(out, 0, 64);
        continue;
      }
      const Fq zi = inv * prefix[m];
      inv = inv * totals[m].z;
      const Fq zi2 = zi.sqr();
      const Fq ax = totals[m].x * zi2, ay = totals[m].y * zi2 * zi;
      std::memcpy(out, ax.v, 32);
      std::memcpy(out + 32, ay.v, 32);
    }
  }

  if (cfg_.acc_trace && j.acc_threads) {
    // debug: when the waves of the accumulation left, as a share of the launch's span (first start .. last exit)
    const size_t waves = j.acc_threads / 64;
    std::vector<uint64_t> t(2 * waves);
    if (hipMemcpy(t.data(), trace_.p, t.size() * sizeof(uint64_t), hipMemcpyDeviceToHost) == hipSuccess) {
      uint64_t t0 = ~0ull, t1 = 0;
      for (size_t w = 0; w < waves; w++) { t0 = std::min(t0, t[2 * w]); t1 = std::max(t1, t[2 * w + 1]); }
      std::vector<double> ends(waves);
      size_t late = 0;
      const double span = (double)(t1 - t0) / 100.0;
      for (size_t w = 0; w < waves; w++) {
        ends[w] = (double)(t[2 * w + 1] - t0) / span;
        late += (double)(t[2 * w] - t0) / span > 5.0 ? 1 : 0;
      }
      std::sort(ends.begin(), ends.end());
      auto pct = [&](double q) { return ends[std::min(waves - 1, (size_t)(q * (double)waves))]; };
      std::fprintf(stderr, "[acc_trace] waves %zu (%zu started after 5 %% of the span), span %.0f ticks; 1 %% of the waves had left by %.1f %% of it, 10 %% by %.1f, 25 %% by %.1f, 50 %% by %.1f, 75 %% by %.1f, 90 %% by %.1f, 99 %% by %.1f\n",
                   waves, late, span * 100.0, pct(0.01), pct(0.10), pct(0.25), pct(0.50), pct(0.75), pct(0.90), pct(0.99));
    }
  }
  if (j.tm) {
    MsmTimings* tm = j.tm;
    float ms;
    (void)hipEventElapsedTime(&ms, j.ev[0], j.ev[1]); tm->digits_ms = ms;
    (void)hipEventElapsedTime(&ms, j.ev[1], j.ev[2]); tm->sort_ms = ms;
    (void)hipEventElapsedTime(&ms, j.ev[5], j.ev[3]); tm->accumulate_ms = ms;   // the accumulation (and merge rounds) alone
    (void)hipEventElapsedTime(&ms, j.ev[2], j.ev[5]); tm->order_ms = ms;        // task ordering + time queued behind other jobs' accumulations
    (void)hipEventElapsedTime(&ms, j.ev[3], j.ev[4]); tm->reduce_ms = ms;
    (void)hipEventElapsedTime(&ms, j.ev[0], j.ev[4]); tm->total_ms = ms;
    tm->window_bits = j.c;
    tm->windows = j.wp.W;
    tm->tasks = j.ntasks;
    tm->max_bucket = j.max_cnt;
    tm->accumulate_threads = j.acc_threads;
    drop_events();
  }
  return hipSuccess;
}

hipError_t MsmEngine::run(const fp_words* d_scalars, const g1_affine_mem* d_bases, size_t n, hipStream_t stream,
                          uint8_t out_affine[64], MsmTimings* tm) {
  SG_TRY(enqueue_front(d_scalars, d_bases, n, stream, out_affine, tm));
  SG_TRY(enqueue_back());
  return finish();
}

// ------------------------------------------------------------------ a handful of points: ONE launch
// The verifier's left-hand side is an MSM of 37 points (csrc/verifier_abi.hip), once per proof served: through the engine above it
// is eleven launches, three staging copies and two host waits for 2 368 point additions -- in a batch of proofs a third of all
// MSM launches.  n <= MSM_TINY_MAX goes through one kernel instead: workgroup = window (c = 4: 64 windows of 8 buckets, the
// engine's own window plan and digit rule), one wave each.  Lane i derives the digit of scalar i; lane b < 8 gathers bucket b + 1
// (a scan of the n digits in LDS: 4.6 mixed additions on average at n = 37); the weighted sum sum_b (b + 1) B_b is a suffix scan
// over the eight lanes and a tree sum of the suffix sums (6 dependent additions instead of 16 running-sum steps); lane 0 writes the
// window's sum, canonical, into mapped host memory.  Scalars and points are READ from mapped host memory (6 KB: no staging
// copy); the host tail is the engine's (terms at their bit offsets, one double-and-add sweep).
__global__ void __launch_bounds__(64) msm_tiny_kernel(const fp_words* __restrict__ scalars, const g1_affine_mem* __restrict__ bases,
                                                      uint32_t n, WindowPlan wp, uint32_t* __restrict__ out_words) {
  side_kernel_prio();
  __shared__ int s_dig[MSM_TINY_MAX];
  __shared__ uint32_t s_pt[MSM_TINY_MAX][16];
  __shared__ xyzz29_mem s_x[8];
  const uint32_t w = blockIdx.x, t = threadIdx.x, W = wp.W;
  if (t < n) {
    words8 s;
    {
      f29 k = f29_zero();   // canonical scalar = s~ * 2^5 * 2^-261 (msm_digits)
      k.l[0] = 32;
      f29_to_words(f29_cond_sub_p<Fr29>(f29_mul<Fr29>(f29_load_r256<Fr29>(scalars + t), k)), s.l);
    }
    uint32_t off = 0;
    {
      // s += K = sum_{j < W-1} 2^(o_j + w_j - 1): every window's digit becomes independent of its neighbours
      uint32_t kk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      uint32_t o = 0;
      for (uint32_t j = 0; j + 1 < W; j++) {
        const uint32_t bit = o + wp.width[j] - 1;
        const uint32_t m = 1u << (bit & 31), q = bit >> 5;
#pragma unroll
        for (int i = 0; i < 8; i++) kk[i] |= (q == (uint32_t)i) ? m : 0u;
        if (j < w) off += wp.width[j];
        o += wp.width[j];
      }
      uint32_t carry = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint64_t v = (uint64_t)s.l[q] + kk[q] + carry;
        s.l[q] = (uint32_t)v;
        carry = (uint32_t)(v >> 32);
      }
    }
    const uint32_t width = wp.width[w], q = off >> 5, r = off & 31;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      lo = (q == (uint32_t)i) ? s.l[i] : lo;
      hi = (q + 1 == (uint32_t)i) ? s.l[i] : hi;
    }
    const uint32_t v = (uint32_t)((((uint64_t)hi << 32) | lo) >> r) & ((1u << width) - 1);
    s_dig[t] = (w + 1 < W) ? (int)v - (int)(1u << (width - 1)) : (int)v;
    const uint4* src = bases[t].q;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint4 x = src[i];
      s_pt[t][4 * i] = x.x; s_pt[t][4 * i + 1] = x.y; s_pt[t][4 * i + 2] = x.z; s_pt[t][4 * i + 3] = x.w;
    }
  }
  __syncthreads();
  xyzz29 acc = xyzz29_identity();
  if (t < 8) {
    for (uint32_t i = 0; i < n; i++) {
      const int d = s_dig[i];
      if ((d < 0 ? -d : d) != (int)t + 1) continue;
      affine29 p = affine29_from_words(s_pt[i]);
      if (d < 0) affine29_negate(p);
      xyzz29_madd(acc, p);
    }
  }
  // R_b = sum_{b' >= b} B_b' (three steps), then sum_b R_b = sum_b (b + 1) B_b (three steps)
  for (uint32_t step = 1; step < 8; step <<= 1) {
    if (t < 8) xyzz29_store(s_x + t, acc);
    __syncthreads();
    if (t + step < 8) xyzz29_add(acc, xyzz29_load(s_x + t + step));
    __syncthreads();
  }
  for (uint32_t step = 4; step >= 1; step >>= 1) {
    if (t < 8) xyzz29_store(s_x + t, acc);
    __syncthreads();
    if (t < step) xyzz29_add(acc, xyzz29_load(s_x + t + step));
    __syncthreads();
  }
  if (t == 0) {
    uint32_t wd[32];
    xyzz29_to_words(acc, wd);
#pragma unroll
    for (int i = 0; i < 32; i++) out_words[32 * w + i] = wd[i];
  }
}

hipError_t MsmEngine::run_tiny(const uint8_t* h_scalars, const uint8_t* h_bases, size_t n, hipStream_t stream, uint8_t out_affine[64]) {
  if (n == 0) {
    std::memset(out_affine, 0, 64);
    return hipSuccess;
  }
  if (n > MSM_TINY_MAX) return hipErrorInvalidValue;
  constexpr size_t IN_BYTES = MSM_TINY_MAX * (32 + 64), OUT_WORDS = 64 * 32;
  if (!h_tiny_) {
    SG_TRY(hipHostMalloc(&h_tiny_, IN_BYTES + OUT_WORDS * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
    SG_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_tiny_), h_tiny_, 0));
  }
  if (!ev_tiny_) SG_TRY(hipEventCreateWithFlags(&ev_tiny_, hipEventDisableTiming));
  // (the buffer is this engine's, and the previous call waited for its kernel: nothing reads it now)
  std::memcpy(h_tiny_, h_scalars, 32 * n);
  std::memcpy(h_tiny_ + 32 * MSM_TINY_MAX, h_bases, 64 * n);
  const WindowPlan wp = make_window_plan(4);
  uint32_t* h_out = reinterpret_cast<uint32_t*>(h_tiny_ + IN_BYTES);
  msm_tiny_kernel<<<wp.W, 64, 0, stream>>>(reinterpret_cast<const fp_words*>(d_tiny_),
                                           reinterpret_cast<const g1_affine_mem*>(d_tiny_ + 32 * MSM_TINY_MAX), (uint32_t)n, wp,
                                           reinterpret_cast<uint32_t*>(d_tiny_ + IN_BYTES));
  SG_TRY(hipGetLastError());
  SG_TRY(hipEventRecord(ev_tiny_, stream));
  SG_TRY(host_wait_event(ev_tiny_));
  using namespace host;
  // window sums at their bit offsets, one double-and-add sweep from the top bit (as finish() does)
  int head[255], next[64];
  for (auto& h : head) h = -1;
  uint32_t o = 0, top = 0;
  for (uint32_t w = 0; w < wp.W; w++) {
    next[w] = head[o];
    head[o] = (int)w;
    top = o;
    o += wp.width[w];
  }
  Jac total = Jac::identity();
  for (int bit = (int)top; bit >= 0; bit--) {
    total = jac_double(total);
    for (int id = head[bit]; id >= 0; id = next[id]) {
      Fq x, y, zz, zzz;
      std::memcpy(x.v, h_out + 32 * id, 32);
      std::memcpy(y.v, h_out + 32 * id + 8, 32);
      std::memcpy(zz.v, h_out + 32 * id + 16, 32);
      std::memcpy(zzz.v, h_out + 32 * id + 24, 32);
      total = jac_add(total, jac_from_xyzz(x, y, zz, zzz));
    }
  }
  jac_to_affine_bytes(total, out_affine);
  return hipSuccess;
}

// points[i] on y^2 = x^3 + 3 (or the identity, 64 zero bytes)?  *bad counts the points that are not
__global__ void __launch_bounds__(256) g1_on_curve_kernel(const g1_affine_mem* __restrict__ points, uint32_t n, uint32_t* bad) {
  typedef Fq29 P;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const affine29 p = affine29_load(points + i);
  if (p.inf) return;
  const f29 one = f29_one<P>();
  const f29 x = f29_mul<P>(p.x, one), y = f29_mul<P>(p.y, one);       // bound 32 -> < 2
  const f29 x3 = f29_mul<P>(f29_sqr<P>(x), x), y2 = f29_sqr<P>(y);
  const f29 three = f29_add(f29_add(one, one), one);
  const f29 d = f29_sub<P, 1>(f29_sub<P, 0>(y2, x3), three);          // y^2 - x^3 - 3 (+ multiples of q)
  if (!f29_is_zero_mod_p<P>(d)) atomicAdd(bad, 1u);
}
hipError_t g1_on_curve(const g1_affine_mem* d_points, size_t n, uint32_t* d_bad, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(uint32_t), stream);
  if (e != hipSuccess || !n) return e;
  g1_on_curve_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d_points, (uint32_t)n, d_bad);
  return hipGetLastError();
}

hipError_t fixed_base_mul(const fp_words* d_scalars, size_t n, g1_affine_mem* d_out, hipStream_t stream) {
  if (!n) return hipSuccess;
  g1_fixed_base_mul<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d_scalars, (uint32_t)n, d_out);
  return hipGetLastError();
}

}  // namespace sg
