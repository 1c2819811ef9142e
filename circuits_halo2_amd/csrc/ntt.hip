// Radix-2 NTT / iNTT over BN254 Fr for gfx950 -- the GPU side of halo2's `best_fft`
// and `EvaluationDomain::{ifft, coeff_to_extended, extended_to_coeff}` (SURVEY.md §8a N1-N4;
// reference call sites zk_prover/src/circuits/utils.rs:75,76,94-101).
//
// Layout: natural order in, natural order out, elements are 32-byte Montgomery Fr exactly
// as halo2curves stores them.  A transform of n = 2^L points is factored into 1-3 passes
// (four-step / six-step Cooley-Tukey).  Every pass is one launch of `ntt_pass`: a workgroup
// stages a tile of T contiguous columns x R strided rows in LDS, runs log2(R) radix-2 DIT
// stages there (one butterfly per thread per stage, twiddles = powers of omega_R from an LDS
// tile), multiplies by the inter-pass twiddle (table in the pass's output order, fully
// coalesced) and writes the tile back.  The first pass of a multi-pass plan writes each
// column as a contiguous run ("transposing" store) so that no bit-reversal or separate
// transpose pass ever touches HBM.
//
// HBM traffic per pass: 32 B read + 32 B write per element (+32 B twiddle on twiddled
// passes).  Arithmetic: (n/2) log2 n Montgomery products + one per element per twiddled pass.
#include "ntt.h"
#include "host_wait.h"
#include "side_prio.cuh"

#include <algorithm>
#include <cstring>

namespace sg {
SG_DEFINE_SIDE_PRIO_SETTER(ntt_set_side_prio)

// ------------------------------------------------------------------ device kernels
// Arithmetic: 9 x 29-bit limbs (bn254_f29.cuh).  Data stays in the Montgomery-2^256 domain it
// is stored in; twiddle tables hold w^ = w * 2^261 mod r (as canonical 8 x u32 words), so
// f29_mul(x~, w^) = (x w)~.  Inside a pass additions are lazy (bound grows by 2 per stage,
// <= 25 after 10 stages); every pass ends with one product per element (inter-pass twiddle,
// post-scale, or 1^) which brings the value below 2r, then a conditional subtraction makes it
// canonical for the 32-byte store.
struct PassArgs {
  const fp_words* in;
  fp_words* out;
  const fp_words* tw_local;  // w_R^k in the 2^261 domain, k < R/2
  const fp_words* tw_pass;   // per-element twiddle in output order (nullptr: none)
  uint32_t log_r;            // log2 R  (DFT length of this pass)
  uint32_t log_t;            // log2 T  (contiguous columns per tile)
  uint32_t log_b;            // log2 B  (contiguous inner extent)
  uint32_t kind;             // 0: in-place-like (Y), 1: transposing first pass (X)
  uint32_t sig_lo;           // X only: b = lo + 2^sig_lo * hi  ->  b' = hi + 2^sig_hi * lo
  uint32_t sig_hi;
  uint32_t in_len;           // elements present in `in`; beyond that the input reads as zero
  uint32_t pre3;             // multiply input i by pre[i % 3]   (coeff_to_extended)
  uint32_t post3;            // multiply output j by post[j % 3] (extended_to_coeff / plain scale)
  uint32_t fold29;           // the data of this (last) pass carries a factor 2^29 (the plan's last twiddle table holds
                             // it): close with one Montgomery limb step instead of a product by 1^; post[] absorbs 2^-29
  uint32_t skip;             // leading DIT stages whose second operand is zero (zero-padded first pass): not executed
  uint32_t radix4;           // two DIT stages per sweep over the tile (four elements per thread)
  uint32_t pre[3][8];        // Montgomery-2^256 words, converted once per workgroup
  uint32_t post[3][8];
  // batched launch (nbatch > 0): blockIdx.y selects the vector; same plan for all (small transforms are a chain
  // of launches at their ~5 us floor: the 9 iNTTs of a proof become 3 launches instead of 27)
  uint32_t nbatch;
  const fp_words* in_b[NTT_BATCH_MAX];
  fp_words* out_b[NTT_BATCH_MAX];
  // first pass of a batched launch: input element gi of vector y is multiplied by pre_tab_b[y][gi] (2^261-domain words) on load
  uint32_t pre_tab;
  const fp_words* pre_tab_b[NTT_BATCH_MAX];
};

struct LdsTile {
  uint4* lo;
  uint4* hi;
  uint32_t* top;
};
__device__ __forceinline__ void lds_put(const LdsTile& t, uint32_t i, const f29& v) {
  t.lo[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  t.hi[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
  t.top[i] = v.l[8];
}
__device__ __forceinline__ f29 lds_get(const LdsTile& t, uint32_t i) {
  uint4 a = t.lo[i], b = t.hi[i];
  f29 r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = t.top[i];
  return r;
}

// One pass.  grid.x = number of tiles = (2^log_b / T) * A  where A = n / (B*R).
// Dynamic LDS: (T*R + R/2 + 8) * 36 bytes.
__global__ void __launch_bounds__(1024) ntt_pass(PassArgs p) {
#define SG_NTT_R4 0
#include "ntt_pass_body.inc"
#undef SG_NTT_R4
}
// the same pass with two DIT stages per sweep (four elements per thread: more registers, half the threads)
__global__ void __launch_bounds__(512) ntt_pass_r4(PassArgs p) {
#define SG_NTT_R4 1
#include "ntt_pass_body.inc"
#undef SG_NTT_R4
}

// tw[k] = (w^k)^ for k < count; w given as Montgomery-2^256 words
__global__ void fill_powers(fp_words* tw, words8 w, uint32_t count) {
  side_kernel_prio();
  typedef Fr29 P;
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < count) f29_store_canonical<P>(tw + k, f29_pow_u64<P>(f29_words_to_r261<P>(w.l), k));
}

// inter-pass twiddle tables, in the output order of the pass they are applied in
//  mode 0 (2-pass, pass X):   idx = j2 + n2*i1            -> w^(i1*j2) * scale
//  mode 1 (3-pass, pass A):   idx = j3 + n3*(i2 + n2*i1)  -> w^(n1*i2*j3) * scale
//  mode 2 (3-pass, pass B):   idx = j3 + n3*j2 + n2n3*i1  -> w^(i1*(j3 + n3*j2))
//  times29: the table additionally carries the factor 2^29 that the last pass removes with f29_mont_step
__global__ void fill_pass_twiddles(fp_words* tw, words8 w, words8 scale, uint32_t has_scale, uint32_t mode,
                                   uint32_t l1, uint32_t l2, uint32_t l3, uint32_t log_n, uint32_t times29) {
  side_kernel_prio();
  typedef Fr29 P;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >> log_n) return;
  uint64_t e;
  if (mode == 0) {
    uint64_t j2 = idx & ((1ull << l2) - 1), i1 = idx >> l2;
    e = i1 * j2;
  } else if (mode == 1) {
    uint64_t j3 = idx & ((1ull << l3) - 1), i2 = (idx >> l3) & ((1ull << l2) - 1);
    e = (i2 * j3) << l1;
  } else {
    uint64_t jj = idx & ((1ull << (l2 + l3)) - 1), i1 = idx >> (l2 + l3);
    e = i1 * jj;
  }
  e &= (1ull << log_n) - 1;
  f29 v = f29_pow_u64<P>(f29_words_to_r261<P>(w.l), e);
  if (has_scale) v = f29_mul<P>(v, f29_words_to_r261<P>(scale.l));
  if (times29) {   // v * (2^29 * 2^261) * 2^-261: the integer 2^290 mod r as limbs = (2^261 mod r) doubled 29 times
    f29 k = f29_one<P>();
    for (int i = 0; i < 29; i++) k = f29_cond_sub_p<P>(f29_normalize(f29_dbl(k)));
    v = f29_mul<P>(v, k);
  }
  f29_store_canonical<P>(tw + idx, v);
}
// out = w^e as Montgomery-2^256 words (host-visible domain constants)
__global__ void pow_single(fp_words* out, words8 w, uint64_t e) {
  side_kernel_prio();
  typedef Fr29 P;
  f29 v = f29_pow_u64<P>(f29_words_to_r261<P>(w.l), e);
  uint32_t o[8];
  f29_to_words(f29_reduce_with<P>(v, P::r256), o);
  fp_words_store(out, o);
}

__global__ void scale_kernel(fp_words* a, words8 s, size_t n) {
  side_kernel_prio();
  typedef Fr29 P;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f29_store_canonical<P>(a + i, f29_mul<P>(f29_load_r256<P>(a + i), f29_words_to_r261<P>(s.l)));
}
// a[i] *= tab[i & (period-1)]   (divide_by_vanishing_poly; period = 2^(ext_k-k)); tab in the
// 2^256 domain like the data
__global__ void scale_periodic_kernel(fp_words* a, const fp_words* tab, uint32_t period, size_t n) {
  side_kernel_prio();
  typedef Fr29 P;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    uint32_t t[8];
    fp_words_load(tab + (i & (period - 1)), t);
    f29_store_canonical<P>(a + i, f29_mul<P>(f29_load_r256<P>(a + i), f29_words_to_r261<P>(t)));
  }
}
// dir = 1: canonical integers -> Montgomery-2^256 (x * 2^517 * 2^-261); dir = 0: the inverse
// (x~ * 2^5 * 2^-261)
__global__ void to_mont_kernel(const fp_words* in, fp_words* out, size_t n, int dir) {
  side_kernel_prio();
  typedef Fr29 P;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    f29 k = f29_zero();
    k.l[0] = 32;
    if (dir) k = f29_const<P>(P::r517);
    f29_store_canonical<P>(out + i, f29_mul<P>(f29_load_r256<P>(in + i), k));
  }
}

// ------------------------------------------------------------------ host side
static bool fp_host_eq(const words8& a, const words8& b) { return std::memcmp(&a, &b, sizeof(words8)) == 0; }

NttEngine::~NttEngine() { clear(); }

void NttEngine::clear() {
  for (auto& pl : plans_) {
    for (int i = 0; i < 3; i++) {
      if (pl.tw_pass[i]) (void)hipFree(pl.tw_pass[i]);
    }
  }
  plans_.clear();
  for (auto& kv : local_tw_) (void)hipFree(kv.tw);
  local_tw_.clear();
}

hipError_t NttEngine::local_twiddles(const words8& omega_r, uint32_t log_r, hipStream_t stream, fp_words** out) {
  for (auto& e : local_tw_) {
    if (e.log_r == log_r && fp_host_eq(e.omega_r, omega_r)) {
      *out = e.tw;
      return hipSuccess;
    }
  }
  uint32_t count = log_r ? (1u << (log_r - 1)) : 1;
  fp_words* d = nullptr;
  hipError_t err = hipMalloc(&d, sizeof(fp_words) * count);
  if (err != hipSuccess) return err;
  fill_powers<<<(count + 255) / 256, 256, 0, stream>>>(d, omega_r, count);
  err = hipGetLastError();
  if (err == hipSuccess) err = host_wait_stream(stream);  // complete before any other stream may use the cached table
  if (err != hipSuccess) {
    retire_device_memory(d);
    return err;
  }
  local_tw_.push_back({log_r, omega_r, d});
  *out = d;
  return hipSuccess;
}

// choose the pass factorisation for a 2^log_n transform
static void factor(uint32_t log_n, uint32_t max_single, uint32_t max_multi, int* npass, uint32_t l[3]) {
  l[0] = l[1] = l[2] = 0;
  if (log_n <= max_single) {
    *npass = 1;
    l[0] = log_n;
  } else if (log_n <= 2 * max_multi) {
    *npass = 2;
    l[0] = (log_n + 1) / 2;  // n1 (second pass DFT length)
    l[1] = log_n - l[0];     // n2 (first pass DFT length)
  } else {
    *npass = 3;
    l[0] = (log_n + 2) / 3;
    l[1] = (log_n - l[0] + 1) / 2;
    l[2] = log_n - l[0] - l[1];
  }
}

hipError_t NttEngine::get_plan(uint32_t log_n, const words8& omega, const words8* scale, hipStream_t stream,
                               const NttPlan** out) {
  for (auto& pl : plans_) {
    if (pl.log_n == log_n && fp_host_eq(pl.omega, omega) && pl.has_scale == (scale != nullptr) &&
        (!scale || fp_host_eq(pl.scale, *scale))) {
      *out = &pl;
      return hipSuccess;
    }
  }
  NttPlan pl{};
  pl.log_n = log_n;
  pl.omega = omega;
  pl.has_scale = scale != nullptr;
  if (scale) pl.scale = *scale;
  factor(log_n, cfg_.max_single_log, cfg_.max_multi_log, &pl.npass, pl.l);
  const size_t n = (size_t)1 << log_n;
  hipError_t err;
  // omega_R for a length-R sub-transform is omega^(n/R): square omega (log_n - log_r) times
  // on the device via fill_powers' pow (host has no field arithmetic on purpose)
  auto omega_pow2 = [&](uint32_t times, words8* res) -> hipError_t {
    fp_words* d = nullptr;
    hipError_t e = hipMalloc(&d, sizeof(fp_words) * 2);
    if (e != hipSuccess) return e;
    // fill_powers computes w^k; k = 2^times fits 32 bits (times <= 28)
    pow_single<<<1, 1, 0, stream>>>(d, omega, 1ull << times);
    e = hipMemcpyAsync(res, d, sizeof(fp_words), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = host_wait_stream(stream);
    retire_device_memory(d);   // a few bytes per plan; hipFree would wait for every other stream of the device
    return e;
  };
  for (int i = 0; i < pl.npass; i++) {
    words8 wr;
    err = omega_pow2(log_n - pl.l[i], &wr);
    if (err != hipSuccess) return err;
    err = local_twiddles(wr, pl.l[i], stream, &pl.tw_local[i]);
    if (err != hipSuccess) return err;
  }
  words8 one_or_scale = scale ? *scale : omega;  // placeholder when unused
  if (pl.npass == 2) {
    err = hipMalloc(&pl.tw_pass[0], sizeof(fp_words) * n);
    if (err != hipSuccess) return err;
    fill_pass_twiddles<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(pl.tw_pass[0], omega, one_or_scale,
                                                                         scale ? 1 : 0, 0, pl.l[0], pl.l[1], 0, log_n, 1);
  } else if (pl.npass == 3) {
    err = hipMalloc(&pl.tw_pass[0], sizeof(fp_words) * n);
    if (err != hipSuccess) return err;
    err = hipMalloc(&pl.tw_pass[1], sizeof(fp_words) * n);
    if (err != hipSuccess) return err;
    fill_pass_twiddles<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(pl.tw_pass[0], omega, one_or_scale,
                                                                         scale ? 1 : 0, 1, pl.l[0], pl.l[1], pl.l[2], log_n, 0);
    fill_pass_twiddles<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(pl.tw_pass[1], omega, one_or_scale, 0, 2,
                                                                         pl.l[0], pl.l[1], pl.l[2], log_n, 1);
  }
  err = hipGetLastError();
  if (err != hipSuccess) return err;
  // tables must be complete before any other stream may use the cached plan
  err = host_wait_stream(stream);
  if (err != hipSuccess) return err;
  plans_.push_back(pl);
  *out = &plans_.back();
  return hipSuccess;
}


// leading stages of a zero-padded FIRST pass that can be skipped: rows r >= in_len >> log_b are zero
static uint32_t skippable_stages(const PassArgs& a, uint32_t log_n) {
  if (a.kind != 1 || a.in_len == 0 || (a.in_len & (a.in_len - 1)) || a.in_len >= (1u << log_n)) return 0;
  const uint32_t rows = a.in_len >> a.log_b;   // nonzero rows of every column (in_len is a power of two)
  if (rows == 0 || rows >= (1u << a.log_r)) return 0;
  uint32_t z = 0;
  while ((1u << z) < rows) z++;
  return a.log_r - z;
}

static hipError_t launch_pass(const NttConfig& cfg, PassArgs& a, uint32_t log_n, hipStream_t stream) {
  a.skip = skippable_stages(a, log_n);
  // tile width: as many contiguous columns as the LDS budget allows
  const bool big = (a.nbatch >= cfg.batch_min || log_n >= cfg.big_log) && cfg.big_tile_log;
  const uint32_t tile_log = big ? cfg.big_tile_log : cfg.tile_log, max_threads = big ? cfg.big_threads : cfg.threads;
  uint32_t log_e = std::min<uint32_t>(tile_log, log_n);
  if (log_e < a.log_r) log_e = a.log_r;
  uint32_t log_t = std::min<uint32_t>(log_e - a.log_r, a.log_b);
  a.log_t = log_t;
  size_t E = (size_t)1 << (a.log_r + log_t);
  size_t lds = (E + ((size_t)1 << a.log_r) / 2 + 8) * 36 + 64;
  // two stages per sweep: measured (profiles/r05_sweeps/ntt_radix4.txt) 3-6 % faster for the throughput shapes -- lone transforms
  // of 2^20 points and more, the batched launches of a proof (25 coset blocks: 323 -> 310 us) --, 25 % slower for a lone 2^17
  // transform (two launches at their latency floor, which want many threads)
  const bool want_r4 = cfg.radix4 == 2 || (cfg.radix4 == 1 && big);
  a.radix4 = want_r4 && a.log_r - a.skip >= 2 ? 1u : 0u;
  // one butterfly (radix-4 sweeps: one group of four elements) per thread per sweep
  uint32_t threads = (uint32_t)std::min<size_t>(a.radix4 ? std::min<uint32_t>(max_threads, 512u) : max_threads, std::max<size_t>(64, E / (a.radix4 ? 4 : 2)));
  uint32_t tiles = 1u << (log_n - a.log_r - log_t);
  if (a.radix4) hipLaunchKernelGGL(ntt_pass_r4, dim3(tiles, a.nbatch ? a.nbatch : 1), dim3(threads), lds, stream, a);
  else hipLaunchKernelGGL(ntt_pass, dim3(tiles, a.nbatch ? a.nbatch : 1), dim3(threads), lds, stream, a);
  return hipGetLastError();
}

// `count` in-place transforms of one size as ONE launch per pass.  scratch: count * 2^log_n elements (multi-pass plans)
hipError_t NttEngine::transform_batch(fp_words* const* a, uint32_t count, fp_words* scratch, uint32_t log_n,
                                      const words8& omega, const words8* scale, hipStream_t stream,
                                      const fp_words* const* src, size_t src_len, const words8* pre3, const fp_words* const* pre_tab) {
  if (count == 0) return hipSuccess;
  if (pre_tab && (!src || src_len < ((size_t)1 << log_n))) return hipErrorInvalidValue;   // (the zero-padded load path does not take a table)
  if (count > NTT_BATCH_MAX || log_n == 0) return hipErrorInvalidValue;
  const NttPlan* pl;
  const bool fold_scale = scale && log_n > cfg_.max_single_log;
  hipError_t err = get_plan(log_n, omega, fold_scale ? scale : nullptr, stream, &pl);
  if (err != hipSuccess) return err;
  const size_t n = (size_t)1 << log_n;
  // src == nullptr: in place (multi-pass plans go through `scratch`); otherwise out of place from src[i]
  // (src_len <= n elements, zero beyond; optional pre-scaling by pre3[i % 3]) with a[i] as the intermediate
  PassArgs p{};
  p.nbatch = count;
  enum Where { DATA, MID };
  bool first = true;
  auto io = [&](Where in, Where out) {
    for (uint32_t i = 0; i < count; i++) {
      fp_words* mid = src ? a[i] : scratch + (size_t)i * n;
      p.in_b[i] = (first && src) ? src[i] : in == DATA ? a[i] : mid;
      p.out_b[i] = out == DATA ? a[i] : mid;
    }
    p.in_len = (first && src) ? (uint32_t)std::min(src_len, n) : (uint32_t)n;
    p.pre3 = 0;
    if (first && pre3) {
      p.pre3 = 1;
      for (int i = 0; i < 3; i++) std::memcpy(p.pre[i], pre3[i].l, 32);
    }
    p.pre_tab = 0;
    if (first && pre_tab) {
      p.pre_tab = 1;
      for (uint32_t i = 0; i < count; i++) p.pre_tab_b[i] = pre_tab[i];
    }
    first = false;
  };
  auto last_scale = [&](bool last) {
    p.post3 = 0;
    if (last && scale && !fold_scale) {
      p.post3 = 1;
      for (int i = 0; i < 3; i++) std::memcpy(p.post[i], scale->l, 32);
    }
  };
  if (pl->npass == 1) {
    io(DATA, DATA);
    p.tw_local = pl->tw_local[0]; p.tw_pass = nullptr; p.log_r = log_n; p.log_b = 0; p.kind = 0;
    last_scale(true);
    return launch_pass(cfg_, p, log_n, stream);
  }
  if (pl->npass == 2) {
    const uint32_t l1 = pl->l[0], l2 = pl->l[1];
    io(DATA, MID);
    p.tw_local = pl->tw_local[1]; p.tw_pass = pl->tw_pass[0]; p.log_r = l2; p.log_b = l1; p.kind = 1; p.sig_lo = l1; p.sig_hi = 0;
    last_scale(false);
    err = launch_pass(cfg_, p, log_n, stream);
    if (err != hipSuccess) return err;
    io(MID, DATA);
    p.tw_local = pl->tw_local[0]; p.tw_pass = nullptr; p.log_r = l1; p.log_b = l2; p.kind = 0; p.fold29 = 1;
    last_scale(true);
    return launch_pass(cfg_, p, log_n, stream);
  }
  const uint32_t l1 = pl->l[0], l2 = pl->l[1], l3 = pl->l[2];
  io(DATA, MID);
  p.tw_local = pl->tw_local[2]; p.tw_pass = pl->tw_pass[0]; p.log_r = l3; p.log_b = l1 + l2; p.kind = 1; p.sig_lo = l1; p.sig_hi = l2;
  last_scale(false);
  err = launch_pass(cfg_, p, log_n, stream);
  if (err != hipSuccess) return err;
  io(MID, MID);
  p.tw_local = pl->tw_local[1]; p.tw_pass = pl->tw_pass[1]; p.log_r = l2; p.log_b = l3; p.kind = 0;
  err = launch_pass(cfg_, p, log_n, stream);
  if (err != hipSuccess) return err;
  io(MID, DATA);
  p.tw_local = pl->tw_local[0]; p.tw_pass = nullptr; p.log_r = l1; p.log_b = l2 + l3; p.kind = 0; p.fold29 = 1;
  last_scale(true);
  return launch_pass(cfg_, p, log_n, stream);
}

hipError_t NttEngine::init() {
  // allow the large dynamic-LDS tiles (up to the full 160 KiB of a CU)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_pass), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_pass_r4), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return e;
}

hipError_t NttEngine::transform(const fp_words* in, size_t in_len, fp_words* out, fp_words* scratch, uint32_t log_n,
                                const words8& omega, const words8* scale, const words8* pre3, const words8* post3,
                                hipStream_t stream) {
  const NttPlan* pl;
  // a plain scale is folded into the first inter-pass twiddle table when there is one
  bool fold_scale = scale && !post3 && log_n > cfg_.max_single_log;
  hipError_t err = get_plan(log_n, omega, fold_scale ? scale : nullptr, stream, &pl);
  if (err != hipSuccess) return err;
  if (log_n == 0) {
    if (in != out) err = hipMemcpyAsync(out, in, sizeof(fp_words), hipMemcpyDeviceToDevice, stream);
    if (err == hipSuccess && scale && !post3) scale_kernel<<<1, 64, 0, stream>>>(out, *scale, 1);
    return err;
  }
  PassArgs a{};
  auto set_prepost = [&](bool first, bool last) {
    a.pre3 = 0;
    a.post3 = 0;
    if (first && pre3) {
      a.pre3 = 1;
      for (int i = 0; i < 3; i++) std::memcpy(a.pre[i], pre3[i].l, 32);
    }
    if (last && post3) {
      a.post3 = 1;
      for (int i = 0; i < 3; i++) std::memcpy(a.post[i], post3[i].l, 32);
    } else if (last && scale && !fold_scale) {
      a.post3 = 1;
      for (int i = 0; i < 3; i++) std::memcpy(a.post[i], scale->l, 32);
    }
  };
  const bool inplace = (in == out);
  if (pl->npass == 1) {
    a.in = in; a.out = out; a.tw_local = pl->tw_local[0]; a.tw_pass = nullptr;
    a.log_r = log_n; a.log_b = 0; a.kind = 0; a.in_len = (uint32_t)std::min<size_t>(in_len, (size_t)1 << log_n);
    set_prepost(true, true);
    return launch_pass(cfg_, a, log_n, stream);
  }
  if (pl->npass == 2) {
    const uint32_t l1 = pl->l[0], l2 = pl->l[1];
    fp_words* mid = inplace ? scratch : out;
    // pass X: DFT over i2 (length n2, stride n1), columns i1 contiguous
    a.in = in; a.out = mid; a.tw_local = pl->tw_local[1]; a.tw_pass = pl->tw_pass[0];
    a.log_r = l2; a.log_b = l1; a.kind = 1; a.sig_lo = l1; a.sig_hi = 0;
    a.in_len = (uint32_t)std::min<size_t>(in_len, (size_t)1 << log_n);
    set_prepost(true, false);
    err = launch_pass(cfg_, a, log_n, stream);
    if (err != hipSuccess) return err;
    // pass Y: DFT over i1 (length n1, stride n2), columns j2 contiguous
    a.in = mid; a.out = out; a.tw_local = pl->tw_local[0]; a.tw_pass = nullptr;
    a.log_r = l1; a.log_b = l2; a.kind = 0; a.in_len = 1u << log_n; a.fold29 = 1;
    set_prepost(false, true);
    return launch_pass(cfg_, a, log_n, stream);
  }
  const uint32_t l1 = pl->l[0], l2 = pl->l[1], l3 = pl->l[2];
  fp_words* mid = inplace ? scratch : out;
  // pass A: DFT over i3 (length n3, stride n1 n2); writes j3 + n3*(i2 + n2*i1)
  a.in = in; a.out = mid; a.tw_local = pl->tw_local[2]; a.tw_pass = pl->tw_pass[0];
  a.log_r = l3; a.log_b = l1 + l2; a.kind = 1; a.sig_lo = l1; a.sig_hi = l2;
  a.in_len = (uint32_t)std::min<size_t>(in_len, (size_t)1 << log_n);
  set_prepost(true, false);
  err = launch_pass(cfg_, a, log_n, stream);
  if (err != hipSuccess) return err;
  // pass B: DFT over i2 (length n2, stride n3) for each i1, in place
  a.in = mid; a.out = mid; a.tw_local = pl->tw_local[1]; a.tw_pass = pl->tw_pass[1];
  a.log_r = l2; a.log_b = l3; a.kind = 0; a.in_len = 1u << log_n;
  set_prepost(false, false);
  err = launch_pass(cfg_, a, log_n, stream);
  if (err != hipSuccess) return err;
  // pass C: DFT over i1 (length n1, stride n2 n3)
  a.in = mid; a.out = out; a.tw_local = pl->tw_local[0]; a.tw_pass = nullptr;
  a.log_r = l1; a.log_b = l2 + l3; a.kind = 0; a.fold29 = 1;
  set_prepost(false, true);
  return launch_pass(cfg_, a, log_n, stream);
}

hipError_t ntt_scale(fp_words* a, const words8& s, size_t n, hipStream_t stream) {
  scale_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(a, s, n);
  return hipGetLastError();
}
hipError_t ntt_scale_periodic(fp_words* a, const fp_words* tab, uint32_t period, size_t n, hipStream_t stream) {
  scale_periodic_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(a, tab, period, n);
  return hipGetLastError();
}
hipError_t fr_montgomery(const fp_words* in, fp_words* out, size_t n, int to_mont, hipStream_t stream) {
  if (!n) return hipSuccess;
  to_mont_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(in, out, n, to_mont);
  return hipGetLastError();
}

}  // namespace sg
