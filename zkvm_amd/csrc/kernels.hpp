// kernels.hpp -- the HIP kernels of the MSM pipeline (gfx950).
//
// Pipeline (one HIP stream, no host round trips until the end):
//
//   k_decompress        32-byte ristretto encodings -> 128-byte affine-Niels rows
//   k_digits_count      scalars -> signed radix-2^w digits; histogram of
//                       (msm, window, |digit|) bins            (L2 atomics)
//   k_scan_*            exclusive prefix sum of the histogram
//   k_digits_scatter    write (sign | point-row) entries into their bins
//   k_bucket_accumulate one lane per bin: sum its points with mixed additions
//                       (the dominant kernel: 7 field multiplications per term
//                       and window)
//   k_bucket_reduce     sum_b (b+1) * bucket[b] per (msm, window[, chunk]) by
//                       running sums
//   k_window_partials   fold the chunk partials of one window (wave shuffles)
//   k_msm_finish        Horner over windows + ristretto identity test, one lane
//                       per MSM (batch mode); single MSMs are finished on the host
//
// Data layout in HBM
//   scalars   n x 32 B little endian (as given)
//   niels row 32 x u32: ypx[10] ymx[10] xy2d[10] valid pad   (128 B, 16 B aligned)
//   ext row   40 x u32: X[10] Y[10] Z[10] T[10]              (160 B)
//   entries   u32: bit 31 = subtract, bit 30 = row lives in the per-call
//             ("dynamic") table, bits 0..29 = row index
#pragma once
#include "curve.hpp"
#include "quad.hpp"
#include "sc_dev.hpp"

namespace zk {

constexpr uint32_t ENTRY_NEG = 1u << 31;
constexpr uint32_t ENTRY_DYN = 1u << 30;
constexpr uint32_t ENTRY_IDX = (1u << 30) - 1;
constexpr int NIELS_WORDS = 32;
constexpr int EXT_WORDS = 40;
constexpr int REDUCE_CHUNK = 64;   // buckets per lane in k_bucket_reduce when a window has at most this many
#ifndef ZK_REDUCE_CHUNK_BIG
#define ZK_REDUCE_CHUNK_BIG 16
#endif
constexpr int REDUCE_CHUNK_BIG = ZK_REDUCE_CHUNK_BIG;  // ... and when it has more (more lanes, shorter serial chains)
constexpr int TABLE_WORDS = 24;    // fixed-base table row: three canonical 255-bit values, 96 B
constexpr int TABLE_STRIDE = 32;   // ... one row per 128-byte line (96-byte rows at a 96-byte stride straddle lines: 2.07x the bytes addressed)

struct JobDesc {
  // dynamic terms (own compressed points)
  const uint32_t* dyn_scalars;   // n_dyn x 8 words
  const uint64_t* dyn_offsets;   // n_msm + 1   (nullptr when n_msm == 1)
  uint64_t n_dyn;
  // static terms (points of a resident set)
  const uint32_t* st_scalars;    // n_static x 8 words
  const uint32_t* st_index;      // n_static, or nullptr = position within the row
  const uint64_t* st_offsets;    // n_msm + 1
  uint64_t n_static;
  uint32_t n_msm;
  int w;                         // window bits
  int n_windows;                 // 255 / w + 1
  uint32_t n_buckets;            // 2^(w-1)
};

// ---- small helpers ---------------------------------------------------------

__device__ __forceinline__ void load_niels(ge_niels& q, const uint32_t* row) {
  const uint4* r4 = reinterpret_cast<const uint4*>(row);
  uint32_t w[32];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint4 v = r4[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) { q.ypx.v[i] = w[i]; q.ymx.v[i] = w[10 + i]; q.xy2d.v[i] = w[20 + i]; }
}

__device__ __forceinline__ void store_niels(uint32_t* row, const ge_niels& q, uint32_t valid) {
  uint32_t w[32];
#pragma unroll
  for (int i = 0; i < 10; ++i) { w[i] = q.ypx.v[i]; w[10 + i] = q.ymx.v[i]; w[20 + i] = q.xy2d.v[i]; }
  w[30] = valid; w[31] = 0;
  uint4* r4 = reinterpret_cast<uint4*>(row);
#pragma unroll
  for (int i = 0; i < 8; ++i) r4[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

__device__ __forceinline__ void load_ext(ge& p, const uint32_t* row) {
  const uint4* r4 = reinterpret_cast<const uint4*>(row);
  uint32_t w[40];
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint4 v = r4[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) { p.X.v[i] = w[i]; p.Y.v[i] = w[10 + i]; p.Z.v[i] = w[20 + i]; p.T.v[i] = w[30 + i]; }
}

__device__ __forceinline__ void store_ext(uint32_t* row, const ge& p) {
  uint32_t w[40];
#pragma unroll
  for (int i = 0; i < 10; ++i) { w[i] = p.X.v[i]; w[10 + i] = p.Y.v[i]; w[20 + i] = p.Z.v[i]; w[30 + i] = p.T.v[i]; }
  uint4* r4 = reinterpret_cast<uint4*>(row);
#pragma unroll
  for (int i = 0; i < 10; ++i) r4[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// fixed-base table rows are packed (3 x 32 B) to cut the gather traffic by a quarter
__device__ __forceinline__ void load_table_row(ge_niels& q, const uint32_t* row) {
  const uint4* r4 = reinterpret_cast<const uint4*>(row);
  uint32_t w[24];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    uint4 v = r4[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
  fe_from_words(q.ypx, w);
  fe_from_words(q.ymx, w + 8);
  fe_from_words(q.xy2d, w + 16);
}

__device__ __forceinline__ void store_table_row(uint32_t* row, const ge_niels& q) {
  uint32_t w[24];
  fe_to_words(w, q.ypx);
  fe_to_words(w + 8, q.ymx);
  fe_to_words(w + 16, q.xy2d);
  uint4* r4 = reinterpret_cast<uint4*>(row);
#pragma unroll
  for (int i = 0; i < 6; ++i) r4[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// largest m with offsets[m] <= g  (offsets has n_msm + 1 entries, non-decreasing)
__device__ __forceinline__ uint32_t find_row(const uint64_t* __restrict__ offsets, uint32_t n_msm, uint64_t g) {
  uint32_t lo = 0, hi = n_msm;   // invariant: offsets[lo] <= g < offsets[hi]
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

// ---- k_decompress ----------------------------------------------------------
// status[0] |= 1 when any point is invalid; status[1] = min bad index (atomicMin)
__global__ void __launch_bounds__(256)
k_decompress(const uint32_t* __restrict__ pts, uint32_t* __restrict__ rows, uint64_t n,
             const uint64_t* __restrict__ offsets, uint32_t n_msm, uint32_t* __restrict__ msm_fail,
             unsigned long long* __restrict__ bad_index, uint8_t* __restrict__ ok_out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* p4 = reinterpret_cast<const uint4*>(pts + 8 * i);
  uint4 a = p4[0], b = p4[1];
  uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  fe x, y;
  bool ok = ristretto_decode_affine(x, y, w);
  ge_niels q;
  niels_from_affine(q, x, y);
  if (!ok) niels_identity(q);
  if (rows) store_niels(rows + NIELS_WORDS * i, q, ok ? 1u : 0u);
  if (ok_out) ok_out[i] = ok ? 1 : 0;
  if (!ok) {
    if (bad_index) atomicMin(bad_index, (unsigned long long)i);
    if (msm_fail) {
      uint32_t m = (n_msm > 1 && offsets) ? find_row(offsets, n_msm, i) : 0;
      atomicOr(&msm_fail[m], 1u);
    }
  }
}

// ---- split decompression (large point counts) ----------------------------------
// DECODE is one 250-squaring chain wrapped in a dozen multiplications.  Fused, the chain
// shares its registers with everything live around it and the kernel runs one wave per
// SIMD; split in three, the chain runs alone in a ~100-register kernel at full occupancy
// and the ends trade 280 B/point of scratch traffic for it.
//   scratch, word-major (word k of point i at scratch[k n + i]: every load and store of a wavefront is one contiguous
//   256-byte run -- point-major rows of 64 words cost four times the traffic): s[10] u1[10] u2[10] v[10] w[10] r[10] flags
//   (w = v * u2^2, r = w^((p-5)/8))
constexpr int DEC_WORDS = 64;

// what DECODE computes before the square root: s, u1 = 1 - s^2, u2 = 1 + s^2, v, w = v u2^2 (the argument of
// SQRT_RATIO_M1(1, w)), w^7 (what the chain is handed) and "s is canonical and non-negative".  A dozen products: cheap enough
// to compute TWICE -- k_decompress_pre keeps w^7 alone, k_decompress_post reads the 32 bytes of the point again and
// recomputes the rest -- which takes 200 B per point out of the scratch traffic of each of the two kernels (round 4: they
// are bandwidth-bound and sit on the critical path of the 2^20 MSM; until round 3 s, u1, u2, v, w went through the scratch).
struct DecFront { fe s, u1, u2, v, w, w7; bool pre_ok; };
__device__ __forceinline__ void dec_front(DecFront& d, const uint32_t* __restrict__ pts, uint64_t i) {
  const uint4* p4 = reinterpret_cast<const uint4*>(pts + 8 * i);
  uint4 a = p4[0], b = p4[1];
  uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  fe ss, u2_sqr, t;
  fe_from_words(d.s, w);
  uint32_t chk[8];
  fe_to_words(chk, d.s);
  bool canonical = true;
#pragma unroll
  for (int k = 0; k < 8; ++k) canonical &= (chk[k] == w[k]);
  d.pre_ok = canonical && !(w[0] & 1);
  fe_sq(ss, d.s);
  fe_sub(d.u1, fe_one(), ss);
  fe_add(d.u2, fe_one(), ss);
  fe_carry(d.u1);
  fe_carry(d.u2);
  fe_sq(u2_sqr, d.u2);
  fe_sq(t, d.u1);
  fe_mul(t, t, fe_D());
  fe_add(t, t, u2_sqr);
  fe_sub_c(d.v, fe_zero(), t);
  fe_mul(d.w, d.v, u2_sqr);           // w: the argument of SQRT_RATIO_M1(1, w)
  // (1 * w^7)^((p-5)/8) is what the chain computes
  fe w2, w3;
  fe_sq(w2, d.w);
  fe_mul(w3, w2, d.w);
  fe_sq(d.w7, w3);
  fe_mul(d.w7, d.w7, d.w);
}

__global__ void __launch_bounds__(256)
k_decompress_pre(const uint32_t* __restrict__ pts, uint32_t* __restrict__ scratch, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  DecFront d;
  dec_front(d, pts, i);
  uint32_t* col = scratch + (uint64_t)50 * n + i;
#pragma unroll
  for (int k = 0; k < 10; ++k) col[(uint64_t)k * n] = d.w7.v[k];
}

// row[50..59] <- row[50..59]^((p-5)/8)
// ZK_POW_WAVES wavefronts per SIMD: at 4 (until round 6) the kernel is held to 128 registers -- 84 bytes of it in scratch -- and
// takes EVERY register of every SIMD, so that the digit sort "beside" it (k_part_*: 20 - 52 registers a lane) only ran as its
// workgroups retired, a quarter of the chip every 226 us (profiles/r06c_*: k_part_offsets, 4 MB of counters,
// 207 us).  One wavefront alone issues every 4.6 cycles, two every 4.3: a chain of dependent squarings
// does not need the fourth.
#ifndef ZK_POW_WAVES
#define ZK_POW_WAVES 3
#endif
__global__ void __launch_bounds__(256, ZK_POW_WAVES)
k_pow22523(uint32_t* __restrict__ scratch, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t* col = scratch + (uint64_t)50 * n + i;
  fe x, r;
#pragma unroll
  for (int k = 0; k < 10; ++k) x.v[k] = col[(uint64_t)k * n];
  fe_pow22523(r, x);
#pragma unroll
  for (int k = 0; k < 10; ++k) col[(uint64_t)k * n] = r.v[k];
}

__global__ void __launch_bounds__(256)
k_decompress_post(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ scratch, uint32_t* __restrict__ rows, uint64_t n,
                  const uint64_t* __restrict__ offsets, uint32_t n_msm, uint32_t* __restrict__ msm_fail,
                  unsigned long long* __restrict__ bad_index) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  DecFront d;
  dec_front(d, pts, i);
  const fe s = d.s, u1 = d.u1, u2 = d.u2, v = d.v, w = d.w;
  const bool pre_ok = d.pre_ok;
  const uint32_t* col = scratch + (uint64_t)50 * n + i;
  fe pw;
#pragma unroll
  for (int k = 0; k < 10; ++k) pw.v[k] = col[(uint64_t)k * n];
  // SQRT_RATIO_M1(1, w): r = w^3 * (w^7)^((p-5)/8); check = w r^2
  fe w3, r, t, check, r_prime, neg_r;
  fe_sq(t, w);
  fe_mul(w3, t, w);
  fe_mul(r, pw, w3);
  fe_sq(t, r);
  fe_mul(check, t, w);
  const fe one = fe_one();
  fe neg_one, neg_i;
  fe_neg(neg_one, one);
  fe_mul(neg_i, neg_one, fe_SQRT_M1());
  const bool correct_sign = fe_eq(check, one);
  const bool flipped_sign = fe_eq(check, neg_one);
  const bool flipped_sign_i = fe_eq(check, neg_i);
  fe_mul(r_prime, r, fe_SQRT_M1());
  fe_cmov(r, r_prime, flipped_sign | flipped_sign_i);
  fe_neg(neg_r, r);
  fe_cmov(r, neg_r, fe_is_negative(r));
  const bool was_square = correct_sign | flipped_sign;
  fe den_x, den_y, x, y, nx;
  fe_mul(den_x, r, u2);
  fe_mul(den_y, r, den_x);
  fe_mul(den_y, den_y, v);
  fe_mul(x, s, den_x);
  fe_add(x, x, x);
  fe_carry(x);
  fe_neg(nx, x);
  fe_cmov(x, nx, fe_is_negative(x));
  fe_mul(y, u1, den_y);
  fe_mul(t, x, y);
  const bool ok = pre_ok & was_square & !fe_is_negative(t) & !fe_is_zero(y);
  ge_niels q;
  niels_from_affine(q, x, y);
  if (!ok) niels_identity(q);
  store_niels(rows + NIELS_WORDS * i, q, ok ? 1u : 0u);
  if (!ok) {
    if (bad_index) atomicMin(bad_index, (unsigned long long)i);
    if (msm_fail) {
      uint32_t m = (n_msm > 1 && offsets) ? find_row(offsets, n_msm, i) : 0;
      atomicOr(&msm_fail[m], 1u);
    }
  }
}

// ---- digits ----------------------------------------------------------------
// Signed radix-2^w recoding, digits in (-2^(w-1), 2^(w-1)], least significant
// window first.  `f(t, d)` is called for every non-zero digit.
template <typename F>
__device__ __forceinline__ void for_each_digit(const uint32_t* __restrict__ sc, int w, int n_windows, F f) {
  uint32_t s[8];
  const uint4* s4 = reinterpret_cast<const uint4*>(sc);
  uint4 a = s4[0], b = s4[1];
  s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
  const uint32_t mask = (1u << w) - 1, half = 1u << (w - 1);
  uint32_t carry = 0;
  for (int t = 0; t < n_windows; ++t) {
    const int pos = t * w;
    uint32_t bits = 0;
    if (pos < 256) {
      const int idx = pos >> 5, sh = pos & 31;
      // select words idx, idx+1 without dynamic register indexing
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) { lo = (k == idx) ? s[k] : lo; hi = (k == idx + 1) ? s[k] : hi; }
      uint64_t two = (uint64_t)lo | ((uint64_t)hi << 32);
      bits = (uint32_t)(two >> sh) & mask;
    }
    uint32_t v = bits + carry;
    int d;
    if (v > half) { d = (int)v - (int)(mask + 1); carry = 1; }
    else { d = (int)v; carry = 0; }
    if (d != 0) f(t, d);
  }
}

__device__ __forceinline__ void term_lookup(const JobDesc& j, uint64_t g, const uint32_t*& sc, uint32_t& m,
                                            uint32_t& entry) {
  if (g < j.n_dyn) {
    sc = j.dyn_scalars + 8 * g;
    m = (j.n_msm > 1) ? find_row(j.dyn_offsets, j.n_msm, g) : 0;
    entry = ENTRY_DYN | (uint32_t)g;
  } else {
    const uint64_t k = g - j.n_dyn;
    sc = j.st_scalars + 8 * k;
    m = (j.n_msm > 1) ? find_row(j.st_offsets, j.n_msm, k) : 0;
    entry = j.st_index ? j.st_index[k] : (uint32_t)(k - (j.n_msm > 1 ? j.st_offsets[m] : 0));
  }
}

// status[0] bit 1: a scalar had bit 255 set
__global__ void __launch_bounds__(256)
k_digits_count(JobDesc j, uint32_t* __restrict__ hist, uint32_t* __restrict__ status) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= j.n_dyn + j.n_static) return;
  const uint32_t* sc; uint32_t m, entry;
  term_lookup(j, g, sc, m, entry);
  if (sc[7] >> 31) atomicOr(&status[0], 2u);
  const uint64_t base = (uint64_t)m * j.n_windows;
  for_each_digit(sc, j.w, j.n_windows, [&](int t, int d) {
    const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
    atomicAdd(&hist[(base + t) * j.n_buckets + b], 1u);
  });
}

__global__ void __launch_bounds__(256)
k_digits_scatter(JobDesc j, uint32_t* __restrict__ cursor, uint32_t* __restrict__ entries) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= j.n_dyn + j.n_static) return;
  const uint32_t* sc; uint32_t m, entry;
  term_lookup(j, g, sc, m, entry);
  const uint64_t base = (uint64_t)m * j.n_windows;
  for_each_digit(sc, j.w, j.n_windows, [&](int t, int d) {
    const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
    const uint32_t pos = atomicAdd(&cursor[(base + t) * j.n_buckets + b], 1u);
    entries[pos] = entry | (d < 0 ? ENTRY_NEG : 0u);
  });
}

// ---- exclusive scan over u32 (three launches) --------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_TILE = SCAN_ITEMS * SCAN_BLOCK;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t& total) {
  __shared__ uint32_t wave_tot[SCAN_BLOCK / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) wave_tot[wid] = x;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < SCAN_BLOCK / 64; ++k) {
    uint32_t wt = wave_tot[k];
    if (k < wid) base += wt;
    tot += wt;
  }
  __syncthreads();
  total = tot;
  return base + x - v;
}

__global__ void __launch_bounds__(SCAN_BLOCK)
k_scan_reduce(const uint32_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ block_sums) {
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) s += in[base + k];
  uint32_t total;
  (void)block_exclusive_scan(s, total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(SCAN_BLOCK)
k_scan_blocksums(uint32_t* __restrict__ block_sums, uint32_t n_blocks) {
  uint32_t carry = 0;
  for (uint32_t start = 0; start < n_blocks; start += SCAN_BLOCK) {
    const uint32_t i = start + threadIdx.x;
    uint32_t v = i < n_blocks ? block_sums[i] : 0;
    uint32_t total;
    uint32_t ex = block_exclusive_scan(v, total);
    if (i < n_blocks) block_sums[i] = carry + ex;
    carry += total;
  }
}

__global__ void __launch_bounds__(SCAN_BLOCK)
k_scan_apply(uint32_t* __restrict__ data, uint64_t n, const uint32_t* __restrict__ block_sums) {
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = (base + k < n) ? data[base + k] : 0; s += v[k]; }
  uint32_t total;
  uint32_t ex = block_exclusive_scan(s, total) + block_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < n) data[base + k] = ex; ex += v[k]; }
}

// ---- partition sort (single large MSM) ------------------------------------------------
// The (window, bucket) sort of a 2^20-term MSM is 16.7 M keys over 524 288 bins.  Doing
// it with one global atomic per key runs at the chip's random-atomic rate (~20 G/s,
// MI355X_MICROARCH.md "Global float atomics": they execute at the memory side) and
// cost more than the point additions.  Here every atomic is an LDS atomic:
//   k_part_hist     tile of PART_TILE terms -> LDS histogram over partitions
//                   p = (window, high bucket bits); counts out partition-major
//   k_scan_*        exclusive scan of the (partition, tile) counts
//   k_part_scatter  same tile: rank inside (partition, tile) by LDS atomic, entries
//                   land in 128-byte runs per partition
//   k_part_sort     one workgroup per partition: histogram of the low bucket bits
//                   (LDS), scan, then place every entry at its final position and
//                   emit the bin end offsets the accumulate kernel reads
constexpr int PART_TILE = 2048;      // terms per tile
constexpr int PART_LO_BITS = 8;      // low bucket bits sorted inside a partition

struct PartShape {
  uint32_t n_part, n_tiles, hi_bits, lo_bits;
};

__global__ void __launch_bounds__(256)
k_part_hist(JobDesc j, PartShape ps, uint32_t* __restrict__ hist, uint32_t* __restrict__ status) {
  extern __shared__ uint32_t cnt[];
  for (uint32_t p = threadIdx.x; p < ps.n_part; p += 256) cnt[p] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * PART_TILE;
  for (int r = 0; r < PART_TILE / 256; ++r) {
    const uint64_t g = base + r * 256 + threadIdx.x;
    if (g < j.n_dyn) {
      const uint32_t* sc = j.dyn_scalars + 8 * g;
      if (sc[7] >> 31) atomicOr(&status[0], 2u);
      for_each_digit(sc, j.w, j.n_windows, [&](int t, int d) {
        const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
        atomicAdd(&cnt[((uint32_t)t << ps.hi_bits) | (b >> ps.lo_bits)], 1u);
      });
    }
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < ps.n_part; p += 256) hist[(uint64_t)p * ps.n_tiles + blockIdx.x] = cnt[p];
}

// Offsets of the (partition, tile) cells in ONE launch (until round 3: the generic three-launch scan over all n_part x n_tiles
// cells, 0.3 ms of a 2.8 ms call for 4 MB of counters -- launch latency, not bandwidth).  One workgroup per partition: an
// exclusive scan of its tile counts in place (the position of every tile's run INSIDE the partition) and the partition's
// total; the workgroup that finishes last -- a ticket counter tells it -- scans the totals into the partitions' bases
// (part_base[n_part] = the grand total).  The scatter adds the two.
__global__ void __launch_bounds__(256)
k_part_offsets(PartShape ps, uint32_t* __restrict__ hist /*[n_part][n_tiles], in place*/, uint32_t* __restrict__ totals /*[n_part]*/,
               uint32_t* __restrict__ part_base /*[n_part + 1]*/, uint32_t* __restrict__ ticket) {
  __shared__ uint32_t last;
  const uint32_t p = blockIdx.x, t = threadIdx.x;
  uint32_t* row = hist + (uint64_t)p * ps.n_tiles;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < ps.n_tiles; base += 256) {
    const uint32_t i = base + t;
    const uint32_t v = i < ps.n_tiles ? row[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(v, total);
    if (i < ps.n_tiles) row[i] = carry + ex;
    carry += total;
  }
  if (t == 0) {
    totals[p] = carry;
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  uint32_t run = 0;
  for (uint32_t base = 0; base < ps.n_part; base += 256) {
    const uint32_t i = base + t;
    const uint32_t v = i < ps.n_part ? atomicAdd(&totals[i], 0u) : 0u;      // (read at the L2: written by other workgroups)
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(v, total);
    if (i < ps.n_part) part_base[i] = run + ex;
    run += total;
  }
  if (t == 0) { part_base[ps.n_part] = run; *ticket = 0; }
}

// An entry on its way through the partition sort carries its low bucket bits with it: [31:24] low bucket bits, [23] sign,
// [22:0] term index (the path is taken for at most 2^23 terms) -- one scattered 4-byte store per entry instead of a 4-byte
// and a 1-byte one, and one array for the partition kernel to read.
constexpr uint32_t PART_IDX_BITS = 23;

__global__ void __launch_bounds__(256)
k_part_scatter(JobDesc j, PartShape ps, const uint32_t* __restrict__ offs, const uint32_t* __restrict__ part_base,
               uint32_t* __restrict__ pe) {
  extern __shared__ uint32_t cur[];
  for (uint32_t p = threadIdx.x; p < ps.n_part; p += 256) cur[p] = part_base[p] + offs[(uint64_t)p * ps.n_tiles + blockIdx.x];
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * PART_TILE;
  const uint32_t lo_mask = (1u << ps.lo_bits) - 1;
  for (int r = 0; r < PART_TILE / 256; ++r) {
    const uint64_t g = base + r * 256 + threadIdx.x;
    if (g < j.n_dyn) {
      const uint32_t* sc = j.dyn_scalars + 8 * g;
      for_each_digit(sc, j.w, j.n_windows, [&](int t, int d) {
        const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
        const uint32_t pos = atomicAdd(&cur[((uint32_t)t << ps.hi_bits) | (b >> ps.lo_bits)], 1u);
        pe[pos] = ((b & lo_mask) << 24) | (d < 0 ? 1u << PART_IDX_BITS : 0u) | (uint32_t)g;
      });
    }
  }
}

// One workgroup per partition: histogram of the low bucket bits (LDS), scan, then every entry to its final position -- in
// LDS when the partition fits (PART_SORT_LDS entries: the sorted run then leaves as whole cache lines; placed straight into
// HBM, as until round 3, the 4-byte stores of 2048 concurrent workgroups thrashed the L2: 492 MB written for 67 MB of
// entries), straight to HBM otherwise (adversarial inputs: every term in one partition).  Also emits the bin end offsets
// the accumulate kernel reads and, for the bin ordering, the size-class histogram of its 256 bins (k_bin_classes folded in).
constexpr uint32_t PART_SORT_LDS = 12288;
constexpr int SIZE_CLASSES = 256;

__global__ void __launch_bounds__(256)
k_part_sort(PartShape ps, const uint32_t* __restrict__ part_base, const uint32_t* __restrict__ pe,
            uint32_t* __restrict__ entries, uint32_t* __restrict__ cursor, uint32_t* __restrict__ class_count) {
  __shared__ uint32_t c[256], cur[256], cls[SIZE_CLASSES];
  extern __shared__ uint32_t staged[];                  // PART_SORT_LDS entries
  const uint32_t p = blockIdx.x, t = threadIdx.x;
  const uint32_t start = part_base[p], end = part_base[p + 1];
  c[t] = 0; cls[t] = 0;
  __syncthreads();
  for (uint32_t e = start + t; e < end; e += 256) atomicAdd(&c[pe[e] >> 24], 1u);
  __syncthreads();
  uint32_t total;
  const uint32_t mine = c[t];
  const uint32_t ex = block_exclusive_scan(mine, total);
  cur[t] = ex;
  const uint32_t n_lo = 1u << ps.lo_bits;
  if (t < n_lo) {
    cursor[(uint64_t)p * n_lo + t] = start + ex + mine;   // END offset of bin (p, lo = t)
    atomicAdd(&cls[(SIZE_CLASSES - 1) - min(mine, (uint32_t)(SIZE_CLASSES - 1))], 1u);
  }
  __syncthreads();
  if (cls[t]) atomicAdd(&class_count[t], cls[t]);
  const uint32_t n = end - start;
  const bool in_lds = n <= PART_SORT_LDS;
  for (uint32_t e = start + t; e < end; e += 256) {
    const uint32_t v = pe[e];
    const uint32_t pos = atomicAdd(&cur[v >> 24], 1u);
    const uint32_t out = ENTRY_DYN | (v & ((1u << PART_IDX_BITS) - 1)) | ((v >> PART_IDX_BITS) & 1u ? ENTRY_NEG : 0u);
    if (in_lds) staged[pos] = out; else entries[start + pos] = out;
  }
  if (!in_lds) return;
  __syncthreads();
  for (uint32_t i = t; i < n; i += 256) entries[start + i] = staged[i];
}

__device__ __forceinline__ void shfl_up_ge(ge& out, const ge& in, int delta) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    out.X.v[i] = __shfl_up(in.X.v[i], delta);
    out.Y.v[i] = __shfl_up(in.Y.v[i], delta);
    out.Z.v[i] = __shfl_up(in.Z.v[i], delta);
    out.T.v[i] = __shfl_up(in.T.v[i], delta);
  }
}
__device__ __forceinline__ void shfl_down_ge(ge& out, const ge& in, int delta) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    out.X.v[i] = __shfl_down(in.X.v[i], delta);
    out.Y.v[i] = __shfl_down(in.Y.v[i], delta);
    out.Z.v[i] = __shfl_down(in.Z.v[i], delta);
    out.T.v[i] = __shfl_down(in.T.v[i], delta);
  }
}

// ---- bin ordering ---------------------------------------------------------------
// One lane sums one bin, so a wavefront takes as long as its fullest bin.  Bins are
// therefore handed to lanes in order of decreasing size (256 size classes, counting
// sort): the lanes of a wavefront get bins of (nearly) equal size, and the fullest
// bins -- e.g. the top window, whose few significant bits concentrate the terms in a
// handful of buckets -- start first instead of forming a tail.
__device__ __forceinline__ uint32_t bin_size_class(const uint32_t* __restrict__ cursor, uint64_t bin) {
  const uint32_t cnt = cursor[bin] - (bin ? cursor[bin - 1] : 0u);
  return (SIZE_CLASSES - 1) - min(cnt, (uint32_t)(SIZE_CLASSES - 1));   // class 0 = fullest
}

__global__ void __launch_bounds__(256)
k_bin_classes(const uint32_t* __restrict__ cursor, uint64_t n_bins, uint32_t* __restrict__ class_count) {
  __shared__ uint32_t h[SIZE_CLASSES];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t bin = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (bin < n_bins) atomicAdd(&h[bin_size_class(cursor, bin)], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&class_count[threadIdx.x], h[threadIdx.x]);
}

// class_count[256] -> exclusive offsets in class_cursor[256]
__global__ void __launch_bounds__(256)
k_class_scan(const uint32_t* __restrict__ class_count, uint32_t* __restrict__ class_cursor) {
  uint32_t total;
  const uint32_t ex = block_exclusive_scan(class_count[threadIdx.x], total);
  class_cursor[threadIdx.x] = ex;
}

// Bins above HEAVY_BIN entries (all terms sharing one digit: equal scalars, tiny scalars,
// a top window with one significant bit) would pin one lane for milliseconds to seconds;
// they are listed in heavy[1..] (heavy[0] = count) and summed by a whole workgroup each.
constexpr uint32_t HEAVY_BIN = 2048;
constexpr uint32_t HEAVY_MAX = 65536;
// Bins between FAT_BIN and HEAVY_BIN entries are summed by FAT_LANES lanes each (k_bucket_fat).  The case that matters is the
// TOP window of a large multiscalar multiplication with uniform scalars: l < 2^253, so window 15 of sixteen 16-bit windows has 13
// significant bits and its 2^20 terms land in ~4100 buckets of ~256 -- eight times the ~32 of every other bin.  One lane per
// bin, those 64 wavefronts ran for the WHOLE kernel (started first, they still shared their SIMDs three ways for the first two
// thirds and then finished alone: 0.84 - 0.89 ms against the 0.50 ms the kernel's instructions cost) -- round 6.
constexpr uint32_t FAT_BIN = 96;
constexpr int FAT_LANES = 16;

__global__ void __launch_bounds__(256)
k_bin_order(const uint32_t* __restrict__ cursor, uint64_t n_bins, uint32_t* __restrict__ class_cursor,
            uint32_t* __restrict__ order, uint32_t* __restrict__ heavy, uint32_t* __restrict__ fat, uint32_t fat_cap) {
  __shared__ uint32_t h[SIZE_CLASSES], base[SIZE_CLASSES];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t bin = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t cls = 0, rank = 0;
  if (bin < n_bins) {
    cls = bin_size_class(cursor, bin);
    rank = atomicAdd(&h[cls], 1u);
    const uint32_t cnt = cursor[bin] - (bin ? cursor[bin - 1] : 0u);
    if (cnt > HEAVY_BIN) {
      const uint32_t slot = atomicAdd(&heavy[0], 1u);
      if (slot < HEAVY_MAX) heavy[1 + slot] = (uint32_t)bin;
    } else if (cnt > FAT_BIN) {
      const uint32_t slot = atomicAdd(&fat[0], 1u);        // (cannot overflow: fat_cap >= entries / FAT_BIN)
      if (slot < fat_cap) fat[1 + slot] = (uint32_t)bin;
    }
  }
  __syncthreads();
  if (h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&class_cursor[threadIdx.x], h[threadIdx.x]);
  __syncthreads();
  if (bin < n_bins) order[base[cls] + rank] = (uint32_t)bin;
}

// ---- k_bucket_accumulate ------------------------------------------------------
// After the scatter, cursor[bin] is the END offset of bin; its start is the end
// of the previous bin.  One lane per bin, bins taken in `order` (fullest first);
// the row of the next entry is fetched while the current addition runs.
// FAT_LANES lanes per fat bin (four bins per wavefront): strided partial sums, folded by four shuffle steps inside the group.
// The lane groups walk the fat list (n_groups of them in all).
__device__ __forceinline__ void bucket_fat_role(const uint32_t* __restrict__ cursor, const uint32_t* __restrict__ entries,
                                                const uint32_t* __restrict__ static_rows, const uint32_t* __restrict__ dyn_rows,
                                                uint32_t* __restrict__ buckets, const uint32_t* __restrict__ fat, uint32_t fat_cap,
                                                uint32_t first_group, uint32_t n_groups) {
  const uint32_t n_fat = min(fat[0], fat_cap);
  const uint32_t sub = threadIdx.x & (FAT_LANES - 1);
  for (uint32_t f = first_group + threadIdx.x / FAT_LANES; ; f += n_groups) {
    // (a wavefront leaves the loop together: its four groups hold consecutive list positions, and the shuffles below need all lanes)
    if ((f & ~3u) >= n_fat) break;
    const bool live = f < n_fat;
    const uint64_t bin = live ? fat[1 + f] : 0;
    const uint32_t start = live ? (bin ? cursor[bin - 1] : 0u) : 0u, end = live ? cursor[bin] : 0u;
    ge acc;
    ge_identity(acc);
    uint32_t k = start + sub;
    uint32_t e_next = k < end ? entries[k] : 0u;
    for (; k < end; k += FAT_LANES) {
      const uint32_t e = e_next;
      if (k + FAT_LANES < end) e_next = entries[k + FAT_LANES];
      const uint32_t* row = ((e & ENTRY_DYN) ? dyn_rows : static_rows) + (uint64_t)(e & ENTRY_IDX) * NIELS_WORDS;
      ge_niels q;
      load_niels(q, row);
      ge_madd(acc, acc, q, (e & ENTRY_NEG) != 0);
    }
#pragma unroll 1
    for (int delta = FAT_LANES / 2; delta >= 1; delta >>= 1) {
      ge other;
      shfl_down_ge(other, acc, delta);
      if (sub < (uint32_t)delta) ge_add(acc, acc, other);
    }
    if (live && sub == 0) store_ext(buckets + bin * EXT_WORDS, acc);
  }
}

// The first `fat_blocks` workgroups of the grid sum the fat bins (bucket_fat_role), the others one bin per lane.  ONE launch:
// the fat groups carry the longest chains (up to HEAVY_BIN / FAT_LANES additions and four folds) and must START first -- as a
// kernel of their own on a second stream they got their slots only as the one-lane-per-bin workgroups retired and ended when
// those did (profiles/r06c_*), queued ahead on the same stream they cost their 84 us on an idle chip.
__global__ void __launch_bounds__(256)
k_bucket_accumulate(const uint32_t* __restrict__ cursor, const uint32_t* __restrict__ entries,
                    const uint32_t* __restrict__ static_rows, const uint32_t* __restrict__ dyn_rows,
                    uint32_t* __restrict__ buckets, uint64_t n_bins, const uint32_t* __restrict__ order,
                    const uint32_t* __restrict__ fat, uint32_t fat_cap, uint32_t fat_blocks) {
  if (blockIdx.x < fat_blocks) {
    bucket_fat_role(cursor, entries, static_rows, dyn_rows, buckets, fat, fat_cap, blockIdx.x * (256 / FAT_LANES), fat_blocks * (256 / FAT_LANES));
    return;
  }
  const uint64_t gid = (uint64_t)(blockIdx.x - fat_blocks) * blockDim.x + threadIdx.x;
  if (gid >= n_bins) return;
  const uint64_t bin = order ? order[gid] : gid;
  const uint32_t start = bin ? cursor[bin - 1] : 0u, end = cursor[bin];
  if (start == end || end - start > FAT_BIN) return;     // fat bins: the first workgroups; heavy bins: k_bucket_heavy
  ge acc;
  ge_identity(acc);
  // two dependent gathers per term (entry -> row): entries run two terms ahead, rows one
  auto fetch_row = [&](uint32_t e, ge_niels& q, bool& neg) {
    const uint32_t* row = ((e & ENTRY_DYN) ? dyn_rows : static_rows) + (uint64_t)(e & ENTRY_IDX) * NIELS_WORDS;
    load_niels(q, row);
    neg = (e & ENTRY_NEG) != 0;
  };
  ge_niels cur, nxt;
  bool cur_neg = false, nxt_neg = false;
  fetch_row(entries[start], cur, cur_neg);
  uint32_t e_next = start + 1 < end ? entries[start + 1] : 0u;
  for (uint32_t k = start; k < end; ++k) {
    const uint32_t e_cur = e_next;
    if (k + 2 < end) e_next = entries[k + 2];
    if (k + 1 < end) fetch_row(e_cur, nxt, nxt_neg);
    ge_madd(acc, acc, cur, cur_neg);
    cur = nxt; cur_neg = nxt_neg;
  }
  store_ext(buckets + bin * EXT_WORDS, acc);
}

// One workgroup per heavy bin: 256 strided partial sums, folded by wavefront shuffles and
// one pass through LDS.  The grid is fixed; workgroups walk the heavy list.
__global__ void __launch_bounds__(256)
k_bucket_heavy(const uint32_t* __restrict__ cursor, const uint32_t* __restrict__ entries,
               const uint32_t* __restrict__ static_rows, const uint32_t* __restrict__ dyn_rows,
               uint32_t* __restrict__ buckets, const uint32_t* __restrict__ heavy) {
  __shared__ uint32_t wave_pts[4 * EXT_WORDS];
  const uint32_t n_heavy = min(heavy[0], HEAVY_MAX);
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  for (uint32_t hI = blockIdx.x; hI < n_heavy; hI += gridDim.x) {
    const uint64_t bin = heavy[1 + hI];
    const uint32_t start = bin ? cursor[bin - 1] : 0u, end = cursor[bin];
    ge acc;
    ge_identity(acc);
    for (uint32_t k = start + t; k < end; k += 256) {
      const uint32_t e = entries[k];
      const uint32_t* row = ((e & ENTRY_DYN) ? dyn_rows : static_rows) + (uint64_t)(e & ENTRY_IDX) * NIELS_WORDS;
      ge_niels q;
      load_niels(q, row);
      ge_madd(acc, acc, q, (e & ENTRY_NEG) != 0);
    }
#pragma unroll 1
    for (int delta = 32; delta >= 1; delta >>= 1) {
      ge other;
      shfl_down_ge(other, acc, delta);
      if (lane < delta) ge_add(acc, acc, other);
    }
    __syncthreads();
    if (lane == 0) {
      uint32_t* w = wave_pts + wid * EXT_WORDS;
      for (int q = 0; q < 10; ++q) { w[q] = acc.X.v[q]; w[10 + q] = acc.Y.v[q]; w[20 + q] = acc.Z.v[q]; w[30 + q] = acc.T.v[q]; }
    }
    __syncthreads();
    if (t == 0) {
      for (int wv = 1; wv < 4; ++wv) {
        ge o;
        const uint32_t* w = wave_pts + wv * EXT_WORDS;
        for (int q = 0; q < 10; ++q) { o.X.v[q] = w[q]; o.Y.v[q] = w[10 + q]; o.Z.v[q] = w[20 + q]; o.T.v[q] = w[30 + q]; }
        ge_add(acc, acc, o);
      }
      store_ext(buckets + bin * EXT_WORDS, acc);
    }
  }
}

// ---- k_bucket_reduce ------------------------------------------------------------
// One lane per (msm, window, chunk of chunk_size buckets):
//   out = sum_{b in chunk} (b + 1) * bucket[b]
//       = sum (b - lo + 1) * bucket[b]  +  lo * sum bucket[b]
// via the running-sum recurrence; empty buckets are skipped.
__global__ void __launch_bounds__(256)
k_bucket_reduce(const uint32_t* __restrict__ cursor, const uint32_t* __restrict__ buckets,
                uint32_t* __restrict__ partials, uint32_t* __restrict__ partial_nonempty,
                uint64_t n_tasks, uint32_t n_buckets, uint32_t chunks_per_window, uint32_t chunk_size) {
  const uint64_t task = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (task >= n_tasks) return;
  const uint64_t win = task / chunks_per_window;           // msm * n_windows + t
  const uint32_t chunk = (uint32_t)(task % chunks_per_window);
  const uint32_t lo = chunk * chunk_size;
  const uint32_t hi = min(lo + chunk_size, n_buckets);
  const uint64_t bin0 = win * n_buckets;
  ge run, sum;
  bool run_set = false, sum_set = false;
  for (uint32_t b = hi; b-- > lo;) {
    const uint64_t bin = bin0 + b;
    const uint32_t s = bin ? cursor[bin - 1] : 0u, e = cursor[bin];
    if (s != e) {
      ge p;
      load_ext(p, buckets + bin * EXT_WORDS);
      if (run_set) ge_add(run, run, p); else { run = p; run_set = true; }
    }
    if (run_set) {
      if (sum_set) ge_add(sum, sum, run); else { sum = run; sum_set = true; }
    }
  }
  if (sum_set && lo != 0) {
    // sum += lo * run   (lo < 2^15, double-and-add from the top bit)
    ge acc = run;
    const int top = 31 - __clz(lo);
    for (int bit = top - 1; bit >= 0; --bit) {
      ge_double(acc, acc);
      if ((lo >> bit) & 1) ge_add(acc, acc, run);
    }
    ge_add(sum, sum, acc);
  }
  partial_nonempty[task] = sum_set ? 1u : 0u;
  if (sum_set) store_ext(partials + task * EXT_WORDS, sum);
}

// The same with a QUAD of lanes per task (quad.hpp: an addition is 3 multiplications deep instead of 9, a doubling 2 instead of
// 8): for a single large multiscalar multiplication the reduction is 32 768 chains of ~54 dependent point operations on a chip
// that is otherwise idle -- latency, 0.17 ms of a 2.2 ms call.  Four lanes hold copies of the task's points and split every
// field operation; the four loads of a bucket are one address.  (Round 6.)
__global__ void __launch_bounds__(256)
k_bucket_reduce_quad(const uint32_t* __restrict__ cursor, const uint32_t* __restrict__ buckets,
                     uint32_t* __restrict__ partials, uint32_t* __restrict__ partial_nonempty,
                     uint64_t n_tasks, uint32_t n_buckets, uint32_t chunks_per_window, uint32_t chunk_size) {
  const uint64_t task = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const int r = threadIdx.x & 3;
  if (task >= n_tasks) return;                             // (whole quads leave together)
  const uint64_t win = task / chunks_per_window;
  const uint32_t chunk = (uint32_t)(task % chunks_per_window);
  const uint32_t lo = chunk * chunk_size;
  const uint32_t hi = min(lo + chunk_size, n_buckets);
  const uint64_t bin0 = win * n_buckets;
  ge run, sum;
  bool run_set = false, sum_set = false;
  for (uint32_t b = hi; b-- > lo;) {
    const uint64_t bin = bin0 + b;
    const uint32_t s = bin ? cursor[bin - 1] : 0u, e = cursor[bin];
    if (s != e) {
      ge p;
      load_ext(p, buckets + bin * EXT_WORDS);
      if (run_set) quad_add(run, p, r); else { run = p; run_set = true; }
    }
    if (run_set) {
      if (sum_set) quad_add(sum, run, r); else { sum = run; sum_set = true; }
    }
  }
  if (sum_set && lo != 0) {
    ge acc = run;
    const int top = 31 - __clz(lo);
    for (int bit = top - 1; bit >= 0; --bit) {
      quad_double(acc, r);
      if ((lo >> bit) & 1) quad_add(acc, run, r);
    }
    quad_add(sum, acc, r);
  }
  if (r == 0) {
    partial_nonempty[task] = sum_set ? 1u : 0u;
    if (sum_set) store_ext(partials + task * EXT_WORDS, sum);
  }
}

// ---- k_window_partials ------------------------------------------------------------
// One workgroup of four wavefronts per window: fold `chunks_per_window` partials into the window sum.  Each lane adds its
// strided share, the 64 lane sums of a wavefront are combined with shuffles (40 dwords per point per step), the four
// wavefront sums through LDS.  (Until round 3 one wavefront per window: 32 + 6 dependent additions for the 2048 partials of
// a 16-bit window; now 8 + 6 + 3.)
__global__ void __launch_bounds__(256)
k_window_partials(const uint32_t* __restrict__ partials, const uint32_t* __restrict__ partial_nonempty,
                  uint32_t* __restrict__ window_sums, uint32_t* __restrict__ window_nonempty,
                  uint32_t chunks_per_window) {
  __shared__ uint32_t wave_pts[4 * EXT_WORDS];
  __shared__ int wave_have[4];
  const uint64_t win = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  ge acc;
  ge_identity(acc);
  int have = 0;
  for (uint32_t c = t; c < chunks_per_window; c += 256) {
    const uint64_t task = win * chunks_per_window + c;
    if (partial_nonempty[task]) {
      ge p;
      load_ext(p, partials + task * EXT_WORDS);
      if (have) ge_add(acc, acc, p); else { acc = p; have = 1; }
    }
  }
#pragma unroll 1
  for (int delta = 32; delta >= 1; delta >>= 1) {
    ge other;
    shfl_down_ge(other, acc, delta);
    const int other_have = __shfl_down(have, delta);
    if (lane < delta) {
      if (other_have) {
        if (have) ge_add(acc, acc, other); else { acc = other; have = 1; }
      }
    }
  }
  if (lane == 0) {
    wave_have[wid] = have;
    uint32_t* w = wave_pts + wid * EXT_WORDS;
    for (int q = 0; q < 10; ++q) { w[q] = acc.X.v[q]; w[10 + q] = acc.Y.v[q]; w[20 + q] = acc.Z.v[q]; w[30 + q] = acc.T.v[q]; }
  }
  __syncthreads();
  if (t != 0) return;
  for (int wv = 1; wv < 4; ++wv) {
    if (!wave_have[wv]) continue;
    ge o;
    const uint32_t* w = wave_pts + wv * EXT_WORDS;
    for (int q = 0; q < 10; ++q) { o.X.v[q] = w[q]; o.Y.v[q] = w[10 + q]; o.Z.v[q] = w[20 + q]; o.T.v[q] = w[30 + q]; }
    if (have) ge_add(acc, acc, o); else { acc = o; have = 1; }
  }
  window_nonempty[win] = (uint32_t)have;
  if (have) store_ext(window_sums + win * EXT_WORDS, acc);
}

// ---- k_msm_finish ----------------------------------------------------------------
// Batch mode: one lane per MSM.  result = sum_t 2^(w t) W_t by Horner from the
// top window; accept = identity test & no undecodable point.
__global__ void __launch_bounds__(64)
k_msm_finish(const uint32_t* __restrict__ window_sums, const uint32_t* __restrict__ window_nonempty,
             const uint32_t* __restrict__ msm_fail, uint8_t* __restrict__ accept, uint32_t* __restrict__ out_enc,
             uint32_t* __restrict__ out_ext, uint32_t n_msm, int w, int n_windows) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_msm) return;
  ge acc;
  bool have = false;
  for (int t = n_windows - 1; t >= 0; --t) {
    if (have) {
      for (int k = 0; k < w - 1; ++k) ge_double<false>(acc, acc);
      ge_double<true>(acc, acc);
    }
    const uint64_t win = (uint64_t)m * n_windows + t;
    if (window_nonempty[win]) {
      ge p;
      load_ext(p, window_sums + win * EXT_WORDS);
      if (have) ge_add(acc, acc, p); else { acc = p; have = true; }
    }
  }
  const bool failed = msm_fail && msm_fail[m];
  if (out_ext) {
    // point mode: hand the (extended) sum to k_static_combine; identity when empty
    if (!have) ge_identity(acc);
    store_ext(out_ext + (uint64_t)m * EXT_WORDS, acc);
    accept[m] = failed ? 0 : 1;
    return;
  }
  if (out_enc) {
    // value mode: accept[m] = "every point decoded", out = canonical encoding (zeros on failure)
    uint32_t enc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (have && !failed) ristretto_encode(enc, acc);
    uint4* o = reinterpret_cast<uint4*>(out_enc + 8 * (uint64_t)m);
    o[0] = make_uint4(enc[0], enc[1], enc[2], enc[3]);
    o[1] = make_uint4(enc[4], enc[5], enc[6], enc[7]);
    accept[m] = failed ? 0 : 1;
    return;
  }
  const bool ident = have ? ge_is_identity(acc) : true;
  accept[m] = (ident && !failed) ? 1 : 0;
}

// ---- small multiscalar multiplications: k_small_tables + k_small_accumulate ------------
// Window sums of a SMALL multiscalar multiplication (the few dozen proof-specific
// points of one transaction) without the global sort.  Window width 4, so one
// wavefront's 64 lanes ARE the 64 windows of a 256-bit scalar.
//   k_small_tables      per point: P, 2P .. 8P in "cached" form (Y+X, Y-X, 2Z, 2dT), 8 x 160 B
//                       in HBM (L2-resident while in use), and the scalar recoded as
//                       s' = s + 0x88..8 (digit t = nibble t of s' minus 8: no carry chain).
//   k_small_accumulate  one workgroup of `parts` wavefronts per MSM; wavefront p takes the points
//                       p, p + parts, ...; lane t: W_t += sign(d_it) * (|d_it| P_i), one 8M
//                       addition per point, no buckets, no atomics; the parts meet in LDS.
// A lone wavefront per SIMD issues a dependent multiply chain at a fraction of the pipe's
// rate, so the point loop is split over `parts` wavefronts to keep 3-4 of them per SIMD.
struct ge_cached { fe YpX, YmX, Z2, T2d; };

__device__ __forceinline__ void ge_add_cached(ge& r, const ge& p, const ge_cached& q, bool negate) {
  fe a, b, c, d, e, f, g, h, t0, t1, qa = q.YmX, qb = q.YpX;
  fe_cswap(qa, qb, negate);
  fe_sub(t0, p.Y, p.X);
  fe_add(t1, p.Y, p.X);
  fe_mul(a, t0, qa);
  fe_mul(b, t1, qb);
  fe_mul(c, p.T, q.T2d);
  fe_mul(d, p.Z, q.Z2);
  fe_sub(e, b, a);
  fe_add(h, b, a);
  fe_sub(f, d, c);
  fe_add(g, d, c);
  fe_cswap(f, g, negate);
  fe_mul(r.X, e, f);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, f, g);
  fe_mul(r.T, e, h);
}

constexpr int SMALL_TBL = 8;   // multiples per point (signed 4-bit digits)

__device__ __forceinline__ void store_cached(uint32_t* row, const ge& p) {
  ge c4;   // cached form packed in an ext row: YpX, YmX, 2Z, 2dT
  fe_add_c(c4.X, p.Y, p.X);
  fe_sub_c(c4.Y, p.Y, p.X);
  fe_add_c(c4.Z, p.Z, p.Z);
  fe_mul(c4.T, p.T, fe_D2());
  store_ext(row, c4);
}

__global__ void __launch_bounds__(64)
k_small_tables(const uint32_t* __restrict__ dyn_scalars, const uint32_t* __restrict__ dyn_rows, uint64_t n,
               uint32_t* __restrict__ tbl /*[n][8][40]*/, uint32_t* __restrict__ recoded /*[n][8]*/,
               uint32_t* __restrict__ status) {
  const uint64_t k = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  // s' = s + 0x8888...8 (mod 2^256; a wrap leaves top nibble 0, which k_small_accumulate reads as +8).
  // dyn_scalars == NULL: the scalars are not known yet (the tables are built while the transcript is
  // still being replayed) and k_prepare writes the recoded form itself.
  if (dyn_scalars) {
    const uint32_t* sc = dyn_scalars + 8 * k;
    if (sc[7] >> 31) atomicOr(&status[0], 2u);
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t v = (uint64_t)sc[i] + 0x88888888u + carry;
      recoded[8 * k + i] = (uint32_t)v;
      carry = (uint32_t)(v >> 32);
    }
  }
  // one rolled loop (a mixed addition + the cached-form conversion): the body stays in the
  // instruction cache, where a straight-line chain of different doublings / additions does not
  ge_niels nq;
  load_niels(nq, dyn_rows + k * NIELS_WORDS);
  ge cur;
  ge_identity(cur);
  uint32_t* row = tbl + k * (SMALL_TBL * EXT_WORDS);
#pragma unroll 1
  for (int e = 0; e < SMALL_TBL; ++e) {
    ge_madd(cur, cur, nq, false);
    store_cached(row + e * EXT_WORDS, cur);
  }
}

__global__ void __launch_bounds__(256)
k_small_accumulate(const uint32_t* __restrict__ recoded, const uint64_t* __restrict__ dyn_offsets,
                   const uint32_t* __restrict__ tbl, uint32_t n_msm, uint32_t* __restrict__ window_sums,
                   uint32_t* __restrict__ window_nonempty) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];   // [parts - 1][41][64]
  const uint32_t m = blockIdx.x;
  const int t = threadIdx.x & 63;
  const int part = threadIdx.x >> 6, parts = blockDim.x >> 6;
  const uint64_t k0 = dyn_offsets[m], k1 = dyn_offsets[m + 1];
  ge total;
  bool total_set = false;
#pragma unroll 1
  for (uint64_t k = k0 + part; k < k1; k += parts) {
    const uint32_t n4 = (recoded[8 * k + (t >> 3)] >> (4 * (t & 7))) & 15u;
    const int d = (t == 63 && n4 < 8u) ? (int)n4 + 8 : (int)n4 - 8;
    if (d != 0) {
      ge c4;
      load_ext(c4, tbl + (k * SMALL_TBL + (uint64_t)((d < 0 ? -d : d) - 1)) * EXT_WORDS);
      ge_cached c;
      c.YpX = c4.X; c.YmX = c4.Y; c.Z2 = c4.Z; c.T2d = c4.T;
      if (!total_set) { ge_identity(total); total_set = true; }
      ge_add_cached(total, total, c, d < 0);
    }
  }
  if (part > 0) {
    uint32_t* slot = smem + (uint32_t)(part - 1) * 41 * 64;
    slot[40 * 64 + t] = total_set ? 1u : 0u;
    if (total_set) {
#pragma unroll
      for (int q = 0; q < 10; ++q) {
        slot[q * 64 + t] = total.X.v[q]; slot[(10 + q) * 64 + t] = total.Y.v[q];
        slot[(20 + q) * 64 + t] = total.Z.v[q]; slot[(30 + q) * 64 + t] = total.T.v[q];
      }
    }
  }
  __syncthreads();
  if (part != 0) return;
#pragma unroll 1
  for (int o = 1; o < parts; ++o) {
    const uint32_t* slot = smem + (uint32_t)(o - 1) * 41 * 64;
    if (slot[40 * 64 + t]) {
      ge other;
#pragma unroll
      for (int q = 0; q < 10; ++q) {
        other.X.v[q] = slot[q * 64 + t]; other.Y.v[q] = slot[(10 + q) * 64 + t];
        other.Z.v[q] = slot[(20 + q) * 64 + t]; other.T.v[q] = slot[(30 + q) * 64 + t];
      }
      if (total_set) ge_add(total, total, other); else { total = other; total_set = true; }
    }
  }
  const uint64_t win = (uint64_t)m * 64 + t;
  window_nonempty[win] = total_set ? 1u : 0u;
  if (total_set) store_ext(window_sums + win * EXT_WORDS, total);
}

// ---- k_msm_finish_quad -----------------------------------------------------------
// Same contract as k_msm_finish, four lanes per MSM (quad.hpp): the Horner chain
// of ~255 doublings is the latency floor of a batch, and a quad walks it ~3x faster.
__global__ void __launch_bounds__(256)
k_msm_finish_quad(const uint32_t* __restrict__ window_sums, const uint32_t* __restrict__ window_nonempty,
                  const uint32_t* __restrict__ msm_fail, uint8_t* __restrict__ accept,
                  uint32_t* __restrict__ out_enc, uint32_t* __restrict__ out_ext, uint32_t n_msm, int w,
                  int n_windows, const uint32_t* __restrict__ sel_state, uint32_t sel_group) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = g & 3;
  const bool live = (g >> 2) < n_msm;
  const uint32_t m = live ? (g >> 2) : (n_msm - 1);   // keep whole quads active for DPP
  // sel_state: only the multiscalar multiplications whose group (m / sel_group) is marked take part (the
  // transactions of the groups whose check failed); a quad leaves as a whole
  if (sel_state && sel_state[m / sel_group] == 0) return;
  ge acc;
  ge_identity(acc);
  bool have = false;
  // The sum of window t - 1 (and its flag) is fetched BEFORE the doublings of step t, unconditionally: in the chain of a lone
  // batch the two dependent loads of every step (flag, then 160 bytes) were ~1.5 us of its ~7 -- a quarter of the 0.46 ms the
  // Horner chain is of a 1024-transaction batch's 1.1 ms (round 6).  A window that is empty is loaded and not used.
  const uint64_t win0 = (uint64_t)m * n_windows;
  ge p;
  uint32_t ne = window_nonempty[win0 + n_windows - 1];
  load_ext(p, window_sums + (win0 + n_windows - 1) * EXT_WORDS);
  for (int t = n_windows - 1; t >= 0; --t) {
    ge pn;
    uint32_t nen = 0;
    if (t > 0) {
      nen = window_nonempty[win0 + t - 1];
      load_ext(pn, window_sums + (win0 + t - 1) * EXT_WORDS);
    }
    if (have) {
      for (int k = 0; k < w; ++k) quad_double(acc, r);
    }
    if (ne) {
      if (have) quad_add(acc, p, r); else { acc = p; have = true; }
    }
    if (t > 0) { p = pn; ne = nen; }
  }
  if (!live || r != 0) return;
  const bool failed = msm_fail && msm_fail[m];
  if (out_ext) {
    store_ext(out_ext + (uint64_t)m * EXT_WORDS, acc);
    if (accept) accept[m] = failed ? 0 : 1;
    return;
  }
  if (out_enc) {
    uint32_t enc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (have && !failed) ristretto_encode(enc, acc);
    uint4* o = reinterpret_cast<uint4*>(out_enc + 8 * (uint64_t)m);
    o[0] = make_uint4(enc[0], enc[1], enc[2], enc[3]);
    o[1] = make_uint4(enc[4], enc[5], enc[6], enc[7]);
    accept[m] = failed ? 0 : 1;
    return;
  }
  const bool ident = have ? ge_is_identity(acc) : true;
  accept[m] = (ident && !failed) ? 1 : 0;
}

// ---- k_from_uniform ------------------------------------------------------------
// RFC 9496 sec 4.3.4 element derivation: 64 uniform bytes -> encoding of the point
__global__ void __launch_bounds__(64)
k_from_uniform(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[16];
  const uint4* i4 = reinterpret_cast<const uint4*>(in + 16 * i);
#pragma unroll
  for (int k = 0; k < 4; ++k) { uint4 v = i4[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
  ge p;
  ristretto_from_uniform_words(p, w);
  uint32_t enc[8];
  ristretto_encode(enc, p);
  uint4* o = reinterpret_cast<uint4*>(out + 8 * i);
  o[0] = make_uint4(enc[0], enc[1], enc[2], enc[3]);
  o[1] = make_uint4(enc[4], enc[5], enc[6], enc[7]);
}

// ---- k_spin ----------------------------------------------------------------------
// One wavefront that does nothing for `cycles` shader cycles: the probe of whether two streams really run side by side
// (zkgpu.hip, streams_overlap) -- the runtime maps streams onto a limited set of hardware queues, and two streams that
// share one take turns.
__global__ void __launch_bounds__(64)
k_spin(unsigned long long cycles, uint32_t* __restrict__ sink) {
  const unsigned long long t0 = clock64();
  unsigned long long spins = 0;
  while ((unsigned long long)clock64() - t0 < cycles) ++spins;
  if (sink && spins == ~0ull) *sink = 1;
}

// ---- k_hbm_copy -----------------------------------------------------------------
// The achievable-HBM yardstick of SURVEY.md sec 8(d) ("measure achievable HBM with a copy kernel and report both"): ONE
// 16-byte vector per lane, one workgroup per 4 KiB, no loop -- the shape that reaches the 6.2-6.3 TB/s the microarchitecture
// guide quotes for a float4 copy.  Measured on an MI355X, 2 GiB, read + write bytes (tools/ubench/hbm_copy.hip,
// profiles/archive/r04_hbm_copy_variants.txt): this form 6.24 TB/s; the grid-stride loop with four loads in flight that was here
// until round 3: 4.7-4.95 TB/s whatever the grid; block-contiguous tiles of 4 / 8 vectors per lane 5.4-5.7, non-temporal
// 5.6-6.0; hipMemcpyDtoD 5.45.
__global__ void __launch_bounds__(256)
k_hbm_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n_vec) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_vec) dst[i] = src[i];
}

// per-batch scratch state in one launch (instead of five fills): status words, per-MSM failure
// flags, per-transaction wellformed flags
__global__ void __launch_bounds__(256)
k_batch_init(uint32_t* __restrict__ status, uint32_t* __restrict__ msm_fail, uint32_t* __restrict__ wellformed,
             uint32_t n_msm) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_msm) {
    msm_fail[i] = 0;
    if (wellformed) wellformed[i] = 0xffffffffu;
  }
  if (i == 0) {
    status[0] = 0; status[1] = 0;                      // flags
    status[2] = 0xffffffffu; status[3] = 0xffffffffu;  // lowest undecodable index (atomicMin)
    status[8] = 0;                                     // transactions queued for the individual re-check
    status[9] = 0;                                     // groups whose check failed
  }
}

// pack accept bytes into a bitmap (byte i/8, bit i%8)
__global__ void __launch_bounds__(256)
k_pack_bitmap(const uint8_t* __restrict__ accept, const uint32_t* __restrict__ wellformed /*optional*/,
              uint8_t* __restrict__ bitmap, uint32_t n_msm) {
  const uint32_t byte = blockIdx.x * blockDim.x + threadIdx.x;
  if (byte >= (n_msm + 7) / 8) return;
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t i = byte * 8 + k;
    if (i < n_msm && accept[i] && (!wellformed || wellformed[i])) v |= 1u << k;
  }
  bitmap[byte] = (uint8_t)v;
}

// =============================================================================
// Fixed-base path for the resident generators (BulletproofGens / PedersenGens)
// =============================================================================
// The generators never change, so the device keeps, for every window position t
// and every generator j, the affine-Niels rows of d * 2^(w t) * G_j for
// d = 1 .. 2^(w-1):
//     table[((t * n_set + j) * H + (d - 1)) * 32 words],  H = 2^(w-1)   (96-byte packed rows, one per 128-byte line)
// A generator term s * G_j then costs one mixed addition per window with NO
// doublings, no sorting and no bucket reduction, and every lane of the kernel
// does the same number of additions.  Memory is what MI355X has plenty of:
// w = 12 -> 22 * 514 * 2048 rows * 128 B = 2.96 GB for the 2-in/2-out generators.

// lane j: base[t][j] = 2^(w t) * G_j for t = 0 .. W-1 (extended, 160 B rows)
__global__ void __launch_bounds__(64)
k_tbl_base(const uint32_t* __restrict__ rows /*niels, n_set*/, uint32_t* __restrict__ base, uint32_t n_set, int w, int W) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_set) return;
  ge_niels q;
  load_niels(q, rows + (uint64_t)j * NIELS_WORDS);
  ge p;
  ge_identity(p);
  ge_madd(p, p, q, false);
  for (int t = 0; t < W; ++t) {
    store_ext(base + ((uint64_t)t * n_set + j) * EXT_WORDS, p);
    if (t + 1 < W) {
      for (int k = 0; k < w - 1; ++k) ge_double<false>(p, p);
      ge_double<true>(p, p);
    }
  }
}

// lane (t, j): multiples d * base[t][j], d = 1..H, normalised to affine with one
// inversion per lane (Montgomery's trick), written as Niels rows.
// tmp: H x 40 words per lane: X, Y, Z of d*P and the running product of the Z's.
__global__ void __launch_bounds__(64)
k_tbl_multiples(const uint32_t* __restrict__ base, uint32_t* __restrict__ tmp, uint32_t* __restrict__ table,
                uint64_t n_lanes, uint32_t H) {
  const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= n_lanes) return;
  ge p, q;
  load_ext(p, base + lane * EXT_WORDS);
  q = p;
  fe prod = fe_one();
  uint32_t* mytmp = tmp + lane * (uint64_t)H * EXT_WORDS;
  for (uint32_t d = 0; d < H; ++d) {
    if (d) ge_add(q, q, p);
    fe_mul(prod, prod, q.Z);
    ge rec;                       // reuse the ext row layout: X, Y, Z, running product
    rec.X = q.X; rec.Y = q.Y; rec.Z = q.Z; rec.T = prod;
    store_ext(mytmp + (uint64_t)d * EXT_WORDS, rec);
  }
  fe inv;
  fe_invert(inv, prod);           // 1 / (Z_1 ... Z_H)
  uint32_t* myrows = table + lane * (uint64_t)H * TABLE_STRIDE;
  for (uint32_t d = H; d-- > 0;) {
    ge rec;
    load_ext(rec, mytmp + (uint64_t)d * EXT_WORDS);
    fe prev = fe_one();
    if (d) {
      ge r2;
      load_ext(r2, mytmp + (uint64_t)(d - 1) * EXT_WORDS);
      prev = r2.T;
    }
    fe zinv, x, y;
    fe_mul(zinv, inv, prev);      // 1 / Z_d
    fe_mul(inv, inv, rec.Z);      // 1 / (Z_1 ... Z_{d-1})
    fe_mul(x, rec.X, zinv);
    fe_mul(y, rec.Y, zinv);
    ge_niels nq;
    niels_from_affine(nq, x, y);
    store_table_row(myrows + (uint64_t)d * TABLE_STRIDE, nq);
  }
}

// lane k (one static term): all W signed digits of its scalar -> digits[t * n_static + k]
__global__ void __launch_bounds__(256)
k_static_digits(const uint32_t* __restrict__ st_scalars, int16_t* __restrict__ digits, uint64_t n_static, int w,
                int W, uint32_t* __restrict__ status, const uint32_t* __restrict__ row_map /*optional*/,
                const uint32_t* __restrict__ n_active /*optional*/, uint32_t rows_per_msm, uint32_t fold_small_negatives) {
  uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_static) return;
  if (row_map) {   // only the terms of the MSMs queued in row_map (uniform rows of rows_per_msm terms)
    const uint32_t slot = (uint32_t)(k / rows_per_msm);
    if (slot >= *n_active) return;
    k = (uint64_t)row_map[slot] * rows_per_msm + k % rows_per_msm;
  }
  const uint32_t* sc = st_scalars + 8 * k;
  if (sc[7] >> 31) atomicOr(&status[0], 2u);
  for (int t = 0; t < W; ++t) digits[(uint64_t)t * n_static + k] = 0;
  // fold_small_negatives (the prover's commitment rows, whose results are ENCODED: a multiple of l added to a term changes
  // nothing there): a scalar l - v with 0 < v < 2^14 -- the a_R = -1 of every bit multiplier -- is v on the negated point:
  // one digit -v in window 0 instead of a non-zero digit in every window.
  if (fold_small_negatives) {
    const uint32_t l[8] = ZK_SC_L;
    uint32_t v[8];
    uint64_t br = 0;
    for (int i = 0; i < 8; ++i) { const uint64_t d = (uint64_t)l[i] - sc[i] - br; v[i] = (uint32_t)d; br = (d >> 32) & 1; }
    uint32_t high = v[0] >> 14;
    for (int i = 1; i < 8; ++i) high |= v[i];
    if (br == 0 && high == 0 && v[0] != 0 && w >= 15) { digits[k] = (int16_t)(-(int)v[0]); return; }
  }
  for_each_digit(sc, w, W, [&](int t, int d) { digits[(uint64_t)t * n_static + k] = (int16_t)d; });
}

// lane (tx, t, part): sum over its share of the tx's static terms of
// sign(d) * table[t][idx][|d|-1]; the next row is fetched while the current
// addition runs.  partials[((tx * W + t) * P + part)] (extended).
// kSkipZeros (the prover's rows): a zero digit loads no table row, and an addition that no lane of the wavefront needs is
// skipped; the verifier's scalars have no zeros to speak of and keep the plain loop.
template <bool kSkipZeros>
__global__ void __launch_bounds__(256)
k_static_accumulate(const int16_t* __restrict__ digits, const uint64_t* __restrict__ st_offsets,
                    const uint32_t* __restrict__ st_index, const uint32_t* __restrict__ table, uint32_t n_set,
                    uint32_t H, int W, int P, uint32_t n_msm, uint64_t n_static, uint32_t* __restrict__ partials,
                    const uint32_t* __restrict__ row_map /*optional: slot -> MSM*/,
                    const uint32_t* __restrict__ n_active /*optional: slots in use, device side*/,
                    uint32_t interleave /*1, or the rows come in groups of this many kinds (the prover's A_I A_O S per proof)*/) {
  const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= (uint64_t)n_msm * W * P) return;
  // window-major lane order: the chip sweeps the table one window slice at a time
  const uint32_t part = (uint32_t)(lane % P);
  uint32_t slot = (uint32_t)((lane / P) % n_msm);
  const uint32_t t = (uint32_t)(lane / ((uint64_t)P * n_msm));
  if (n_active && slot >= *n_active) return;
  // rows of one KIND side by side in a wavefront: their zero digits are in the same places (the a_O = 0 of every bit
  // multiplier, the windows above the first of a_L = 0 / 1), and an addition that no lane needs is skipped below
  if (interleave > 1) { const uint32_t per = n_msm / interleave; slot = (slot % per) * interleave + slot / per; }
  const uint32_t tx = row_map ? row_map[slot] : slot;
  const uint64_t k0 = st_offsets[tx], k1 = st_offsets[tx + 1];
  const int16_t* dig = digits + (uint64_t)t * n_static;
  const uint64_t tbase = (uint64_t)t * n_set;
  ge acc;
  ge_identity(acc);
  auto fetch = [&](uint64_t k, ge_niels& q, bool& neg, bool& any) {
    int d = dig[k];
    d = (d == -32768) ? 32768 : d;   // w = 16: +2^15 is stored wrapped (the recoding never yields -2^15)
    any = d != 0;
    const uint32_t idx = st_index ? st_index[k] : (uint32_t)(k - k0);
    const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
    if (!kSkipZeros || any) load_table_row(q, table + ((tbase + idx) * H + (mag ? mag - 1 : 0)) * TABLE_STRIDE);
    if (!any) niels_identity(q);
    neg = d < 0;
  };
  uint64_t k = k0 + part;
  ge_niels cur, nxt;
  bool cur_neg = false, nxt_neg = false, cur_any = false, nxt_any = false;
  if (k < k1) fetch(k, cur, cur_neg, cur_any);
  while (k < k1) {
    const uint64_t kn = k + P;
    if (kn < k1) fetch(kn, nxt, nxt_neg, nxt_any);
    if (!kSkipZeros || __any(cur_any)) ge_madd(acc, acc, cur, cur_neg);      // (a lane whose digit is zero adds the identity)
    cur = nxt; cur_neg = nxt_neg; cur_any = nxt_any;
    k = kn;
  }
  store_ext(partials + (((uint64_t)slot * W + t) * P + part) * EXT_WORDS, acc);
}

// one wave per tx: sum the W*P static partials and the dynamic-term sum, then
// the ristretto identity test.  Lane sums are folded with wavefront shuffles.
// With row_map / n_active (the per-transaction re-check of failed groups) block b handles
// MSM row_map[b]; its partials sit at slot b.
// (one wavefront; `lane` = its lane)
__device__ inline void static_combine_slot(const uint32_t* __restrict__ partials, uint32_t n_partials, const uint32_t* __restrict__ dyn_sum,
                                           const uint8_t* __restrict__ dyn_ok, uint32_t slot, uint32_t tx, int lane,
                                           uint8_t* __restrict__ accept, uint32_t* __restrict__ out_points) {
  ge acc;
  ge_identity(acc);
  for (uint32_t c = lane; c < n_partials; c += 64) {
    ge p;
    load_ext(p, partials + ((uint64_t)slot * n_partials + c) * EXT_WORDS);
    ge_add(acc, acc, p);
  }
  if (lane == 0 && dyn_sum) {
    ge p;
    load_ext(p, dyn_sum + (uint64_t)tx * EXT_WORDS);
    ge_add(acc, acc, p);
  }
#pragma unroll 1
  for (int delta = 32; delta >= 1; delta >>= 1) {
    ge other;
    shfl_down_ge(other, acc, delta);
    if (lane < delta) ge_add(acc, acc, other);
  }
  if (lane == 0) {
    accept[tx] = (ge_is_identity(acc) && (!dyn_ok || dyn_ok[tx])) ? 1 : 0;
    if (out_points) store_ext(out_points + (uint64_t)slot * EXT_WORDS, acc);
  }
}

// COMBINE_LANES lanes per slot, eight slots per wavefront: a slot's W x P partial sums (19 for a payment over 14-bit tables) are
// three additions per lane and three folds.  (Until round 6 a wavefront per slot: 19 of 64 lanes with a partial sum to load, then
// six folds at 32, 16, .. 1 live lanes -- 0.17 of its lanes at work, 7 % of a transaction call's instructions:
// profiles/archive/r05z_lane_utilization_per_kernel.txt; the rework k_static_row_sums got in round 5.)
constexpr uint32_t COMBINE_LANES = 8;

__global__ void __launch_bounds__(64)
k_static_combine(const uint32_t* __restrict__ partials, uint32_t n_partials, const uint32_t* __restrict__ dyn_sum,
                 const uint8_t* __restrict__ dyn_ok, const uint32_t* __restrict__ row_map,
                 const uint32_t* __restrict__ n_active, uint32_t n_slots, uint8_t* __restrict__ accept,
                 uint32_t* __restrict__ out_points /*optional: the sum of slot b, extended, for k_locate_finish*/) {
  const uint32_t n = n_active ? min(*n_active, n_slots) : n_slots;
  const uint32_t g = blockIdx.x * 64 + threadIdx.x, sub = g % COMBINE_LANES, first = (blockIdx.x * 64) / COMBINE_LANES;
  if (first >= n) return;                                  // (the whole wavefront: its slots are first .. first + 7)
  const uint32_t mine = g / COMBINE_LANES, slot = mine < n ? mine : n - 1;      // (whole wavefronts stay active for the shuffles)
  const uint32_t tx = row_map ? row_map[slot] : slot;
  ge acc;
  ge_identity(acc);
  for (uint32_t c = sub; c < n_partials; c += COMBINE_LANES) {
    ge p;
    load_ext(p, partials + ((uint64_t)slot * n_partials + c) * EXT_WORDS);
    ge_add(acc, acc, p);
  }
  if (sub == 0 && dyn_sum) {
    ge p;
    load_ext(p, dyn_sum + (uint64_t)tx * EXT_WORDS);
    ge_add(acc, acc, p);
  }
#pragma unroll 1
  for (int delta = COMBINE_LANES / 2; delta >= 1; delta >>= 1) {
    ge other;
    shfl_down_ge(other, acc, delta);
    if (sub < (uint32_t)delta) ge_add(acc, acc, other);
  }
  if (sub == 0 && mine < n) {
    accept[tx] = (ge_is_identity(acc) && (!dyn_ok || dyn_ok[tx])) ? 1 : 0;
    if (out_points) store_ext(out_points + (uint64_t)slot * EXT_WORDS, acc);
  }
}

// One thread's share of ONE row's fixed-base sum: the (window, term) pairs i = tid, tid + n_threads, .. of the W x ns
// pairs of the row whose terms start at k0 -- the next table row is fetched while the current addition runs.  For the rows
// that are summed on the tail of a batch by a single workgroup (k_locate_fused, k_recheck_fused), where a launch of its own
// for the partial sums costs more than the sums.
__device__ inline void static_row_share(ge& acc, const int16_t* __restrict__ digits, uint64_t n_static_total, uint64_t k0, uint32_t ns,
                                        const uint32_t* __restrict__ st_index, const uint32_t* __restrict__ table, uint32_t n_set,
                                        uint32_t H, int W, uint32_t tid, uint32_t n_threads) {
  ge_identity(acc);
  const uint32_t items = (uint32_t)W * ns;
  auto fetch = [&](uint32_t i, ge_niels& q, bool& neg) {
    const uint32_t t = i / ns, j = i - t * ns;
    const uint64_t k = k0 + j;
    int d = digits[(uint64_t)t * n_static_total + k];
    d = (d == -32768) ? 32768 : d;
    const uint32_t idx = st_index ? st_index[k] : j;
    const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
    load_table_row(q, table + (((uint64_t)t * n_set + idx) * H + (mag ? mag - 1 : 0)) * TABLE_STRIDE);
    if (d == 0) niels_identity(q);
    neg = d < 0;
  };
  uint32_t i = tid;
  ge_niels cur, nxt;
  bool cur_neg = false, nxt_neg = false;
  if (i < items) fetch(i, cur, cur_neg);
  while (i < items) {
    const uint32_t in = i + n_threads;
    if (in < items) fetch(in, nxt, nxt_neg);
    ge_madd(acc, acc, cur, cur_neg);
    cur = nxt; cur_neg = nxt_neg;
    i = in;
  }
}

// The individual re-check of the queued transactions in ONE launch: workgroup `slot` sums the generator terms of
// transaction row_map[slot] (256 shares), then its first wavefront adds the proof-point sum and tests for the identity --
// k_static_accumulate (row_map) + k_static_combine in one, for the tail of a batch.
__global__ void __launch_bounds__(512)
k_recheck_fused(const int16_t* __restrict__ digits, const uint64_t* __restrict__ st_offsets, const uint32_t* __restrict__ st_index,
                const uint32_t* __restrict__ table, uint32_t n_set, uint32_t H, int W, uint64_t n_static_total,
                uint32_t* __restrict__ partials /*[slots][blockDim][40]*/, const uint32_t* __restrict__ dyn_sum,
                const uint8_t* __restrict__ dyn_ok, const uint32_t* __restrict__ row_map, const uint32_t* __restrict__ n_active,
                uint8_t* __restrict__ accept, uint32_t* __restrict__ out_points) {
  const uint32_t slot = blockIdx.x;
  if (slot >= *n_active) return;
  const uint32_t tx = row_map[slot];
  const uint64_t k0 = st_offsets[tx];
  ge acc;
  static_row_share(acc, digits, n_static_total, k0, (uint32_t)(st_offsets[tx + 1] - k0), st_index, table, n_set, H, W, threadIdx.x, blockDim.x);
  store_ext(partials + ((uint64_t)slot * blockDim.x + threadIdx.x) * EXT_WORDS, acc);
  __threadfence_block();
  __syncthreads();
  if (threadIdx.x < 64) static_combine_slot(partials, blockDim.x, dyn_sum, dyn_ok, slot, tx, (int)threadIdx.x, accept, out_points);
}

// one wave per MSM over the resident set: sum of its W*P table partials -> canonical ristretto
// encoding (the value of a Pedersen vector commitment: the prover-side use of the tables)
__global__ void __launch_bounds__(64)
k_static_values(const uint32_t* __restrict__ partials, uint32_t n_partials, uint32_t* __restrict__ out_enc) {
  const uint32_t m = blockIdx.x;
  const int lane = threadIdx.x;
  ge acc;
  ge_identity(acc);
  for (uint32_t c = lane; c < n_partials; c += 64) {
    ge p;
    load_ext(p, partials + ((uint64_t)m * n_partials + c) * EXT_WORDS);
    ge_add(acc, acc, p);
  }
#pragma unroll 1
  for (int delta = 32; delta >= 1; delta >>= 1) {
    ge other;
    shfl_down_ge(other, acc, delta);
    if (lane < delta) ge_add(acc, acc, other);
  }
  if (lane == 0) {
    uint32_t enc[8];
    ristretto_encode(enc, acc);
    uint4* o = reinterpret_cast<uint4*>(out_enc + 8 * (uint64_t)m);
    o[0] = make_uint4(enc[0], enc[1], enc[2], enc[3]);
    o[1] = make_uint4(enc[4], enc[5], enc[6], enc[7]);
  }
}

// The same in two steps for many rows (the prover's phases: thousands of rows per call): the sums as extended points,
// one wave per row -- then the encodings one LANE per row, so that the inverse square root of RFC 9496's ENCODE (the
// long dependent chain of this step) runs on 64 rows per wavefront instead of on one lane of each
// (EIGHT lanes per row, eight rows per wavefront: a row has 32 .. 48 partial sums; with a wavefront per row half the lanes
// had nothing to add and the tree of six halving steps ran at 15 % of its lanes -- 12 % of a proving call's instructions)
constexpr uint32_t ROW_SUM_LANES = 8;
__global__ void __launch_bounds__(64)
k_static_row_sums(const uint32_t* __restrict__ partials, uint32_t n_partials, uint32_t n_rows, uint32_t* __restrict__ out_ext) {
  const uint32_t g = blockIdx.x * 64 + threadIdx.x, sub = g % ROW_SUM_LANES;
  const uint32_t row = g / ROW_SUM_LANES, m = row < n_rows ? row : n_rows - 1;      // (whole wavefronts stay active for the shuffles)
  ge acc;
  ge_identity(acc);
  for (uint32_t c = sub; c < n_partials; c += ROW_SUM_LANES) {
    ge p;
    load_ext(p, partials + ((uint64_t)m * n_partials + c) * EXT_WORDS);
    ge_add(acc, acc, p);
  }
#pragma unroll 1
  for (int delta = ROW_SUM_LANES / 2; delta >= 1; delta >>= 1) {
    ge other;
    shfl_down_ge(other, acc, delta);
    if (sub < (uint32_t)delta) ge_add(acc, acc, other);
  }
  if (sub == 0 && row < n_rows) store_ext(out_ext + (uint64_t)row * EXT_WORDS, acc);
}
__global__ void __launch_bounds__(64)
k_encode_rows(const uint32_t* __restrict__ ext, uint32_t n_rows, uint32_t* __restrict__ out_enc) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_rows) return;
  ge p;
  load_ext(p, ext + (uint64_t)m * EXT_WORDS);
  uint32_t enc[8];
  ristretto_encode(enc, p);
  uint4* o = reinterpret_cast<uint4*>(out_enc + 8 * (uint64_t)m);
  o[0] = make_uint4(enc[0], enc[1], enc[2], enc[3]);
  o[1] = make_uint4(enc[4], enc[5], enc[6], enc[7]);
}

// ---- merging queued batches (session.hpp, tickets): ONE launch copies the three input buffers of up to 16 batches side
// by side into a context's workspace (instead of three device-to-device copies per batch: the host time of those calls
// is what delays the launch of a merged batch)
struct MergeSources {
  const uint4* com[16];
  const uint8_t* proofs[16];
  const uint4* r[16];
  uint32_t first[17];        // first transaction of each source in the merged batch; first[n] = total
  uint32_t n;
};
__global__ void __launch_bounds__(256)
k_merge_inputs(MergeSources src, uint32_t com_vec /*uint4 per transaction*/, uint32_t proof_len, uint4* __restrict__ com,
               uint8_t* __restrict__ proofs, uint4* __restrict__ r) {
  const uint32_t total = src.first[src.n];
  const uint64_t n_com = (uint64_t)total * com_vec, n_r = (uint64_t)total * 4, n_pw = ((uint64_t)total * proof_len + 3) / 4;
  const uint64_t g0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
  auto source_of = [&src](uint32_t tx) { uint32_t k = 0; while (k + 1 < src.n && tx >= src.first[k + 1]) ++k; return k; };
  for (uint64_t g = g0; g < n_com; g += stride) {
    const uint32_t tx = (uint32_t)(g / com_vec), k = source_of(tx);
    com[g] = src.com[k][g - (uint64_t)src.first[k] * com_vec];
  }
  for (uint64_t g = g0; g < n_r; g += stride) {
    const uint32_t tx = (uint32_t)(g / 4), k = source_of(tx);
    r[g] = src.r[k][g - (uint64_t)src.first[k] * 4];
  }
  // proofs: byte-aligned sources (a proof is 1 + 32 k bytes): four bytes per lane of the destination -- from two ALIGNED
  // words of the source and a funnel shift wherever the four bytes lie inside one source and not at its very end (a word
  // read there could reach past the caller's buffer); byte by byte at the seams.  (Until round 4 always byte by byte, one
  // thread per word: 3.5 M threads for a device batch of 10 240 -- beside a chip-filling kernel its 13 700 workgroups
  // trickled in over 0.55 ms and held up the batch's head; now a bounded grid that is resident at once, grid-stride.)
  const uint64_t n_bytes = (uint64_t)total * proof_len;
  for (uint64_t g = g0; g < n_pw; g += stride) {
    const uint64_t at0 = 4 * g;
    const uint32_t tx0 = (uint32_t)(at0 / proof_len), k0 = source_of(tx0);
    const uint64_t lo_b = (uint64_t)src.first[k0] * proof_len, hi_b = (uint64_t)src.first[k0 + 1] * proof_len;
    uint32_t w = 0;
    if (at0 >= lo_b + 4 && at0 + 12 <= hi_b) {            // all four bytes in source k0, a word of it before and two after (aligned reads stay inside)
      const uint8_t* p = src.proofs[k0] + (at0 - lo_b);
      const uintptr_t a = (uintptr_t)p & ~(uintptr_t)3;
      const uint32_t sh = (uint32_t)((uintptr_t)p & 3);
      const uint32_t w0 = *reinterpret_cast<const uint32_t*>(a), w1 = *reinterpret_cast<const uint32_t*>(a + 4);
      w = __builtin_amdgcn_alignbyte(w1, w0, sh);
    } else {
      for (int b = 0; b < 4; ++b) {
        const uint64_t at = at0 + b;
        if (at >= n_bytes) break;
        const uint32_t tx = (uint32_t)(at / proof_len), k = source_of(tx);
        w |= (uint32_t)src.proofs[k][at - (uint64_t)src.first[k] * proof_len] << (8 * b);
      }
    }
    reinterpret_cast<uint32_t*>(proofs)[g] = w;
  }
}

// ---- group checks ---------------------------------------------------------------------
// A group of transactions whose equations E_t are weighted by independent random rho's (k_transcript)
// sums to the identity iff every one of them does (up to probability ~2^-250), and the generator
// terms of the sum collapse into ONE set of n_static scalars for the whole group: g times fewer
// table gathers.  Transactions already known to be bad when the sums are formed (an undecodable proof
// point, a malformed proof) are left out of their group and rejected on the spot.
//
// A group that fails is not re-checked transaction by transaction (g multiscalar multiplications) but
// LOCATED with one more: with S1 = sum_t E_t and S2 = sum_t i_t E_t (i_t = 1, 2, .. the position in the
// group), a single bad transaction b gives S2 = i_b S1, and i_b is found by trying the g candidates.
// Transaction b alone is then checked on its own (E_b != 0: rejected), and the others are accepted iff
// S1 - E_b is the identity -- the very group check, restricted to them.  Only when no i fits (two or more
// bad transactions in one group) are all of its transactions checked individually.  Every accept is
// therefore still backed by a random-weighted identity test, every reject by a non-zero equation.
// (If S1 != E_b although an i fitted -- probability ~2^-248 -- the batch is flagged in status[0] bit 2 and
// the host re-runs it with every transaction on its own.)
constexpr uint32_t LOCATE_NONE = 0xffffffffu;

__device__ __forceinline__ bool tx_excluded(const uint32_t* __restrict__ msm_fail, const uint32_t* __restrict__ wellformed, uint32_t tx) {
  return (msm_fail && msm_fail[tx]) || (wellformed && !wellformed[tx]);
}

// k * p for a small k (< 2^bits): double-and-add from the top bit, the same instruction stream on every lane
__device__ inline void ge_small_mul(ge& r, const ge& p, uint32_t k, int bits) {
  ge_identity(r);
#pragma unroll 1
  for (int bit = bits - 1; bit >= 0; --bit) {
    ge d;
    ge_double<true>(d, r);
    r = d;
    ge s;
    ge_add(s, r, p);
    if ((k >> bit) & 1) r = s;
  }
}

// RFC 9496 sec 4.3.3 equality of the ristretto elements two curve points represent
__device__ inline bool ge_ristretto_eq(const ge& p, const ge& q) {
  fe a, b, c, d;
  fe_mul(a, p.X, q.Y);
  fe_mul(b, p.Y, q.X);
  fe_mul(c, p.Y, q.Y);
  fe_mul(d, p.X, q.X);
  return fe_eq(a, b) | fe_eq(c, d);
}

__device__ __forceinline__ void shfl_ge(ge& out, const ge& in, int src_lane) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    out.X.v[i] = __shfl(in.X.v[i], src_lane);
    out.Y.v[i] = __shfl(in.Y.v[i], src_lane);
    out.Z.v[i] = __shfl(in.Z.v[i], src_lane);
    out.T.v[i] = __shfl(in.T.v[i], src_lane);
  }
}

// window sums of a GROUP: lane (G, t) adds window t of the group's transactions (those not left out), so that the
// Horner chain over the windows -- the longest dependent chain of a batch, and a third of its point doublings -- runs
// once per group instead of once per transaction; only the transactions of a group whose check FAILS get chains of
// their own afterwards (k_msm_finish_quad with sel_state)
__global__ void __launch_bounds__(256)
k_group_windows(const uint32_t* __restrict__ window_sums, const uint32_t* __restrict__ window_nonempty,
                const uint32_t* __restrict__ msm_fail, const uint32_t* __restrict__ wellformed, uint32_t n_msm,
                uint32_t group, uint32_t n_windows, uint32_t* __restrict__ out_sums, uint32_t* __restrict__ out_nonempty) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_groups = (n_msm + group - 1) / group;
  if (g >= n_groups * n_windows) return;
  const uint32_t G = g / n_windows, t = g % n_windows;
  ge acc;
  bool have = false;
#pragma unroll 1
  for (uint32_t i = 0; i < group; ++i) {
    const uint32_t tx = G * group + i;
    if (tx >= n_msm) break;
    if (tx_excluded(msm_fail, wellformed, tx)) continue;
    const uint64_t win = (uint64_t)tx * n_windows + t;
    if (!window_nonempty[win]) continue;
    ge p;
    load_ext(p, window_sums + win * EXT_WORDS);
    if (have) ge_add(acc, acc, p); else { acc = p; have = true; }
  }
  out_nonempty[g] = have ? 1u : 0u;
  if (have) store_ext(out_sums + (uint64_t)g * EXT_WORDS, acc);
}

__device__ inline void locate_group(const uint32_t* __restrict__ partials2, uint32_t n_partials, const uint32_t* __restrict__ dyn_sum,
                                    const uint32_t* __restrict__ msm_fail, const uint32_t* __restrict__ wellformed, uint32_t n_msm,
                                    uint32_t group, uint32_t G, uint32_t f, const uint32_t* __restrict__ fail_sum,
                                    uint32_t* __restrict__ row_map, uint32_t* __restrict__ n_recheck, uint32_t* __restrict__ cand,
                                    const uint32_t* __restrict__ st_scalars, uint32_t n_static, int16_t* __restrict__ digits, int w, int W,
                                    unsigned long long* sh_queue);

// one wave per group: generator partials of the group + the proof-point sums of its transactions
// blockDim = 256: wavefront 0 sums the points; a failed group's locating scalars are then spread over all four.
// locate == 0 (small batches, where the extra stage costs more latency than it saves work): a failed group queues
// all of its transactions for the individual re-check right here (cand = LOCATE_NONE).
// kSpec: the instantiation that also knows how to locate (locate == 2); kept apart because the locating code triples the
// kernel's registers (463 against ~150: one wavefront per SIMD, a whole CU per group) for a mode that is off by default
template <bool kSpec>
__global__ void __launch_bounds__(256)
k_group_combine(const uint32_t* __restrict__ partials, uint32_t n_partials, const uint32_t* __restrict__ grp_dyn /*[n_groups][40], or null: sum dyn_sum over the group*/,
                const uint32_t* __restrict__ dyn_sum, const uint32_t* __restrict__ msm_fail, const uint32_t* __restrict__ wellformed, uint32_t n_msm,
                uint32_t group, uint8_t* __restrict__ accept, uint32_t* __restrict__ grp_state /*[n_groups]: 0 passed, f + 1 failed*/,
                uint32_t* __restrict__ fail_list, uint32_t* __restrict__ fail_sum /*[n_groups][40]*/, uint32_t* __restrict__ n_fail,
                const uint32_t* __restrict__ st_scalars, uint32_t n_static, uint32_t* __restrict__ loc_sc /*[n_groups][n_static][8]*/,
                int16_t* __restrict__ loc_digits, int w, int W, uint32_t locate /*0 no, 1 by a multiplication to come, 2 its rows are there*/,
                uint32_t* __restrict__ row_map, uint32_t* __restrict__ n_recheck, uint32_t* __restrict__ cand,
                int16_t* __restrict__ rechk_digits /*locate == 2: [W][n_msm * n_static]*/) {
  __shared__ uint32_t sh_fail;           // 0: the group passed, else f + 1
  __shared__ unsigned long long sh_queue;
  const uint32_t G = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63;
  const uint32_t in_group = min(group, n_msm - G * group);
  if (t < 64) {
    ge acc;
    ge_identity(acc);
    for (uint32_t c = lane; c < n_partials; c += 64) {
      ge p;
      load_ext(p, partials + ((uint64_t)G * n_partials + c) * EXT_WORDS);
      ge_add(acc, acc, p);
    }
    if (grp_dyn) {                        // the proof-point sum of the whole group (k_group_windows + one Horner chain)
      if (lane == 0) {
        ge p;
        load_ext(p, grp_dyn + (uint64_t)G * EXT_WORDS);
        ge_add(acc, acc, p);
      }
    } else {
      for (uint32_t i = lane; i < group; i += 64) {
        const uint32_t tx = G * group + i;
        if (tx < n_msm && !tx_excluded(msm_fail, wellformed, tx)) {
          ge p;
          load_ext(p, dyn_sum + (uint64_t)tx * EXT_WORDS);
          ge_add(acc, acc, p);
        }
      }
    }
#pragma unroll 1
    for (int delta = 32; delta >= 1; delta >>= 1) {
      ge other;
      shfl_down_ge(other, acc, delta);
      if (lane < delta) ge_add(acc, acc, other);
    }
    const int ok = __shfl((lane == 0 && ge_is_identity(acc)) ? 1 : 0, 0);
    for (uint32_t i = lane; i < in_group; i += 64) {
      const uint32_t tx = G * group + i;
      accept[tx] = (ok && !tx_excluded(msm_fail, wellformed, tx)) ? 1 : 0;
    }
    if (lane == 0) {
      uint32_t f1 = 0;
      if (!ok) {
        const uint32_t f = atomicAdd(n_fail, 1u);
        fail_list[f] = G;
        store_ext(fail_sum + (uint64_t)f * EXT_WORDS, acc);
        f1 = f + 1;
      }
      grp_state[G] = f1;
      sh_fail = f1;
    }
    if (!ok && !locate) {                 // every transaction of the group, one by one
      unsigned long long lives = 0;
      for (uint32_t i0 = 0; i0 < in_group; i0 += 64) {
        const uint32_t i = i0 + lane, tx = G * group + i;
        const bool live = i < in_group && !tx_excluded(msm_fail, wellformed, tx);
        lives = __ballot(live);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(n_recheck, (uint32_t)__popcll(lives));
        base = __shfl(base, 0);
        if (live) row_map[base + (uint32_t)__popcll(lives & ((1ull << lane) - 1))] = tx;
      }
    }
  }
  __syncthreads();
  const uint32_t f1 = sh_fail;
  if (f1 == 0) return;
  const uint32_t f = f1 - 1;
  if (!locate) { if (t == 0) cand[f] = LOCATE_NONE; return; }
  if (kSpec && locate == 2) {
    // the locating sums of ALL groups were formed beside the group sums (rows n_groups + G of the same multiplication):
    // the culprit is named right here, two dependent launches earlier
    const uint32_t n_groups2 = (n_msm + group - 1) / group;
    locate_group(partials + (uint64_t)(n_groups2 + G) * n_partials * EXT_WORDS, n_partials, dyn_sum, msm_fail, wellformed, n_msm, group, G, f,
                 fail_sum, row_map, n_recheck, cand, st_scalars, n_static, rechk_digits, w, W, &sh_queue);
    return;
  }
  // failed: the generator scalars of the group's LOCATING sum, sum_t i_t s_(t,j) (i_t = 1, 2, .. the position in the
  // group), and their digits, stored at the group's place in the failed list
  const uint32_t n_groups = (n_msm + group - 1) / group;
  const uint64_t stride = (uint64_t)n_groups * n_static;
  for (uint32_t j = (uint32_t)t; j < n_static; j += blockDim.x) {
    scm run = scm_zero(), sum = scm_zero();
    for (uint32_t i = in_group; i-- > 0;) {        // sum_k (sum_{t >= k} s_t) = sum_t (t + 1) s_t
      const uint32_t tx = G * group + i;
      if (!tx_excluded(msm_fail, wellformed, tx)) {
        const uint4* src = reinterpret_cast<const uint4*>(st_scalars + ((uint64_t)tx * n_static + j) * 8);
        const uint4 a = src[0], b = src[1];
        scm v;
        v.v[0] = a.x; v.v[1] = a.y; v.v[2] = a.z; v.v[3] = a.w; v.v[4] = b.x; v.v[5] = b.y; v.v[6] = b.z; v.v[7] = b.w;
        run = scm_add(run, v);
      }
      sum = scm_add(sum, run);
    }
    uint32_t* o = loc_sc + ((uint64_t)f * n_static + j) * 8;
    uint4* dst = reinterpret_cast<uint4*>(o);
    dst[0] = make_uint4(sum.v[0], sum.v[1], sum.v[2], sum.v[3]);
    dst[1] = make_uint4(sum.v[4], sum.v[5], sum.v[6], sum.v[7]);
    const uint64_t g = (uint64_t)f * n_static + j;
    for (int tt = 0; tt < W; ++tt) loc_digits[(uint64_t)tt * stride + g] = 0;
    for_each_digit(o, w, W, [&](int tt, int d) { loc_digits[(uint64_t)tt * stride + g] = (int16_t)d; });
  }
}

// Failed group f (= group G): S2 = generator partials of the locating sum (`partials2`: its n_partials rows) + sum_t i_t dyn_t;
// the position i with i S1 = S2 names the one bad transaction, which alone is queued for the individual check (cand[f] = its
// slot in row_map); no such i: every transaction of the group is queued (cand[f] = LOCATE_NONE).  Called by all 256 threads of
// a workgroup: wavefront 0 locates, then everybody writes the digits of the queued transactions' generator scalars.
__device__ inline void locate_group(const uint32_t* __restrict__ partials2, uint32_t n_partials, const uint32_t* __restrict__ dyn_sum,
                                    const uint32_t* __restrict__ msm_fail, const uint32_t* __restrict__ wellformed, uint32_t n_msm,
                                    uint32_t group, uint32_t G, uint32_t f, const uint32_t* __restrict__ fail_sum,
                                    uint32_t* __restrict__ row_map, uint32_t* __restrict__ n_recheck, uint32_t* __restrict__ cand,
                                    const uint32_t* __restrict__ st_scalars, uint32_t n_static, int16_t* __restrict__ digits, int w, int W,
                                    unsigned long long* sh_queue) {
  const int t = threadIdx.x, lane = t & 63;
  if (t < 64) {
    ge acc;
    ge_identity(acc);
    for (uint32_t c = lane; c < n_partials; c += 64) {
      ge p;
      load_ext(p, partials2 + (uint64_t)c * EXT_WORDS);
      ge_add(acc, acc, p);
    }
    const uint32_t tx = G * group + (uint32_t)lane;
    const bool live = (uint32_t)lane < group && tx < n_msm && !tx_excluded(msm_fail, wellformed, tx);
    // S2 = sum_lanes share + sum_k k d_k (k = lane + 1, d_k the proof-point sum of transaction k of the group), and the
    // candidates M_k = k S1.  Neither needs a multiplication: sum_k k d_k = sum_j (sum_{k >= j} d_k) is the sum of the SUFFIX
    // sums, and k S1 is the PREFIX sum of S1 over the lanes -- log2(group) shuffled additions each, where round 3 ran a
    // five-step double-and-add on two chains per lane (seven points live at once: 1084 bytes of scratch in this function;
    // the kernel sits on the tail of every batch that has a failed group).
    const int span = 1 << (31 - __clz((int)(2 * group - 1)));      // group (1 .. 64) rounded up to a power of two
    {
      ge suf;
      ge_identity(suf);
      if (live) load_ext(suf, dyn_sum + (uint64_t)tx * EXT_WORDS);
#pragma unroll 1
      for (int delta = 1; delta < span; delta <<= 1) {
        ge other;
        shfl_down_ge(other, suf, delta);           // (lanes >= group hold the identity)
        if (lane + delta < span) ge_add(suf, suf, other);
      }
      if (lane < span) ge_add(acc, acc, suf);
    }
#pragma unroll 1
    for (int delta = 32; delta >= 1; delta >>= 1) {
      ge other;
      shfl_down_ge(other, acc, delta);
      if (lane < delta) ge_add(acc, acc, other);
    }
    ge M;
    load_ext(M, fail_sum + (uint64_t)f * EXT_WORDS);  // S1 in every lane -> (lane + 1) S1 by an inclusive scan
#pragma unroll 1
    for (int delta = 1; delta < span; delta <<= 1) {
      ge other;
      shfl_up_ge(other, M, delta);
      if (lane >= delta) ge_add(M, M, other);
    }
    ge S2;
    shfl_ge(S2, acc, 0);
    const bool match = live && ge_ristretto_eq(M, S2);
    const unsigned long long hits = __ballot(match);
    const unsigned long long lives = __ballot(live);
    if (hits) {
      const int b = __ffsll((long long)hits) - 1;
      if (lane == 0) {
        const uint32_t slot = atomicAdd(n_recheck, 1u);
        row_map[slot] = G * group + (uint32_t)b;
        cand[f] = slot;
        *sh_queue = 1ull << b;
      }
    } else {
      uint32_t base = 0;
      if (lane == 0) { base = atomicAdd(n_recheck, (uint32_t)__popcll(lives)); cand[f] = LOCATE_NONE; *sh_queue = lives; }
      base = __shfl(base, 0);
      if (live) row_map[base + (uint32_t)__popcll(lives & ((1ull << lane) - 1))] = tx;
    }
  }
  __syncthreads();
  const unsigned long long queue = *sh_queue;
  for (uint32_t i = 0; i < group && i < 64; ++i) {
    if (!((queue >> i) & 1)) continue;
    const uint32_t ti = G * group + i;
    for (uint32_t j = (uint32_t)t; j < n_static; j += blockDim.x) {
      const uint64_t g = (uint64_t)ti * n_static + j;
      for (int tt = 0; tt < W; ++tt) digits[(uint64_t)tt * n_msm * n_static + g] = 0;
      for_each_digit(st_scalars + 8 * g, w, W, [&](int tt, int dd) { digits[(uint64_t)tt * n_msm * n_static + g] = (int16_t)dd; });
    }
  }
}

// one workgroup per failed group f, after the locating multiscalar multiplication (rows indexed by f)
__global__ void __launch_bounds__(256)
k_locate_combine(const uint32_t* __restrict__ partials, uint32_t n_partials, const uint32_t* __restrict__ dyn_sum,
                 const uint32_t* __restrict__ msm_fail, const uint32_t* __restrict__ wellformed, uint32_t n_msm,
                 uint32_t group, const uint32_t* __restrict__ fail_list, const uint32_t* __restrict__ n_fail,
                 const uint32_t* __restrict__ fail_sum, uint32_t* __restrict__ row_map, uint32_t* __restrict__ n_recheck,
                 uint32_t* __restrict__ cand, const uint32_t* __restrict__ st_scalars, uint32_t n_static,
                 int16_t* __restrict__ digits /*[W][n_msm * n_static], as k_static_digits writes them*/, int w, int W) {
  __shared__ unsigned long long sh_queue;    // bit i: transaction i of the group is queued for the individual check
  const uint32_t f = blockIdx.x;
  if (f >= *n_fail) return;
  locate_group(partials + (uint64_t)f * n_partials * EXT_WORDS, n_partials, dyn_sum, msm_fail, wellformed, n_msm, group, fail_list[f], f,
               fail_sum, row_map, n_recheck, cand, st_scalars, n_static, digits, w, W, &sh_queue);
}

// The locating multiplication and k_locate_combine in ONE launch: workgroup f sums the locating scalars' generator terms
// itself (256 shares, digits at the group's place f in the failed list) and goes on to name the culprit.
__global__ void __launch_bounds__(512)
k_locate_fused(const int16_t* __restrict__ loc_digits, const uint64_t* __restrict__ st_offsets, const uint32_t* __restrict__ st_index,
               const uint32_t* __restrict__ table, uint32_t n_set, uint32_t H, uint32_t n_groups,
               uint32_t* __restrict__ partials /*[n_groups][blockDim][40]*/, const uint32_t* __restrict__ dyn_sum,
               const uint32_t* __restrict__ msm_fail, const uint32_t* __restrict__ wellformed, uint32_t n_msm,
               uint32_t group, const uint32_t* __restrict__ fail_list, const uint32_t* __restrict__ n_fail,
               const uint32_t* __restrict__ fail_sum, uint32_t* __restrict__ row_map, uint32_t* __restrict__ n_recheck,
               uint32_t* __restrict__ cand, const uint32_t* __restrict__ st_scalars, uint32_t n_static,
               int16_t* __restrict__ digits, int w, int W) {
  __shared__ unsigned long long sh_queue;
  const uint32_t f = blockIdx.x;
  if (f >= *n_fail) return;
  {
    ge acc;
    static_row_share(acc, loc_digits, (uint64_t)n_groups * n_static, st_offsets[f], n_static, st_index, table, n_set, H, W, threadIdx.x, blockDim.x);
    store_ext(partials + ((uint64_t)f * blockDim.x + threadIdx.x) * EXT_WORDS, acc);
  }
  __threadfence_block();
  __syncthreads();
  locate_group(partials + (uint64_t)f * blockDim.x * EXT_WORDS, blockDim.x, dyn_sum, msm_fail, wellformed, n_msm, group, fail_list[f], f,
               fail_sum, row_map, n_recheck, cand, st_scalars, n_static, digits, w, W, &sh_queue);
}

// Final verdicts of a batch checked in groups, packed into the bitmap (byte i / 8, bit i % 8):
//   transaction of a group that passed, or left out of its group          accept[] as k_group_combine wrote it
//   ... of a failed group with no single culprit, or the culprit itself    accept[] as k_static_combine wrote it
//   ... of a failed group with a located culprit b, not b                  accepted iff S1 - E_b is the identity
// (E_b = points[cand]: the sum k_static_combine formed for b).  Should that last test ever fail, status[0] bit 2
// is raised and the host re-runs the batch ungrouped.
__global__ void __launch_bounds__(256)
k_pack_bitmap_groups(const uint8_t* __restrict__ accept, const uint32_t* __restrict__ wellformed, const uint32_t* __restrict__ msm_fail,
                     uint8_t* __restrict__ bitmap, uint32_t n_msm, uint32_t group, const uint32_t* __restrict__ grp_state,
                     const uint32_t* __restrict__ fail_sum, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ row_map,
                     const uint32_t* __restrict__ points, uint32_t* __restrict__ status, uint32_t force_unresolved /*test hook*/) {
  const uint32_t byte = blockIdx.x * blockDim.x + threadIdx.x;
  if (byte >= (n_msm + 7) / 8) return;
  uint32_t v = 0, seen_group = 0xffffffffu, b = 0;
  int rest_ok = 0, located = 0;
  for (int k = 0; k < 8; ++k) {
    const uint32_t i = byte * 8 + k;
    if (i >= n_msm) break;
    const uint32_t G = i / group;
    if (G != seen_group) {
      seen_group = G;
      located = 0;
      const uint32_t f1 = grp_state[G];
      if (f1 != 0 && cand[f1 - 1] != LOCATE_NONE) {
        const uint32_t f = f1 - 1, slot = cand[f];
        located = 1;
        b = row_map[slot];
        ge S1, Eb, nb, d;
        load_ext(S1, fail_sum + (uint64_t)f * EXT_WORDS);
        load_ext(Eb, points + (uint64_t)slot * EXT_WORDS);
        ge_neg(nb, Eb);
        ge_add(d, S1, nb);
        rest_ok = (ge_is_identity(d) && !force_unresolved) ? 1 : 0;
        if (!rest_ok) atomicOr(&status[0], 4u);
      }
    }
    int bit;
    if (located && i != b) bit = rest_ok && !tx_excluded(msm_fail, wellformed, i);
    else bit = accept[i] && (!wellformed || wellformed[i]);
    if (bit) v |= 1u << k;
  }
  bitmap[byte] = (uint8_t)v;
}

}  // namespace zk
