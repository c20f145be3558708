// prep_kernels.hpp -- the host half of r1cs::Verifier::verify, on the device
// (SURVEY.md sec 8 row f-2): proof bytes -> the scalars of the verification
// multiscalar multiplication, for a batch of cloak statements of ONE shape.
//
//   k_proof_unpack   byte-aligned proofs -> word-aligned rows, version check
//   k_transcript     one lane per transaction: Merlin replay, challenges,
//                    canonicity / identity checks, the one field inversion
//   k_prepare        one workgroup per transaction: monomials, z powers,
//                    constraint flattening (plan replay), s vector, y^-i,
//                    delta, g_i / h_i and the proof-point scalars
//
// Challenge slots of a transaction (scm, Montgomery form):
//   0 y  1 z  2 u  3 x  4 w  5 P1 = prod u_j  6 U = prod u_j^2  7 r  8 t_x  9 t_x_blinding
//   10 e_blinding  11 a  12 b  13 rho (weight of this transaction in a group check, 1 when checked alone)
//   | 14.. second-phase challenges | u_j | prod_{l != j} u_l^2
#pragma once
#include "keccak_coop.hpp"
#include "merlin_dev.hpp"
#include "sc_dev.hpp"
#include "transcript_tape.hpp"

namespace zk {

constexpr int CH_FIXED = 14;

struct PrepShape {
  uint32_t m, n1, n, pn, k, n_cons, n_chal2, n_mono, n_targets, n_terms;
  uint32_t proof_words;     // (16 + 2k) * 8
  uint32_t n_ch;            // CH_FIXED + n_chal2 + 2k
  uint32_t n_dyn, n_static; // 11 + m + 2k, 2 + 2 pn
  uint32_t n_heavy;         // targets with many terms (e.g. wc) are summed by the whole workgroup
  uint32_t heavy[8];
  // slots per transaction in the challenge buffer: n_ch challenges | n_mono monomials | PREP_STRIDES
  uint32_t n_ch_ext;
  // the flattening runs in n_chunks passes over target ranges [chunk_tgt[i], chunk_tgt[i+1]) so that the
  // per-term products of one pass fit tv_cap LDS slots
  uint32_t n_chunks, tv_cap;
  uint32_t chunk_tgt[6];
  uint32_t chunk_prod[6];   // first product of each pass (products are stored in target order)
};

constexpr uint32_t PREP_STRIDES = 56;   // z^(2^L) [0..15] | y^(2^L) [16..31] | u_j^2 [32..47] | x^2..x^6, r x^2 [48..53] (k_prepare's own)

__device__ __forceinline__ void ld_scm(scm& s, const uint32_t* p) {
#pragma unroll
  for (int i = 0; i < 8; ++i) s.v[i] = p[i];
}
__device__ __forceinline__ void st_scm(uint32_t* p, const scm& s) {
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = s.v[i];
}

__device__ inline scm scm_pow_u32(const scm& base, uint32_t e) {
  scm acc = scm_one();
  if (e == 0) return acc;
  const int top = 31 - __clz(e);
  for (int i = top; i >= 0; --i) {
    acc = scm_sq(acc);
    if ((e >> i) & 1) acc = scm_mul(acc, base);
  }
  return acc;
}

// ---- k_proof_unpack -------------------------------------------------------------------
// compact != 0: the one-phase wire format (version byte 0; A_I2, A_O2, S2 left out, proof_stride = 1 + 4 (proof_words - 24)):
// the identity -- 32 zero bytes -- stands in for the three points, as upstream's R1CSProof::from_bytes has it
__global__ void __launch_bounds__(256)
k_proof_unpack(const uint8_t* __restrict__ proofs, uint64_t proof_stride, uint32_t* __restrict__ pw,
               uint32_t proof_words, uint32_t batch, uint32_t* __restrict__ wellformed, uint32_t compact) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)batch * proof_words) return;
  const uint32_t tx = (uint32_t)(g / proof_words), j = (uint32_t)(g % proof_words);
  const uint8_t* p = proofs + (uint64_t)tx * proof_stride;
  uint32_t v = 0;
  if (!compact || j < 24 || j >= 48) {
    const uint8_t* b = p + 1 + 4 * (uint64_t)((compact && j >= 48) ? j - 24 : j);
    v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
  }
  pw[g] = v;
  if (j == 0 && p[0] != (compact ? 0 : 1)) atomicAnd(&wellformed[tx], 0u);   // version byte and length must agree
}

// ---- k_transcript -----------------------------------------------------------------------
__device__ __forceinline__ bool words_are_zero(const uint32_t* w) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc |= w[i];
  return acc == 0;
}

// One lane per transaction runs the shape's tape (transcript_tape.hpp).  LDS per lane, word-
// interleaved over the 64 lanes of the block: the 50 words of STROBE state.
// init: the STROBE state after Transcript::new("ZkVM.r1cs") + r1cs_domain_sep (same for every tx).
//
// No inversion anywhere: the verification equation is used multiplied through by
// c = y^(pn-1) prod_j u_j^2, which turns every y^-i, 1/u_j and u_j^-2 into a product of positive
// powers (k_prepare); "sum == identity" is unchanged by a non-zero factor, and a zero y or u_j
// (where the reference's inversion has no answer either) rejects the proof.
__global__ void __launch_bounds__(64)
k_transcript(PrepShape sh, const uint32_t* __restrict__ init_state /*50 words*/, const uint4* __restrict__ tape,
             uint32_t n_ops, const uint32_t* __restrict__ com /*[B][m][8]*/,
             const uint32_t* __restrict__ pw /*[B][proof_words]*/, const uint32_t* __restrict__ rbytes /*[B][16]*/,
             uint32_t batch, uint32_t* __restrict__ ch /*[B][n_ch_ext][8]*/, uint32_t* __restrict__ wellformed,
             const uint32_t* __restrict__ mono_chal, const uint32_t* __restrict__ mono_pow, uint32_t grouped) {
  __shared__ uint32_t lds[50 * 64];
  const uint32_t lane = threadIdx.x;
  const uint32_t tx_raw = blockIdx.x * 64 + lane;
  const bool live = tx_raw < batch;
  const uint32_t tx = live ? tx_raw : batch - 1;
  uint32_t* st = lds + lane;                    // word w of this lane: st[w * 64]
  for (int i = 0; i < 50; ++i) st[i * 64] = init_state[i];
  const uint32_t* c = com + (uint64_t)tx * sh.m * 8;
  const uint32_t* p = pw + (uint64_t)tx * sh.proof_words;
  uint32_t* out = ch + (uint64_t)tx * sh.n_ch_ext * 8;
#pragma unroll 1
  for (uint32_t i = 0; i < n_ops; ++i) {
    const uint4 op = tape[i];
    if (op.x == TAPE_XOR) {
      st[op.y * 64] ^= op.z;
    } else if (op.x == TAPE_DATA) {
      const uint32_t* src = op.y == TAPE_SRC_PROOF ? p : c;
      const int last_word = (int)(op.y == TAPE_SRC_PROOF ? sh.proof_words : sh.m * 8) - 1;
      const uint32_t spos = op.w & 0xffffu, n = op.w >> 16;
      const int delta = (int)op.z - (int)spos;            // source byte = state byte + delta
#pragma unroll 1
      for (uint32_t w = spos >> 2; w <= (spos + n - 1) >> 2; ++w) {
        const int sb = (int)(4 * w) + delta;              // source byte under state byte 4w
        const int wi = sb >> 2;
        const uint32_t sh8 = (uint32_t)(sb & 3) * 8;
        const uint32_t lo = src[min(max(wi, 0), last_word)], hi = src[min(max(wi + 1, 0), last_word)];
        const uint32_t v = sh8 ? (lo >> sh8) | (hi << (32 - sh8)) : lo;
        const uint32_t first = max(spos, 4 * w) - 4 * w, last = min(spos + n, 4 * w + 4) - 4 * w;
        const uint32_t mask = (last == 4 ? 0xffffffffu : ((1u << (8 * last)) - 1)) & ~((1u << (8 * first)) - 1);
        st[w * 64] ^= v & mask;
      }
    } else if (op.x == TAPE_PERM) {
      uint32_t klo[25], khi[25];
#pragma unroll
      for (int q = 0; q < 25; ++q) { klo[q] = st[(2 * q) * 64]; khi[q] = st[(2 * q + 1) * 64]; }
      keccak_f1600_halves(klo, khi);
#pragma unroll
      for (int q = 0; q < 25; ++q) { st[(2 * q) * 64] = klo[q]; st[(2 * q + 1) * 64] = khi[q]; }
    } else {                                               // TAPE_CHAL
      uint32_t wv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) { wv[q] = st[q * 64]; st[q * 64] = 0; }
      st_scm(out + op.y * 8, scm_from_wide(wv));
    }
  }
  // well-formedness: no identity among the proof points, canonical scalars
  bool ok = true;
  // (A_I2, A_O2, S2 are the identity in single-phase proofs: not tested, as in the reference)
  for (int i = 0; i < 11; ++i) ok &= (i >= 3 && i < 6) | !words_are_zero(p + 8 * i);
  const uint32_t* sc3 = p + 88;               // t_x, t_x_blinding, e_blinding
  const uint32_t* lr = p + 112;               // L_0 R_0 L_1 R_1 ...
  const uint32_t* ab = lr + 16 * sh.k;        // a, b
  ok &= scm_is_canonical(sc3) & scm_is_canonical(sc3 + 8) & scm_is_canonical(sc3 + 16) & scm_is_canonical(ab) &
        scm_is_canonical(ab + 8);
  for (uint32_t j = 0; j < sh.k; ++j) ok &= !words_are_zero(lr + 16 * j) & !words_are_zero(lr + 16 * j + 8);
#pragma unroll 1
  for (int q = 0; q < 5; ++q) st_scm(out + (8 + q) * 8, scm_from_words(q < 3 ? sc3 + 8 * q : ab + 8 * (q - 3)));
  {
    // r combines the two halves of this proof's equation; rho = r^2 weighs the whole equation inside a
    // group of transactions checked by one multiscalar multiplication (coefficients r^2, r^3 of a
    // transaction's two halves: a polynomial identity in independent r's, Schwartz-Zippel as for r alone)
    const scm rr = scm_from_wide(rbytes + (uint64_t)tx * 16);
    st_scm(out + 7 * 8, rr);
    scm rho = grouped ? scm_sq(rr) : scm_one();
    uint32_t any = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) any |= rho.v[q];
    if (any == 0) rho = scm_one();
    st_scm(out + 13 * 8, rho);
  }
  // the serial chains k_prepare needs, done here where every lane has one to do: second-phase
  // monomials, z^(2^L), y^(2^L), u_j^2, U = prod u_j^2, prod_{l != j} u_l^2.  (Measured the other way
  // round -- raw challenge bytes out of this kernel, reductions and chains on parallel lanes of
  // k_prepare: transcript 0.57 -> 0.38 ms, prepare 0.25 -> 0.27 ms, and the step 3-5 % SLOWER: the
  // chip-filling kernel's extra work costs more than the light kernel's latency.)
  uint32_t* sym = out + sh.n_ch * 8;
  uint32_t* strides = sym + sh.n_mono * 8;
  uint32_t* uj = out + (CH_FIXED + sh.n_chal2) * 8;
  uint32_t* uex = uj + 8 * sh.k;
#pragma unroll 1
  for (uint32_t j = 0; j < sh.n_mono; ++j) {
    scm v = scm_one();
    if (mono_chal[j] != 0xffffffffu) {
      scm cc;
      ld_scm(cc, out + (CH_FIXED + mono_chal[j]) * 8);
      v = scm_pow_u32(cc, mono_pow[j]);
    }
    st_scm(sym + 8 * j, v);
  }
  {
    scm cc;
    ld_scm(cc, out + 1 * 8);
    st_scm(strides, cc);
#pragma unroll 1
    for (uint32_t L = 1; (1u << L) < sh.n_cons; ++L) { cc = scm_sq(cc); st_scm(strides + 8 * L, cc); }
    ld_scm(cc, out + 0 * 8);                      // y
    ok &= !words_are_zero(cc.v);
    st_scm(strides + 16 * 8, cc);
#pragma unroll 1
    for (uint32_t L = 1; L < sh.k; ++L) { cc = scm_sq(cc); st_scm(strides + (16 + L) * 8, cc); }
    // prefix products of the u_j^2 go to the "excluded" slots, then the suffix pass completes them
    scm run = scm_one(), p1 = scm_one();
#pragma unroll 1
    for (uint32_t j = 0; j < sh.k; ++j) {
      ld_scm(cc, uj + 8 * j);
      ok &= !words_are_zero(cc.v);
      p1 = scm_mul(p1, cc);
      const scm sq = scm_sq(cc);
      st_scm(strides + (32 + j) * 8, sq);
      st_scm(uex + 8 * j, run);                   // prod_{l < j} u_l^2
      run = scm_mul(run, sq);
    }
    st_scm(out + 6 * 8, run);                     // U = prod u_j^2
    st_scm(out + 5 * 8, p1);                      // P1 = prod u_j
    run = scm_one();
#pragma unroll 1
    for (uint32_t j = sh.k; j-- > 0;) {
      scm pre, sq;
      ld_scm(pre, uex + 8 * j);
      st_scm(uex + 8 * j, scm_mul(pre, run));     // prod_{l != j} u_l^2
      ld_scm(sq, strides + (32 + j) * 8);
      run = scm_mul(run, sq);
    }
  }
  if (live && !ok) atomicAnd(&wellformed[tx], 0u);
}

// ---- the cooperative transcript: k_tape_gather + k_transcript_coop + k_challenges ------------------
// The same tape with ONE Keccak state per WAVEFRONT (keccak_coop.hpp) instead of one per lane: a
// 1024-transaction batch is 1024 wavefronts walking ~36 permutations of ~36 instructions per round,
// instead of 16 wavefronts walking ~36 permutations of ~180.  Three launches:
//   k_tape_gather      fully parallel: per (transaction, segment, state word) the 8 bytes that segment
//                      XORs into that word (constants | proof / commitment bytes), as one u64
//   k_transcript_coop  one wavefront per transaction: [challenge bytes out] ^= absorb, permute; no
//                      scalar arithmetic, no byte fiddling, one 8-byte load per lane and segment
//   k_challenges       two wavefronts per transaction: 64-byte challenges -> scalars mod l (one lane
//                      each), then the serial power chains k_prepare needs, one chain per lane
// Together they write exactly what k_transcript writes (tests compare the bytes with the oracle's).
struct DevKcTraits {
  using V = uint32_t;
  static __device__ __forceinline__ V splat(uint32_t x) { return x; }
  static __device__ __forceinline__ V xor2(V a, V b) { return a ^ b; }
  static __device__ __forceinline__ V xor3(V a, V b, V c) { return a ^ b ^ c; }
  static __device__ __forceinline__ V chi(V a, V b, V c) { return a ^ (~b & c); }
  static __device__ __forceinline__ V xor_and(V a, V b, V m) { return a ^ (b & m); }
  static __device__ __forceinline__ V and_(V a, V m) { return a & m; }
  static __device__ __forceinline__ V sel(V m, V a, V b) { return (a & m) | (b & ~m); }
  static __device__ __forceinline__ V alignbit(V hi, V lo, V s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
  static __device__ __forceinline__ V ror8(V v) { return (V)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false); }
  static __device__ __forceinline__ V shr1(V v) { return (V)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false); }
  static __device__ __forceinline__ V shl1(V v) { return (V)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x101, 0xf, 0xf, false); }
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ void swap16(V& a, V& b) { const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r.x; b = r.y; }
  static __device__ __forceinline__ void swap32(V& a, V& b) { const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r.x; b = r.y; }
  static __device__ __forceinline__ V gather(V addr, V v) { return (V)__builtin_amdgcn_ds_bpermute((int)addr, (int)v); }
};

// Test hook: the cross-lane primitives exactly as the cooperative Keccak uses them, so that a GPU test can
// compare the hardware's semantics with the host emulation (coop::HostTraits) primitive by primitive, and
// one Keccak-f per wavefront on caller-supplied states.  out: [8][64] words | states permuted in place.
__global__ void __launch_bounds__(64)
k_coop_selftest(const uint32_t* __restrict__ in /*[3][64]: a, b, addr*/, uint32_t* __restrict__ out /*[8][64]*/,
                uint2* __restrict__ states /*[n][25]*/, uint32_t n_states) {
  const uint32_t lane = threadIdx.x;
  if (blockIdx.x == 0) {
    const uint32_t a = in[lane], b = in[64 + lane], addr = in[128 + lane];
    out[lane] = DevKcTraits::ror8(a);
    out[64 + lane] = DevKcTraits::shr1(a);
    out[128 + lane] = DevKcTraits::shl1(a);
    uint32_t x = a, y = b;
    DevKcTraits::swap16(x, y);
    out[192 + lane] = x; out[256 + lane] = y;
    x = a; y = b;
    DevKcTraits::swap32(x, y);
    out[320 + lane] = x; out[384 + lane] = y;
    out[448 + lane] = DevKcTraits::gather(addr, a);
  }
  const coop::KcLane k = coop::kc_lane(lane);
  coop::KeccakCoop<DevKcTraits>::Consts c = {k.live, k.rot_swap, k.rot_t, k.src[0], k.src[1], k.src[2], k.iota};
  for (uint32_t i = blockIdx.x; i < n_states; i += gridDim.x) {
    uint2 w = k.live ? states[(uint64_t)i * 25 + k.q] : make_uint2(0, 0);
    coop::KeccakCoop<DevKcTraits>::permute(w.x, w.y, c);
    if (k.primary) states[(uint64_t)i * 25 + k.q] = w;
  }
}

__global__ void __launch_bounds__(256)
k_tape_gather(PrepShape sh, uint32_t n_seg, const uint32_t* __restrict__ seg_const /*[n_seg][50]*/,
              const uint16_t* __restrict__ seg_map /*[n_seg][200]*/, const uint32_t* __restrict__ com,
              const uint32_t* __restrict__ pw, uint32_t batch, uint2* __restrict__ absorb /*[B][n_seg][25]*/) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t per_tx = n_seg * 25;
  if (g >= (uint64_t)batch * per_tx) return;
  const uint32_t tx = (uint32_t)(g / per_tx), rem = (uint32_t)(g % per_tx), seg = rem / 25, q = rem % 25;
  const uint4 mp4 = *reinterpret_cast<const uint4*>(seg_map + (uint64_t)seg * 200 + 8 * q);
  const uint32_t mp[4] = {mp4.x, mp4.y, mp4.z, mp4.w};
  const uint32_t* c = com + (uint64_t)tx * sh.m * 8;
  const uint32_t* p = pw + (uint64_t)tx * sh.proof_words;
  const uint32_t n_com_bytes = 32 * sh.m;
  uint32_t lo = seg_const[seg * 50 + 2 * q], hi = seg_const[seg * 50 + 2 * q + 1];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const uint32_t idx = (mp[b >> 1] >> (16 * (b & 1))) & 0xffffu;
    if (idx) {
      const uint32_t i = idx - 1;
      const uint32_t word = i < n_com_bytes ? c[i >> 2] : p[(i - n_com_bytes) >> 2];
      const uint32_t byte = (word >> (8 * (i & 3))) & 0xffu;
      if (b < 4) lo ^= byte << (8 * b); else hi ^= byte << (8 * (b - 4));
    }
  }
  absorb[g] = make_uint2(lo, hi);
}

__global__ void __launch_bounds__(64)
k_transcript_coop(uint32_t n_seg, const uint32_t* __restrict__ seg_info, const uint32_t* __restrict__ init_state /*50 words*/,
                  const uint2* __restrict__ absorb /*[B][n_seg][25]*/, uint32_t batch, uint32_t n_ch,
                  uint32_t* __restrict__ raw /*[B][n_ch][16]*/) {
  const uint32_t tx = blockIdx.x, lane = threadIdx.x;
  if (tx >= batch) return;
  const coop::KcLane k = coop::kc_lane(lane);
  const coop::KeccakCoop<DevKcTraits>::Consts c = {k.live, k.rot_swap, k.rot_t, k.src[0], k.src[1], k.src[2], k.iota};
  uint32_t lo = k.live ? init_state[2 * k.q] : 0, hi = k.live ? init_state[2 * k.q + 1] : 0;
  const uint2* ab = absorb + (uint64_t)tx * n_seg * 25 + k.q;
  const bool first8 = k.live && k.q < 8;           // state bytes 0..63: what a challenge squeezes
  uint2 nxt = k.live ? ab[0] : make_uint2(0, 0);
#pragma unroll 1
  for (uint32_t seg = 0; seg < n_seg; ++seg) {
    const uint32_t info = seg_info[seg];
    const uint2 cur = nxt;
    if (seg + 1 < n_seg && k.live) nxt = ab[(uint64_t)(seg + 1) * 25];      // in flight during the permutation
    const uint32_t slot = info & 0xffffu;
    if (slot) {
      if (k.primary && k.q < 8) {
        uint32_t* o = raw + ((uint64_t)tx * n_ch + (slot - 1)) * 16 + 2 * k.q;
        o[0] = lo; o[1] = hi;
      }
      if (first8) { lo = 0; hi = 0; }
    }
    lo ^= cur.x; hi ^= cur.y;
    if (info >> 31) coop::KeccakCoop<DevKcTraits>::permute(lo, hi, c);
  }
}

// blockDim = 128: wavefront 0 reduces the challenges and runs the power chains, wavefront 1 the
// well-formedness checks and the products of the inner-product challenges.
__global__ void __launch_bounds__(128)
k_challenges(PrepShape sh, const uint32_t* __restrict__ raw /*[B][n_ch][16]*/, const uint32_t* __restrict__ pw,
             const uint32_t* __restrict__ rbytes, uint32_t batch, uint32_t* __restrict__ ch /*[B][n_ch_ext][8]*/,
             uint32_t* __restrict__ wellformed, const uint32_t* __restrict__ mono_chal, const uint32_t* __restrict__ mono_pow,
             uint32_t grouped) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];     // n_ch slots of 8 words
  const uint32_t tx = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  if (tx >= batch) return;
  const uint32_t* p = pw + (uint64_t)tx * sh.proof_words;
  uint32_t* out = ch + (uint64_t)tx * sh.n_ch_ext * 8;
  const uint32_t n2 = sh.n_chal2, k = sh.k;
  // phase 1: every slot is the reduction of 64 little-endian bytes (8-word sources padded with zeros)
  for (uint32_t s = t; s < sh.n_ch; s += nt) {
    uint32_t w[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) w[q] = 0;
    const bool is_chal = s < 5 || (s >= (uint32_t)CH_FIXED && s < CH_FIXED + n2 + k);
    const uint32_t* src = nullptr;
    int words = 0;
    if (is_chal) { src = raw + ((uint64_t)tx * sh.n_ch + s) * 16; words = 16; }
    else if (s == 7 || s == 13) { src = rbytes + (uint64_t)tx * 16; words = 16; }
    else if (s >= 8 && s <= 12) { src = s <= 10 ? p + 88 + 8 * (s - 8) : p + 112 + 16 * k + 8 * (s - 11); words = 8; }
#pragma unroll
    for (int q = 0; q < 16; ++q) if (q < words) w[q] = src[q];
    scm v = scm_from_wide(w);
    if (s == 13) {               // rho = r^2 inside a group check (1 when the transaction is checked alone, or r = 0)
      v = grouped ? scm_sq(v) : scm_one();
      uint32_t any = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) any |= v.v[q];
      if (any == 0) v = scm_one();
    }
    if (words) { st_scm(lds + 8 * s, v); st_scm(out + 8 * s, v); }
  }
  __syncthreads();
  uint32_t* sym = out + sh.n_ch * 8;
  uint32_t* strides = sym + sh.n_mono * 8;
  if (t < 64) {
    // chains, one per lane: 0 the squarings of z, 1 those of y, 2 + j the monomial j (challenge^power).
    // Uniform loop: acc = acc^2 [* base]; a chain stores what it needs as it goes.
    const uint32_t n_chains = 2 + sh.n_mono;
    uint32_t n_zs = 1;
    while ((1u << n_zs) < sh.n_cons) ++n_zs;           // strides z^(2^L), L < n_zs
    for (uint32_t base0 = 0; base0 < n_chains; base0 += 64) {
      const uint32_t cid = base0 + t;
      scm base = scm_one(), acc = scm_one();
      uint32_t steps = 0, e = 0;
      if (cid == 0) { ld_scm(base, lds + 8 * 1); acc = base; steps = n_zs - 1; st_scm(strides, acc); }
      else if (cid == 1) { ld_scm(base, lds + 8 * 0); acc = base; steps = k ? k - 1 : 0; st_scm(strides + 16 * 8, acc); }
      else if (cid < n_chains) {
        const uint32_t j = cid - 2, mc = mono_chal[j];
        e = mono_pow[j];
        if (mc != 0xffffffffu && e != 0) {
          ld_scm(base, lds + 8 * (CH_FIXED + mc));
          acc = base;
          steps = 31 - __clz(e);                        // bits below the top one
        } else {
          e = 0;
        }
      }
      uint32_t max_steps = steps;
#pragma unroll 1
      for (int d = 32; d >= 1; d >>= 1) max_steps = max(max_steps, (uint32_t)__shfl_xor((int)max_steps, d));
#pragma unroll 1
      for (uint32_t i = 1; i <= max_steps; ++i) {
        const bool on = i <= steps;
        const scm sq = scm_sq(acc);
        if (on) acc = sq;
        const bool mul = on && cid >= 2 && ((e >> (steps - i)) & 1);
        if (__any(mul)) { const scm m = scm_mul(acc, base); if (mul) acc = m; }
        if (on && cid == 0) st_scm(strides + 8 * i, acc);
        if (on && cid == 1) st_scm(strides + (16 + i) * 8, acc);
      }
      if (cid >= 2 && cid < n_chains) st_scm(sym + 8 * (cid - 2), acc);
    }
    return;
  }
  // wavefront 1
  const uint32_t lane = t - 64;
  bool ok = true;
  if (lane == 0) {
    // well-formedness: no identity among the proof points (A_I2, A_O2, S2 are the identity in single-phase
    // proofs: not tested, as in the reference), canonical scalars
    for (int i = 0; i < 11; ++i) ok &= (i >= 3 && i < 6) | !words_are_zero(p + 8 * i);
    const uint32_t* sc3 = p + 88;
    const uint32_t* lr = p + 112;
    const uint32_t* ab = lr + 16 * k;
    ok &= scm_is_canonical(sc3) & scm_is_canonical(sc3 + 8) & scm_is_canonical(sc3 + 16) & scm_is_canonical(ab) &
          scm_is_canonical(ab + 8);
    for (uint32_t j = 0; j < k; ++j) ok &= !words_are_zero(lr + 16 * j) & !words_are_zero(lr + 16 * j + 8);
    ok &= !words_are_zero(lds + 0);                    // y = 0: the reference's inversion has no answer either
  }
  // products of the inner-product challenges u_j (lane j < k <= 16): prefix and suffix products by
  // doubling steps, then P1 = prod u_j, U = P1^2, u_j^2 and prod_{l != j} u_l^2 = (prefix_{j-1} suffix_{j+1})^2
  uint32_t* uj = lds + (CH_FIXED + n2) * 8;
  uint32_t* uex = out + (CH_FIXED + n2 + k) * 8;
  scm u = scm_one();
  if (lane < k) { ld_scm(u, uj + 8 * lane); ok &= !words_are_zero(u.v); }
  scm pre = u, suf = u;
#pragma unroll 1
  for (uint32_t d = 1; d < k; d <<= 1) {
    scm a, b;
#pragma unroll
    for (int q = 0; q < 8; ++q) { a.v[q] = (uint32_t)__shfl_up((int)pre.v[q], d); b.v[q] = (uint32_t)__shfl_down((int)suf.v[q], d); }
    const scm pa = scm_mul(pre, a), sb = scm_mul(suf, b);
    if (lane >= d && lane < k) pre = pa;
    if (lane + d < k) suf = sb;
  }
  scm pm, sp;                                           // prefix_{j-1}, suffix_{j+1}
#pragma unroll
  for (int q = 0; q < 8; ++q) { pm.v[q] = (uint32_t)__shfl_up((int)pre.v[q], 1); sp.v[q] = (uint32_t)__shfl_down((int)suf.v[q], 1); }
  if (lane == 0) pm = scm_one();
  if (lane + 1 >= k) sp = scm_one();
  const scm ex = scm_mul(pm, sp), ex2 = scm_sq(ex), u2 = scm_sq(u);
  if (lane < k) { st_scm(strides + (32 + lane) * 8, u2); st_scm(uex + 8 * lane, ex2); }
  if (k == 0 ? lane == 0 : lane == k - 1) {
    const scm p1 = k ? pre : scm_one();
    st_scm(out + 5 * 8, p1);
    st_scm(out + 6 * 8, scm_sq(p1));
  }
  if (!__all(ok) && lane == 0) atomicAnd(&wellformed[tx], 0u);
}

// ---- k_prepare ------------------------------------------------------------------------------
// Scalars live in the lazy limb form of sc_dev.hpp (scl: ten 26-bit limbs, 40-byte slots) from the moment the
// challenge slots are copied into LDS until the canonical words are written out: a product is the multiply-adds and
// one carry pass, sums and differences are limb-wise (bounds in the comments where they matter).
// LDS (10-word slots): chs[n_ch] | sym[n_mono] | strides[PREP_STRIDES] (these three converted from k_transcript's
// 8-word Montgomery slots) | wv[n_targets] | ylo[16] yhi[pn/16] slo[16] shi[pn/16] red[8] shr[16] | region A: zpow[n_cons]
// tv[tv_cap + 32] (zlo[16] zhi[n_cons/16] at tv's place until zpow is made), and once the flattening is done yip[pn] sv[pn]
// in its place.
//
// Values that every lane of the workgroup needs (x U, a P1 rho Y, b P1, c', ...) are computed ONCE, each by one lane, beside
// work that leaves lanes idle anyway, and handed over through shr[]: written as plain expressions they are products of
// wavefront-uniform values, which the compiler moves to the scalar unit -- four scalar instructions per multiply-add in
// one dependent chain, repeated by every wavefront (measured, round 3: 5 000 scalar instructions per wavefront beside
// 7 000 vector ones, SQ_INSTS_SALU 2.0e8 per launch).  lane_zero() keeps single-lane sections on the vector unit.
//
// Power tables are OUTER PRODUCTS of two small tables, entry (16 h + l) = hi[h] * lo[l]: the small tables -- six of them, of
// z^(q+1), rho y^i and s_i -- grow by doubling (entry q + 2^L = entry q * stride_L, one product per entry) all at once on
// the lanes of ONE wavefront, four or five dependent products deep, and every big table is then one product per entry on
// all lanes.  (Until round 5 the big tables themselves grew by doubling, one after the other: 17 dependent products with
// one to 128 lanes busy -- 57 % of a workgroup's life by the per-section clocks of tools/prep_stamps.py.)
// y^-i is kept in PLAIN form: a Montgomery
// product with one plain operand yields a plain result, so the generator scalars come out as
// plain values without a conversion product of their own.
//
// Flattening (plan replay): a term of target g is either a UNIT term, +- z^(q+1) read straight from the power table
// (three quarters of a cloak's terms), or a product const * monomial * z^(q+1) computed in a pass of its own into tv.
constexpr uint32_t SCL_WORDS = 10;
constexpr uint32_t TERM_UNIT = 0x80000000u, TERM_NEG = 0x40000000u, TERM_IDX = 0x00ffffffu;
constexpr uint32_t HEAVY_TERMS = 16;      // lazy sums are reduced every so many terms; targets with more are summed by the whole workgroup

// the small tables every power table is the outer product of (see k_prepare): entries of z^(16 h), of the high factors of
// rho y^i and of s_i; the low factors have 16 entries (2^k when k < 4)
__host__ __device__ inline uint32_t prep_zhi(const PrepShape& sh) { return ((sh.n_cons ? sh.n_cons - 1 : 0) >> 4) + 1; }
__host__ __device__ inline uint32_t prep_lo_bits(const PrepShape& sh) { return sh.k < 4 ? sh.k : 4; }
// bytes: 40-byte slots, except the two tables of the second life (yip, sv: products < 2^255 packed into 32 bytes)
__host__ __device__ inline size_t prepare_lds_bytes(const PrepShape& sh) {
  const size_t scratch = sh.tv_cap + 32 > 16 + prep_zhi(sh) ? sh.tv_cap + 32 : 16 + prep_zhi(sh);
  const size_t first = ((size_t)sh.n_cons + scratch) * 40, second = (size_t)2 * sh.pn * 32;
  const size_t small = 2 * 16 + 2 * ((size_t)sh.pn >> prep_lo_bits(sh)) + 24;      // ylo yhi slo shi | red[8] shr[16]
  return ((size_t)sh.n_ch_ext + sh.n_targets + small) * 40 + 16 + (first > second ? first : second);
}

// a zero in a vector register that the compiler cannot see through: added to a shared address it makes the loaded value
// lane-specific in the compiler's eyes, so that arithmetic on it stays on the vector unit
__device__ __forceinline__ uint32_t lane_zero() {
  uint32_t z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}
__device__ __forceinline__ void ld_scl(scl& s, const uint32_t* p) {
  const uint2* q = reinterpret_cast<const uint2*>(p);
#pragma unroll
  for (int i = 0; i < 5; ++i) { const uint2 w = q[i]; s.v[2 * i] = w.x; s.v[2 * i + 1] = w.y; }
}
// p is the same address in every lane: the value lives in scalar registers afterwards
__device__ __forceinline__ void ld_scl_shared(scl& s, const uint32_t* p) {
  ld_scl(s, p);
#pragma unroll
  for (int i = 0; i < 10; ++i) s.v[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.v[i]);
}
__device__ __forceinline__ void st_scl(uint32_t* p, const scl& s) {
  uint2* q = reinterpret_cast<uint2*>(p);
#pragma unroll
  for (int i = 0; i < 5; ++i) q[i] = make_uint2(s.v[2 * i], s.v[2 * i + 1]);
}
__device__ __forceinline__ void ld_scl8(scl& s, const uint32_t* p) {    // packed table entry
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  s = scl_from_words(w);
}
__device__ __forceinline__ void st_scl8(uint32_t* p, const scl& s) {    // s tight, < 2^256
  uint32_t w[8];
  scl_pack8(w, s);
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ scl shfl_down_scl(const scl& a, int delta) {
  scl o;
#pragma unroll
  for (int q = 0; q < 10; ++q) o.v[q] = (uint32_t)__shfl_down((int)a.v[q], delta);
  return o;
}
// sum over the wavefront, in lane 63: inputs tight (limbs < 2^26, so that 64 of them fit a word) and < 2^256 -> tight, < 2 l.
// Six DPP additions per limb (row_shr 1 2 4 8, then row_bcast 15 and 31) and ONE carry pass at the end.
__device__ __forceinline__ scl wave_sum_scl(scl part) {
#pragma unroll
  for (int q = 0; q < 10; ++q) {
    uint32_t v = part.v[q];
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // lane 15 of rows 0, 2 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // lane 31 -> rows 2, 3
    part.v[q] = v;
  }
  return scl_weak(part);
}

// Per-section clocks (build variant -DZK_PREP_STAMPS only: tools/prep_stamps.py): thread 0 and thread 255 of the workgroups
// 4096 .. 4351 (the middle of a 10 240-transaction launch) note s_memtime where a section ends; read back through zkgpu_debug_read("prep_stamps").
#ifdef ZK_PREP_STAMPS
constexpr int PREP_STAMP_SLOTS = 16;
__device__ unsigned long long g_prep_stamps[256 * 2 * PREP_STAMP_SLOTS];
#define PREP_STAMP(i) do { if ((blockIdx.x >> 8) == 16 && (t == 0 || t == 255)) \
  g_prep_stamps[((blockIdx.x & 255) * 2 + (t ? 1 : 0)) * PREP_STAMP_SLOTS + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PREP_STAMP(i) do { } while (0)
#endif

__global__ void __launch_bounds__(256, 4)
k_prepare(PrepShape sh, const uint32_t* __restrict__ mono_chal, const uint32_t* __restrict__ mono_pow,
          const uint32_t* __restrict__ tgt_off, const uint32_t* __restrict__ term_info,
          const uint2* __restrict__ prod_qm, const uint32_t* __restrict__ prod_coef,
          const uint32_t* __restrict__ ch, const uint32_t* __restrict__ com, const uint32_t* __restrict__ pw,
          uint32_t* __restrict__ dyn_scalars, uint32_t* __restrict__ dyn_recoded, uint32_t* __restrict__ static_scalars) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  constexpr uint32_t SW = SCL_WORDS;
  uint32_t* chs = lds;
  uint32_t* sym = chs + sh.n_ch * SW;
  uint32_t* zs = sym + sh.n_mono * SW;          // z^(2^L)
  uint32_t* ys = zs + 16 * SW;                   // y^-(2^L)
  uint32_t* us2 = ys + 16 * SW;                  // u_j^2
  uint32_t* wv = chs + sh.n_ch_ext * SW;
  const uint32_t LB = prep_lo_bits(sh), NLO = 1u << LB, PH = sh.pn >> LB, ZLO = sh.n_cons < 16 ? sh.n_cons : 16, ZH = prep_zhi(sh);
  uint32_t* ylo = wv + sh.n_targets * SW;       // plain y^l, l < 16            (rho y^i = ylo[i & 15] * yhi[i >> 4], plain)
  uint32_t* yhi = ylo + 16 * SW;                 // rho y^(16 h)
  uint32_t* slo = yhi + PH * SW;                 // s_i = slo[i & 15] * shi[i >> 4]
  uint32_t* shi = slo + 16 * SW;
  uint32_t* red = shi + PH * SW;
  uint32_t* shr = red + 8 * SW;                  // 0: x U  1: a P1  2: b P1  3: c' (plain)  4: a P1 rho Y (plain)  5: c' (Montgomery)  6: rho Y (plain)  7: plain 1
                                                 // 8 .. 12: (x U) u, (a P1 rho Y) u, (b P1) u, U u, c' u -- what slots 0, 4, 2, U, 3 are for i >= n1
  uint32_t* zpow = lds + (((uint32_t)(shr + 16 * SW - lds) + 3u) & ~3u);      // region A (16-byte aligned), first life
  uint32_t* tv = zpow + sh.n_cons * SW;
  uint32_t* zlo = tv;                            // z^(l+1), l < 16              (z^(q+1) = zlo[q & 15] * zhi[q >> 4])
  uint32_t* zhi = tv + 16 * SW;                  // z^(16 h)
  uint32_t* yip = zpow;                          // region A, second life: two packed tables (8 words per entry)
  uint32_t* sv = yip + sh.pn * 8;
  const uint32_t tx = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  PREP_STAMP(0);

  // the transaction's slots (canonical Montgomery words) -> limb form
  for (uint32_t i = t; i < sh.n_ch_ext; i += nt) {
    const uint4* src = reinterpret_cast<const uint4*>(ch + ((uint64_t)tx * sh.n_ch_ext + i) * 8);
    const uint4 a = src[0], b = src[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    st_scl(chs + i * SW, scl_from_words(w));
  }
  __syncthreads();
  uint32_t* xp = us2 + 16 * SW;                  // xp[0..4] = x^2..x^6, xp[5] = r x^2
  if (t == 0) {                                  // what the small tables start from
    scl z, rho;
    ld_scl(z, zs); ld_scl(rho, chs + 13 * SW);
    st_scl(zlo, z); st_scl(zhi, scl_one());
    st_scl(ylo, scl_plain_one()); st_scl(yhi, rho);
    st_scl(slo, scl_one()); st_scl(shi, scl_one());
    st_scl(shr + 7 * SW, scl_plain_one());
  }
  __syncthreads();
  PREP_STAMP(1);                                  // slots converted
  // phase B: the six small tables by doubling, all in the same steps, one job = one product dst = a * b per lane.  Beside
  // them, on the lanes that follow: the powers of x the proof-point scalars need (step 0: x^2 | step 1: x^3, x^4, r x^2 |
  // step 2: x^5, x^6) and, in step 0, x U, a P1, b P1.  A payment's steps have 10, 15, 27, 48, 10 jobs: one wavefront.
#pragma unroll 1
  for (uint32_t L = 0, half = 1;; ++L, half <<= 1) {
    auto fresh = [half](uint32_t n) { return n > half ? (n - half < half ? n - half : half) : 0u; };   // entries [half, 2 half) of a table of n
    const uint32_t c0 = fresh(ZLO), c1 = fresh(ZH), c2 = fresh(NLO), c3 = fresh(PH);
    const uint32_t n_tab = c0 + c1 + 2 * (c2 + c3), n_side = L == 0 ? 4u : L == 1 ? 3u : L == 2 ? 2u : 0u;
    if (n_tab + n_side == 0) break;
    for (uint32_t j0 = t; j0 < n_tab + n_side; j0 += nt) {
      uint32_t j = j0;
      const uint32_t* pa;
      const uint32_t* pb;
      uint32_t* pd;
      if (j < c0) { pa = zlo + j * SW; pb = zs + L * SW; pd = zlo + (half + j) * SW; }
      else if ((j -= c0) < c1) { pa = zhi + j * SW; pb = zs + (L + 4) * SW; pd = zhi + (half + j) * SW; }
      else if ((j -= c1) < c2) { pa = ylo + j * SW; pb = ys + L * SW; pd = ylo + (half + j) * SW; }
      else if ((j -= c2) < c3) { pa = yhi + j * SW; pb = ys + (L + LB) * SW; pd = yhi + (half + j) * SW; }
      else if ((j -= c3) < c2) { pa = slo + j * SW; pb = us2 + (sh.k - 1 - L) * SW; pd = slo + (half + j) * SW; }
      else if ((j -= c2) < c3) { pa = shi + j * SW; pb = us2 + (sh.k - 1 - L - LB) * SW; pd = shi + (half + j) * SW; }
      else {
        j -= c3;
        const uint32_t* const px = chs + 3 * SW;
        if (L == 0) {
          if (j == 0) { pa = px; pb = px; pd = xp; }                                      // x^2
          else { pa = chs + (j == 1 ? 3 : j == 2 ? 11 : 12) * SW; pb = chs + (j == 1 ? 6 : 5) * SW; pd = shr + (j - 1) * SW; }   // x U | a P1 | b P1
        } else if (L == 1) {
          pa = j == 2 ? chs + 7 * SW : xp;                                               // x^2 x | x^2 x^2 | r x^2
          pb = j == 0 ? px : xp;
          pd = xp + (j == 0 ? 1 : j == 1 ? 2 : 5) * SW;
        } else {
          pa = xp + 2 * SW; pb = j == 0 ? px : xp; pd = xp + (j == 0 ? 3 : 4) * SW;       // x^4 x | x^4 x^2
        }
      }
      scl a, b;
      ld_scl(a, pa); ld_scl(b, pb);
      st_scl(pd, scl_mul(a, b));
    }
    __syncthreads();
  }
  PREP_STAMP(11);                                 // (small tables made)
  // ... and z^(q+1) for every constraint q, one product each
  for (uint32_t q = t; q < sh.n_cons; q += nt) {
    scl a, b;
    ld_scl(a, zlo + SW * (q & 15)); ld_scl(b, zhi + SW * (q >> 4));
    st_scl(zpow + SW * q, scl_mul(a, b));
  }
  __syncthreads();
  PREP_STAMP(2);                                  // z powers (doubling)
  // phase C: plan replay, a range of targets at a time: the products of the range (one multiplication per product, two
  // when a second-phase challenge is involved), then one sum per target.  Every stored value is a product (< 2^255) or
  // a weakly reduced sum (< 2 l).
#pragma unroll 1
  for (uint32_t ck = 0; ck < sh.n_chunks; ++ck) {
    const uint32_t g0 = sh.chunk_tgt[ck], g1 = sh.chunk_tgt[ck + 1];
    const uint32_t p0 = sh.chunk_prod[ck], p1 = sh.chunk_prod[ck + 1];
    for (uint32_t p = p0 + t; p < p1; p += nt) {
      scl c, zq;
      ld_scl(c, prod_coef + SW * (uint64_t)p);
      const uint2 qm = prod_qm[p];
      ld_scl(zq, zpow + SW * qm.x);
      if (qm.y != 0) { scl m; ld_scl(m, sym + SW * qm.y); c = scl_mul(c, m); }
      st_scl(tv + SW * (p - p0), scl_mul(c, zq));
    }
    __syncthreads();
    PREP_STAMP(8);                                // (products of the pass)
    // <= HEAVY_TERMS terms, each < 2^255 or its negative (256 l - v, limbs < 2^27.6): limbs < 2^31.6, value < 2^264.1
    auto term_value = [&](uint32_t e) {
      const uint32_t info = term_info[e];
      scl v;
      if (info & TERM_UNIT) ld_scl(v, zpow + SW * (info & TERM_IDX));
      else ld_scl(v, tv + SW * ((info & TERM_IDX) - p0));
      return scl_cneg(v, (info & TERM_NEG) != 0);
    };
    for (uint32_t g = g0 + t; g < g1; g += nt) {
      bool heavy = false;
      for (uint32_t hI = 0; hI < sh.n_heavy; ++hI) heavy |= (sh.heavy[hI] == g);
      if (heavy) continue;
      scl acc = scl_zero();
      uint32_t cnt = 0;
      for (uint32_t e = tgt_off[g]; e < tgt_off[g + 1]; ++e) {
        acc = scl_add(acc, term_value(e));
        if (++cnt == HEAVY_TERMS) { acc = scl_weak(acc); cnt = 1; }
      }
      st_scl(wv + SW * g, scl_weak(acc));
    }
    PREP_STAMP(9);                                // (sums of the light targets)
    // heavy targets: every lane sums a strided share, wavefront shuffles fold the lanes, lane 0 of
    // each wave parks its sum in the scratch slots after the products (4 per heavy target)
    for (uint32_t hI = 0; hI < sh.n_heavy; ++hI) {
      const uint32_t g = sh.heavy[hI];
      if (g < g0 || g >= g1) continue;
      scl acc = scl_zero();
      uint32_t cnt = 0;
      for (uint32_t e = tgt_off[g] + t; e < tgt_off[g + 1]; e += nt) {
        acc = scl_add(acc, term_value(e));
        if (++cnt == HEAVY_TERMS) { acc = scl_weak(acc); cnt = 1; }
      }
      acc = wave_sum_scl(scl_weak(acc));
      uint32_t* wave_sums = tv + (sh.tv_cap + 4 * hI) * SW;
      if ((t & 63) == 63) st_scl(wave_sums + SW * (t >> 6), acc);
      __syncthreads();
      if (t == 0) {
        scl tot = scl_zero();
        for (uint32_t wI = 0; wI < (nt >> 6); ++wI) { scl v; ld_scl(v, wave_sums + SW * wI); tot = scl_add(tot, v); }
        st_scl(wv + SW * g, scl_weak(tot));
      }
    }
    __syncthreads();
  }
  PREP_STAMP(3);                                  // plan replay
  // phase E (region A is dead).  The whole equation is taken times c' = rho y^(pn-1) U, U = prod u_j^2
  // (rho: chs slot 13, 1 unless the batch is checked in groups), which needs no inverse:
  //     c' y^-i        = U * yp[pn-1-i]             yp[j] = rho y^j, kept in PLAIN form
  //     c' s_i         = yp[pn-1] * P1 * sU[i]      sU[i] = prod_j u_j^(2 bit_(k-1-j)(i)),  P1 = prod u_j
  //     c' y^-i s_r    = yp[pn-1-i] * P1 * sU[r]    (U s_i = prod u_j^(2 +- 1) = P1 sU[i])
  // Both tables grow by doubling (entry + 2^L = entry * stride_L).
  for (uint32_t idx = t; idx < 2 * sh.pn; idx += nt) {
    const bool second = idx >= sh.pn;
    const uint32_t i = second ? idx - sh.pn : idx;
    scl a, b;
    ld_scl(a, (second ? slo : ylo) + SW * (i & (NLO - 1)));
    ld_scl(b, (second ? shi : yhi) + SW * (i >> LB));
    st_scl8((second ? sv : yip) + 8 * i, scl_mul(a, b));
  }
  __syncthreads();
  PREP_STAMP(4);                                  // y / s tables (doubling)
  const uint32_t* wL = wv;
  const uint32_t* wR = wv + sh.n * SW;
  const uint32_t* wO = wv + 2 * sh.n * SW;
  const uint32_t* wV = wv + 3 * sh.n * SW;
  const uint32_t* wc = wV + sh.m * SW;
  // dsum = rho sum_{i<n} y^(pn-1-i) wR_i wL_i  (= c' delta / U; plain partial sums, block reduction)
  {
    scl part = scl_zero();
    uint32_t cnt = 0;
    for (uint32_t i = t; i < sh.n; i += nt) {
      scl a, b, c;
      ld_scl8(a, yip + 8 * (sh.pn - 1 - i)); ld_scl(b, wR + SW * i); ld_scl(c, wL + SW * i);
      part = scl_add(part, scl_mul(scl_mul(b, c), a));
      if (++cnt == 32) { part = scl_weak(part); cnt = 1; }
    }
    part = wave_sum_scl(scl_weak(part));          // <= 32 products, limbs < 2^31
    if ((t & 63) == 63) st_scl(red + SW * (1 + (t >> 6)), part);
    if (t >= nt - 2) {                            // c' = U rho Y | a P1 rho Y, both plain (rho Y = yip[pn-1] is plain)
      const bool second = t == nt - 1;
      scl a, b;
      ld_scl8(a, yip + 8 * (sh.pn - 1));
      ld_scl(b, second ? shr + 1 * SW : chs + 6 * SW);
      st_scl(shr + (second ? 4 : 3) * SW, scl_mul(b, a));
      if (!second) st_scl(shr + 6 * SW, a);       // rho Y, unpacked
    }
    __syncthreads();
    if (t == 0) {                                 // the sum, converted plain -> Montgomery
      const uint32_t z = lane_zero();
      scl tot = scl_zero();
      for (uint32_t wI = 0; wI < (nt >> 6); ++wI) { scl v; ld_scl(v, red + SW * (1 + wI) + z); tot = scl_add(tot, v); }
      st_scl(red, scl_mul(scl_weak(tot), scl_r2()));
    } else if (t >= nt - 6) {                     // c' in Montgomery form | the generator loop's factors times u
      const uint32_t w = nt - 1 - t;              // 0: c' R^2 | 1: (x U) u | 2: (a P1 rho Y) u | 3: (b P1) u | 4: U u | 5: c' u
      scl a, b;
      ld_scl(a, w == 0 || w == 5 ? shr + 3 * SW : w == 1 ? shr + 0 * SW : w == 2 ? shr + 4 * SW : w == 3 ? shr + 2 * SW : chs + 6 * SW);
      ld_scl(b, chs + 2 * SW);
      const scl r2 = scl_r2();
#pragma unroll
      for (int q = 0; q < 10; ++q) b.v[q] = w == 0 ? r2.v[q] : b.v[q];
      st_scl(shr + (w == 0 ? 5 : 7 + w) * SW, scl_mul(a, b));
    }
    __syncthreads();
  }
  PREP_STAMP(5);                                  // dsum, c'
  uint32_t* ds = dyn_scalars + (uint64_t)tx * sh.n_dyn * 8;
  uint32_t* dr = dyn_recoded + (uint64_t)tx * sh.n_dyn * 8;
  uint32_t* ss = static_scalars + (uint64_t)tx * sh.n_static * 8;
  // ---- proof-point scalars, B and B_blinding: the last wavefront, BEFORE its share of the generator scalars, so that
  // this short serial tail runs beside the other wavefronts' generator loop instead of after it.  Every lane does the
  // same two rounds v = a * b with operands of its own, read from LDS where they are needed (a lane that needs fewer
  // rounds passes its value through), and the conversion; the scalar of B is spread over four lanes: with c' delta = U dsum,
  //     c' (w (t_x - a b) + r (x^2 (wc + delta) - t_x))  =  c' [w (t_x - [a b])]  +  [r x^2] ([c' wc] + [U dsum]) - r [c' t_x]
  // lane jB: a b, then w (t_x - .), converted TIMES the plain c';  jB+1: c' wc, then r x^2 (. + U dsum), minus jB+3's product,
  // converted;  jB+2: U dsum;  jB+3: c' t_x, then r (.);  the partial results travel by wavefront shuffles and the two halves
  // are added as plain values.  The factor c' of everything else rides on the final Montgomery -> plain conversion too (a
  // product with the plain c' instead of with 1).
  const uint32_t tail0 = nt - 64;
  if (t >= tail0) {
    const uint32_t lane = t - tail0, n_dyn = sh.n_dyn;
    const uint32_t* const p_u = chs + 2 * SW;
    const uint32_t* const p_x = chs + 3 * SW;
    const uint32_t* const p_U = chs + 6 * SW;
    const uint32_t* const p_r = chs + 7 * SW;
    const uint32_t* const p_cp = shr + 5 * SW;     // c', Montgomery
    // positions: 0 B (a b ..), 1 B's second half (c' wc ..), 2 U dsum, 3 c' t_x, 4 B_blinding, 5 + j the proof point j
#pragma unroll 1
    for (uint32_t pos0 = 0; pos0 < n_dyn + 5; pos0 += 64) {
      const uint32_t pos = pos0 + lane;
      const uint32_t j = pos - 5;                 // proof-point index when pos >= 5
      const uint32_t* pa = p_x;
      const uint32_t* pb = p_x;
      const uint32_t* pconv = shr + 3 * SW;       // c', plain
      bool m1 = true;                             // does round 1 multiply?  (otherwise v = a)
      if (pos == 0) { pa = chs + 11 * SW; pb = chs + 12 * SW; }                                              // a b
      else if (pos == 1) { pa = p_cp; pb = wc; }                                                             // c' wc
      else if (pos == 2) { pa = p_U; pb = red; }                                                             // U dsum
      else if (pos == 3) { pa = p_cp; pb = chs + 8 * SW; }                                                   // c' t_x
      else if (pos == 4) { pa = p_r; pb = chs + 9 * SW; }                                                    // r t_x_blinding
      else if (j < 3) { if (j) pa = xp + (j - 1) * SW; m1 = false; }                                         // x, x^2, x^3
      else if (j < 6) { if (j > 3) pa = xp + (j - 4) * SW; pb = p_u; }                                       // u x^(1..3)
      else if (j < 6 + sh.m) { pa = wV + SW * (j - 6); pb = xp + 5 * SW; }                                   // wV_j r x^2
      else if (j < 11 + sh.m) {                                                                              // r x, r x^3 .. r x^6
        const uint32_t q = j - 6 - sh.m;
        if (q) pa = xp + q * SW;
        pb = p_r;
      } else if (j < n_dyn) {
        const uint32_t q = j - 11 - sh.m;         // u_j^2 for L_j; for R_j  c' u_j^-2 = rho Y prod_{l != j} u_l^2
        if (q < sh.k) { pa = pb = chs + (CH_FIXED + sh.n_chal2 + q) * SW; }
        else { pa = chs + (CH_FIXED + sh.n_chal2 + sh.k + (q - sh.k)) * SW; pconv = shr + 6 * SW; m1 = false; }
      } else m1 = false;
      scl a, b, v;
      ld_scl(a, pa); ld_scl(b, pb);
      v = a;
      {
        const scl pr = scl_mul(a, b);
        if (m1) v = pr;
      }
      if (pos0 == 0) {                            // round 2: the scalars of B and B_blinding (first pass only, lanes 0..4)
        scl other = shfl_down_scl(v, 1);
        bool m2 = false;
        if (pos == 0) { scl tx_; ld_scl(tx_, chs + 8 * SW); ld_scl(a, chs + 4 * SW); b = scl_sub(tx_, v); m2 = true; }       // w (t_x - a b)
        else if (pos == 1) { ld_scl(a, xp + 5 * SW); b = scl_add(v, other); m2 = true; }                                      // r x^2 (c' wc + U dsum)
        else if (pos == 3) { ld_scl(a, p_r); b = v; m2 = true; }                                                             // r c' t_x
        else if (pos == 4) { scl e; ld_scl(e, chs + 10 * SW); scl sum = scl_add(e, v); scl_carry(sum); v = scl_neg(sum); }    // -(e_blinding + r t_x_blinding)
        const scl pr2 = scl_mul(a, b);
        if (m2) v = pr2;
        other = shfl_down_scl(v, 2);
        if (pos == 1) { v = scl_sub(v, other); pconv = shr + 7 * SW; }                // r (x^2 (..) - c' t_x), converted as it is
      }                                                                               // (lane 0: w (t_x - a b), converted times c')
      scl conv_by;
      ld_scl(conv_by, pconv);
      scl plain = scl_mul(v, conv_by);            // Montgomery -> plain, times the plain factor
      if (pos0 == 0) {                            // B's two halves meet as plain values
        const scl other = shfl_down_scl(plain, 1);
        if (pos == 0) plain = scl_add(plain, other);
      }
      uint32_t o[8];
      scl_canon_words(o, plain);
      if (pos >= 5 && j < n_dyn) {
        // the scalar, and its recoded form s + 0x88..8 for k_small_accumulate (digit t = nibble t - 8)
        uint32_t carry = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          ds[j * 8 + q] = o[q];
          const uint64_t vv = (uint64_t)o[q] + 0x88888888u + carry;
          dr[j * 8 + q] = (uint32_t)vv;
          carry = (uint32_t)(vv >> 32);
        }
      } else if (pos == 0 || pos == 4) {
#pragma unroll
        for (int q = 0; q < 8; ++q) ss[(pos == 0 ? 0 : 1) * 8 + q] = o[q];
      }
    }
  }
  PREP_STAMP(6);                                  // (thread 255: the proof-point scalars of the last wavefront)
  // generator scalars (times c'), reduced to canonical words at the very end:
  //   c' g_i = (x U) wR_i yp[pn-1-i] - (a P1 rho Y) sU_i
  //   c' h_i = yp[pn-1-i] ((x U) wL_i + U wO_i - (b P1) sU_(pn-1-i)) - c'        (times u for i >= n1)
  {
    // the factors the whole workgroup shares live in scalar registers.  For i >= n1 (second-phase multipliers and the padding)
    // both scalars carry the factor u: a wavefront whose lanes are all there reads the copies that already carry it (slots
    // 8 .. 12: two products less per i), a mixed one multiplies at the end.  (Read per lane from LDS, whichever copy the lane
    // needs, the factors take 50 vector registers more and the kernel spills.)
    for (uint32_t i = t; i < sh.pn; i += nt) {
      const bool hi = i >= sh.n1;
      const bool with_u = __builtin_amdgcn_readfirstlane((int)__all(hi)) != 0;
      const uint32_t* const f = shr + 8 * SW;
      scl xU, aY_plain, bP, U, cp_plain;
      ld_scl_shared(xU, with_u ? f + 0 * SW : shr + 0 * SW); ld_scl_shared(aY_plain, with_u ? f + 1 * SW : shr + 4 * SW);
      ld_scl_shared(bP, with_u ? f + 2 * SW : shr + 2 * SW); ld_scl_shared(U, with_u ? f + 3 * SW : chs + 6 * SW);
      ld_scl_shared(cp_plain, with_u ? f + 4 * SW : shr + 3 * SW);
      scl yp, si, sr;
      ld_scl8(yp, yip + 8 * (sh.pn - 1 - i)); ld_scl8(si, sv + 8 * i); ld_scl8(sr, sv + 8 * (sh.pn - 1 - i));
      scl g = scl_neg(scl_mul(aY_plain, si));                 // limbs < 2^27.6, value < 2^260.1
      scl inner = scl_neg(scl_mul(bP, sr));
      if (i < sh.n) {
        scl wl, wr, wo;
        ld_scl(wl, wL + SW * i); ld_scl(wr, wR + SW * i); ld_scl(wo, wO + SW * i);
        g = scl_add(g, scl_mul(scl_mul(xU, wr), yp));
        inner = scl_add(inner, scl_add(scl_mul(xU, wl), scl_mul(U, wo)));   // limbs < 2^28, value < 2^260.2
      }
      scl h = scl_sub(scl_mul(yp, inner), cp_plain);          // yp, c' tight and < 2^255
      if (!with_u && __any(hi)) {                             // (a wavefront that straddles n1)
        scl u;
        ld_scl_shared(u, chs + 2 * SW);
        const scl gu = scl_mul(g, u), hu = scl_mul(h, u);
        if (hi) { g = gu; h = hu; }
      }
      uint32_t gw[8], hw[8];
      scl_canon_words(gw, g);
      scl_canon_words(hw, h);
      uint4* og = reinterpret_cast<uint4*>(ss + (2 + i) * 8);
      uint4* oh = reinterpret_cast<uint4*>(ss + (2 + sh.pn + i) * 8);
      og[0] = make_uint4(gw[0], gw[1], gw[2], gw[3]); og[1] = make_uint4(gw[4], gw[5], gw[6], gw[7]);
      oh[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]); oh[1] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
    }
  }
  PREP_STAMP(7);                                  // generator scalars
}

// the proof-specific points of every transaction in the order of the MSM's dynamic terms
// [A_I1 A_O1 S1 A_I2 A_O2 S2 | V.. | T_1 T_3 T_4 T_5 T_6 | L.. | R..]: needs the proof bytes only, so the
// decompression and the per-point tables run while the transcript is still being replayed
__global__ void __launch_bounds__(256)
k_gather_dyn_points(PrepShape sh, const uint32_t* __restrict__ com, const uint32_t* __restrict__ pw, uint32_t batch,
                    uint32_t* __restrict__ dyn_points) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)batch * sh.n_dyn * 8) return;
  const uint32_t q = (uint32_t)(g & 7), j = (uint32_t)((g >> 3) % sh.n_dyn), tx = (uint32_t)((g >> 3) / sh.n_dyn);
  const uint32_t* p = pw + (uint64_t)tx * sh.proof_words;
  const uint32_t* c = com + (uint64_t)tx * sh.m * 8;
  const uint32_t* src;
  if (j < 6) src = p + 8 * j;
  else if (j < 6 + sh.m) src = c + 8 * (j - 6);
  else if (j < 11 + sh.m) src = p + 8 * (6 + (j - 6 - sh.m));
  else {
    const uint32_t r = j - 11 - sh.m;
    src = p + 112 + (r < sh.k ? 16 * r : 16 * (r - sh.k) + 8);
  }
  dyn_points[g] = src[q];
}

// The same gather fused with the RFC 9496 DECODE and the per-point tables of the small-MSM path
// (kernels.hpp, k_small_tables): one launch instead of three on the shared stream -- under load every
// small kernel there waits for CU slots behind the long ones, and the waits add up along the chain.
// (two wavefronts per SIMD: left to itself the compiler takes 315 registers -- ONE wavefront per SIMD, and the decoding is a
// chain of 254 dependent squarings; capped at 256 registers (24 spilled) the kernel runs 0.53 -> 0.43 ms per 8192
// transactions; three wavefronts (168 registers, 118 spilled): 0.44 ms)
__global__ void __launch_bounds__(256, 2)
k_points_tables(PrepShape sh, const uint32_t* __restrict__ com, const uint32_t* __restrict__ pw, uint32_t batch,
                uint32_t* __restrict__ tbl /*[B n_dyn][8][40]*/, uint32_t* __restrict__ msm_fail,
                unsigned long long* __restrict__ bad_index) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)batch * sh.n_dyn) return;
  const uint32_t j = (uint32_t)(g % sh.n_dyn), tx = (uint32_t)(g / sh.n_dyn);
  const uint32_t* p = pw + (uint64_t)tx * sh.proof_words;
  const uint32_t* c = com + (uint64_t)tx * sh.m * 8;
  const uint32_t* src;
  if (j < 6) src = p + 8 * j;
  else if (j < 6 + sh.m) src = c + 8 * (j - 6);
  else if (j < 11 + sh.m) src = p + 8 * (6 + (j - 6 - sh.m));
  else {
    const uint32_t r = j - 11 - sh.m;
    src = p + 112 + (r < sh.k ? 16 * r : 16 * (r - sh.k) + 8);
  }
  uint32_t w[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) w[q] = src[q];
  fe x, y;
  const bool ok = ristretto_decode_affine(x, y, w);
  ge_niels nq;
  niels_from_affine(nq, x, y);
  if (!ok) {
    niels_identity(nq);
    atomicMin(bad_index, (unsigned long long)g);
    atomicOr(&msm_fail[tx], 1u);
  }
  ge cur;
  ge_identity(cur);
  uint32_t* row = tbl + g * (SMALL_TBL * EXT_WORDS);
#pragma unroll 1
  for (int e = 0; e < SMALL_TBL; ++e) {
    ge_madd(cur, cur, nq, false);
    store_cached(row + e * EXT_WORDS, cur);
  }
}

// generator scalars of a group: sum over its transactions (canonical words in and out).  both != 0: also the LOCATING sum
// sum_t i_t s_t (i_t = 1, 2, .. the position in the group) as rows n_groups + G of the same multiscalar multiplication:
// outputs and digits then hold 2 n_groups rows
__global__ void __launch_bounds__(256)
k_group_scalars(const uint32_t* __restrict__ st_scalars /*[B][n_static][8]*/, uint32_t n_msm, uint32_t n_static,
                uint32_t group, uint32_t* __restrict__ out /*[groups (x 2)][n_static][8]*/,
                int16_t* __restrict__ digits /*optional: [W][rows * n_static], as k_static_digits writes them*/,
                int w, int W, const uint32_t* __restrict__ msm_fail, const uint32_t* __restrict__ wellformed, uint32_t both) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_groups = (n_msm + group - 1) / group;
  const uint64_t total = (uint64_t)n_groups * n_static, stride = both ? 2 * total : total;
  if (g >= total) return;
  const uint32_t G = (uint32_t)(g / n_static), j = (uint32_t)(g % n_static);
  scm acc = scm_zero(), sum2 = scm_zero();
  const uint32_t in_group = min(group, n_msm - G * group);
  for (uint32_t i = in_group; i-- > 0;) {         // acc: running sum from the last transaction down; sum_k (sum_{t >= k} s_t) = sum_t (t + 1) s_t
    const uint32_t tx = G * group + i;
    if (!tx_excluded(msm_fail, wellformed, tx)) {   // known bad already: left out of the group, rejected on the spot
      const uint4* src = reinterpret_cast<const uint4*>(st_scalars + ((uint64_t)tx * n_static + j) * 8);
      const uint4 a = src[0], b = src[1];
      scm v;
      v.v[0] = a.x; v.v[1] = a.y; v.v[2] = a.z; v.v[3] = a.w; v.v[4] = b.x; v.v[5] = b.y; v.v[6] = b.z; v.v[7] = b.w;
      acc = scm_add(acc, v);
    }
    if (both) sum2 = scm_add(sum2, acc);
  }
  auto emit = [&](uint64_t at, const scm& val) {
    uint4* dst = reinterpret_cast<uint4*>(out + at * 8);
    dst[0] = make_uint4(val.v[0], val.v[1], val.v[2], val.v[3]);
    dst[1] = make_uint4(val.v[4], val.v[5], val.v[6], val.v[7]);
    if (digits) {   // the sum is < l: no range flag to raise
      for (int t = 0; t < W; ++t) digits[(uint64_t)t * stride + at] = 0;
      for_each_digit(out + at * 8, w, W, [&](int t, int d) { digits[(uint64_t)t * stride + at] = (int16_t)d; });
    }
  };
  emit(g, acc);
  if (both) emit(total + g, sum2);
}

}  // namespace zk
