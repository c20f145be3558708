// prover_kernels.hpp -- the phase functions of prover_dev.hpp as kernels: one workgroup per proof, thread 0 on the
// transcript, all threads on the vectors.
#pragma once
#include "prover_dev.hpp"

namespace zk {

constexpr uint32_t PV_LDS_WORDS = 52 + 4 * 6 * 10;   // STROBE scratch | per-wavefront partial sums

struct PvDevEnv {
  static constexpr bool kInvertInEveryLane = false;
  uint32_t* lds;
  __device__ __forceinline__ uint32_t tid() const { return threadIdx.x; }
  __device__ __forceinline__ uint32_t nt() const { return blockDim.x; }
  __device__ __forceinline__ void sync() { __threadfence_block(); __syncthreads(); }
  __device__ __forceinline__ uint32_t* strobe() { return lds; }
  // sums over the workgroup, left in thread 0 (inputs tight and < 2 l; at most four wavefronts, count <= 6)
  __device__ inline void sum(scl* v, int count) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = (blockDim.x + 63) >> 6;
    for (int k = 0; k < count; ++k) {
#pragma unroll 1
      for (int delta = 32; delta >= 1; delta >>= 1) {
        scl o;
#pragma unroll
        for (int q = 0; q < 10; ++q) o.v[q] = (uint32_t)__shfl_down((int)v[k].v[q], delta);
        v[k] = scl_add_c(v[k], o);
      }
      if (lane == 0) for (int q = 0; q < 10; ++q) lds[52 + (wave * 6 + k) * 10 + q] = v[k].v[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int k = 0; k < count; ++k)
        for (uint32_t wv = 1; wv < waves; ++wv) {
          scl o;
          for (int q = 0; q < 10; ++q) o.v[q] = lds[52 + (wv * 6 + k) * 10 + q];
          v[k] = scl_add_c(v[k], o);
        }
    }
    __syncthreads();
  }
};

__global__ void __launch_bounds__(64)
k_pv_phase0(PvShape sh, PvBatch B) {
  __shared__ uint32_t lds[PV_LDS_WORDS];
  PvDevEnv env{lds};
  pv_phase0(env, sh, B, blockIdx.x);
}
// the TranscriptRng's draws of a commitment phase, one LANE per proof (state in registers, one Keccak-f per draw)
__global__ void __launch_bounds__(64)
k_pv_rng(PvShape sh, PvBatch B, uint32_t batch, uint32_t phase) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= batch) return;
  uint32_t* state = B.state + (uint64_t)p * sh.state_words;
  if (phase == 1) pv_rng_draw(sh, state, B.rows1 + (uint64_t)p * sh.r1_terms * 8, 0, sh.n1, PV_IBL1);
  else pv_rng_draw(sh, state, B.rows2 + (uint64_t)p * sh.r2_terms * 8, sh.n1, sh.n, PV_IBL2);
}
// ... and one WAVEFRONT per proof (PvRngCoop): the permutations on the spread state, 256 draws at a time: their 64-byte
// outputs parked in LDS (16 KB), then the reductions mod l and the stores one lane per draw.
constexpr uint32_t PV_RNG_CHUNK = 256;
__global__ void __launch_bounds__(64)
k_pv_rng_coop(PvShape sh, PvBatch B, uint32_t batch, uint32_t phase) {
  __shared__ __attribute__((aligned(16))) uint32_t wide[PV_RNG_CHUNK * 16];
  const uint32_t p = blockIdx.x, lane = threadIdx.x;
  if (p >= batch) return;
  uint32_t* state = B.state + (uint64_t)p * sh.state_words;
  const uint32_t first = phase == 1 ? 0 : sh.n1, last = phase == 1 ? sh.n1 : sh.n, bl_slot = phase == 1 ? PV_IBL1 : PV_IBL2;
  uint32_t* rows = phase == 1 ? B.rows1 + (uint64_t)p * sh.r1_terms * 8 : B.rows2 + (uint64_t)p * sh.r2_terms * 8;
  const uint32_t cnt = last - first, n_draws = 3 + 2 * cnt;
  const coop::KcLane k = coop::kc_lane(lane);
  const coop::KeccakCoop<DevKcTraits>::Consts c = {k.live, k.rot_swap, k.rot_t, k.src[0], k.src[1], k.src[2], k.iota};
  auto holds = [&k](uint32_t q) { return (k.live && k.q == q) ? ~0u : 0u; };
  const PvRngCoop<DevKcTraits>::Masks m = {holds(4), holds(5), holds(8), holds(9), holds(20), (k.live && k.q < 8) ? 0u : ~0u};
  uint32_t lo = k.live ? state[sh.o_rng + 2 * k.q] : 0, hi = k.live ? state[sh.o_rng + 2 * k.q + 1] : 0;
  const uint32_t pos = state[sh.o_rng + 50], pos_begin = state[sh.o_rng + 51];
  if (pos_begin != 0 || (pos != 32 && pos != 64)) { if (lane == 0) state[sh.o_flag] = 2; return; }
  PvView V{sh, state};
  uint32_t* rI = rows;
  uint32_t* rO = rI + 8 * (1 + 2 * cnt);
  uint32_t* rS = rO + 8 * (1 + cnt);
#pragma unroll 1
  for (uint32_t d0 = 0; d0 < n_draws; d0 += PV_RNG_CHUNK) {
    const uint32_t d1 = min(d0 + PV_RNG_CHUNK, n_draws);
#pragma unroll 1
    for (uint32_t d = d0; d < d1; ++d) {
      PvRngCoop<DevKcTraits>::draw(lo, hi, c, m, d == 0 && pos == 32);
      if (k.primary && k.q < 8) { wide[16 * (d - d0) + 2 * k.q] = lo; wide[16 * (d - d0) + 2 * k.q + 1] = hi; }
      PvRngCoop<DevKcTraits>::taken(lo, hi, m);
    }
    __syncthreads();
    for (uint32_t d = d0 + lane; d < d1; d += 64) {
      uint32_t wd[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) wd[q] = wide[16 * (d - d0) + q];
      const scm v = scm_from_wide(wd);
      uint32_t *st_slot, *row_slot;
      if (d == 0) { st_slot = V.at(sh.o_blind, bl_slot); row_slot = rI; }
      else if (d == 1) { st_slot = V.at(sh.o_blind, bl_slot + 1); row_slot = rO; }
      else if (d == 2) { st_slot = V.at(sh.o_blind, bl_slot + 2); row_slot = rS; }
      else if (d < 3 + cnt) { st_slot = V.at(sh.o_sL, first + (d - 3)); row_slot = rS + 8 * (1 + (d - 3)); }
      else { st_slot = V.at(sh.o_sR, first + (d - 3 - cnt)); row_slot = rS + 8 * (1 + cnt + (d - 3 - cnt)); }
      pv_st(st_slot, v);
      pv_st_plain(row_slot, scl_from_scm(v));
    }
    __syncthreads();
  }
  if (k.primary) { state[sh.o_rng + 2 * k.q] = lo; state[sh.o_rng + 2 * k.q + 1] = hi; }
  if (lane == 0) { state[sh.o_rng + 50] = 64; state[sh.o_rng + 51] = 0; }
}
// A round of the inner-product argument with one LANE per proof: the round is one thread's work (L_j, R_j into the transcript,
// the challenge, its inverse), and with a workgroup per proof (k_pv_ipa, until round 5) 63 lanes of every wavefront idled
// through ~50 000 instructions -- a fifth of all the wavefront instructions of a proving call (profiles/archive/r05_proverprog_*).  STROBE states side by side in LDS (53 words apart:
// no bank conflicts), the inverse by the fixed chain pv_invert_uniform.  Byte-identical proofs.
struct PvLaneEnv {
  static constexpr bool kInvertInEveryLane = true;
  uint32_t* st;
  __device__ __forceinline__ uint32_t tid() const { return 0; }
  __device__ __forceinline__ uint32_t nt() const { return 1; }
  __device__ __forceinline__ void sync() {}
  __device__ __forceinline__ uint32_t* strobe() { return st; }
  __device__ __forceinline__ void sum(scl*, int) {}
};
__global__ void __launch_bounds__(64)
k_pv_ipa_lanes(PvShape sh, PvBatch B, uint32_t round, const uint32_t* __restrict__ points, uint32_t batch) {
  __shared__ uint32_t lds[64 * 53];
  const uint32_t proof = blockIdx.x * 64 + threadIdx.x;
  if (proof >= batch) return;
  PvLaneEnv env{lds + threadIdx.x * 53};
  pv_ipa_round(env, sh, B, proof, round, points + (uint64_t)proof * 16);
}

// The stages of phases 1 .. 4 (prover_dev.hpp, "STAGES of a phase"): PH the phase, ST the stage mask.  k_pv_lanes: a stage
// that is one thread's work, one lane per proof; k_pv_wg: a stage of the whole workgroup, one workgroup per proof.
template <int PH, uint32_t ST, class Env>
__device__ __forceinline__ void pv_stage(Env& env, const PvShape& sh, const PvPlan& P, const PvBatch& B, uint32_t proof, const uint32_t* points) {
  if (PH == 1) pv_phase1(env, sh, P, B, proof, points + (uint64_t)proof * sh.m * 8, ST);
  else if (PH == 2) pv_phase2(env, sh, P, B, proof, points + (uint64_t)proof * 24, ST);
  else if (PH == 3) pv_phase3(env, sh, P, B, proof, points + (uint64_t)proof * 24, ST);
  else pv_phase4(env, sh, P, B, proof, points + (uint64_t)proof * 40, ST);
}
template <int PH, uint32_t ST>
__global__ void __launch_bounds__(64)
k_pv_lanes(PvShape sh, PvPlan P, PvBatch B, const uint32_t* __restrict__ points, uint32_t batch) {
  __shared__ uint32_t lds[64 * 53];
  const uint32_t proof = blockIdx.x * 64 + threadIdx.x;
  if (proof >= batch) return;
  PvLaneEnv env{lds + threadIdx.x * 53};
  pv_stage<PH, ST>(env, sh, P, B, proof, points);
}
template <int PH, uint32_t ST>
__global__ void __launch_bounds__(256)
k_pv_wg(PvShape sh, PvPlan P, PvBatch B, const uint32_t* __restrict__ points) {
  __shared__ uint32_t lds[PV_LDS_WORDS];
  PvDevEnv env{lds};
  pv_stage<PH, ST>(env, sh, P, B, blockIdx.x, points);
}
__global__ void __launch_bounds__(256)
k_pv_finish(PvShape sh, PvBatch B, const uint32_t* __restrict__ ab, uint32_t batch, uint32_t* __restrict__ status) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= batch) return;
  pv_finish(sh, B, p, ab + (uint64_t)p * 16);
  const uint32_t flag = B.state[(uint64_t)p * sh.state_words + sh.o_flag];
  if (flag) atomicOr(status, flag == 2 ? 2u : 1u);
}


// ---- test hook: the arithmetic layers alone (zkgpu_debug_arith), one lane per element ------------------------
// op: 0 fe_mul  1 fe_sq  2 fe_invert  3 fe_add then fe_sub (a + b - b + a)  4 fe_pow22523
//     10 scm product  11 scl product  12 scl chain ((a - b) (a + b) + 16 a b - b, lazily)  13 inverse mod l (scm_invert)
//     14 inverse mod l (pv_invert: Euclid)  15 scm sum / difference (a + b, then - a)  16 inverse mod l (pv_invert_uniform)
// field elements travel as 32 little-endian bytes (bit 255 ignored on input, canonical on output), scalars as
// canonical words (inputs are reduced mod l first).
__global__ void __launch_bounds__(64)
k_debug_arith(uint32_t op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wa[8], wb[8], wo[8];
  for (int q = 0; q < 8; ++q) { wa[q] = a[8 * i + q]; wb[q] = b[8 * i + q]; wo[q] = 0; }
  if (op < 10) {
    fe x, y, r;
    fe_from_words(x, wa);
    fe_from_words(y, wb);
    switch (op) {
      case 0: fe_mul(r, x, y); break;
      case 1: fe_sq(r, x); break;
      case 2: fe_invert(r, x); break;
      case 3: { fe t; fe_add(t, x, y); fe_carry(t); fe_sub(r, t, y); fe_carry(r); fe_add(r, r, x); fe_carry(r); break; }
      default: fe_pow22523(r, x); break;
    }
    fe_to_words(wo, r);
  } else {
    const scm x = scm_from_words(wa), y = scm_from_words(wb);
    scm r = scm_zero();
    switch (op) {
      case 10: r = scm_mul(x, y); break;
      case 11: r = scl_to_scm(scl_mul(scl_from_scm(x), scl_from_scm(y))); break;
      case 12: {
        const scl lx = scl_from_scm(x), ly = scl_from_scm(y);
        scl acc = scl_mul(scl_sub(lx, ly), scl_add(lx, ly));
        const scl xy = scl_mul(lx, ly);
        for (int k = 0; k < 16; ++k) acc = scl_add(acc, xy);
        r = scl_to_scm(scl_sub(acc, ly));
        break;
      }
      case 13: r = scm_invert(x); break;
      case 14: r = pv_invert(x); break;
      case 16: r = pv_invert_uniform(x); break;
      default: r = scm_sub(scm_add(x, y), x); break;
    }
    scm_to_words(wo, r);
  }
  for (int q = 0; q < 8; ++q) out[8 * i + q] = wo[q];
}

}  // namespace zk
