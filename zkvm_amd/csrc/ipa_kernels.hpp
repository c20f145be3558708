// ipa_kernels.hpp -- the inner-product argument of the R1CS PROVER on the device (SURVEY.md sec 8 row f-4;
// upstream bulletproofs `InnerProductProof::create`, as restated in r1cs_prover.hpp / oracle/r1cs.c).
//
// Per proof the argument folds two scalar vectors l, r of length pn in k = lg pn rounds and emits two points
// L_j, R_j per round.  r1cs_prover.hpp keeps the folded GENERATORS as coefficient vectors cG, cH over the
// original ones, so every L_j / R_j is a multiscalar multiplication over the resident generator tables; what is
// left per round is elementwise scalar algebra over four vectors of pn entries -- done here, one workgroup
// per proof -- and a three-message transcript step that stays on the host (two appends and a challenge: ~4
// Keccak-f per proof and round, against ~6 pn scalar products).
//
// Device state per proof (32-byte slots): LV, RV in Montgomery form; CG, CH as PLAIN words -- a Montgomery
// product with one plain operand is plain, so the multiscalar-multiplication scalars lv * cG, rv * cH come
// out canonical without a conversion product, and so do the updates cG * u.
#pragma once
#include "sc_dev.hpp"

namespace zk {

__device__ __forceinline__ void ipa_ld(scm& s, const uint32_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  s.v[0] = a.x; s.v[1] = a.y; s.v[2] = a.z; s.v[3] = a.w; s.v[4] = b.x; s.v[5] = b.y; s.v[6] = b.z; s.v[7] = b.w;
}
__device__ __forceinline__ void ipa_st(uint32_t* p, const scm& s) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(s.v[0], s.v[1], s.v[2], s.v[3]);
  q[1] = make_uint4(s.v[4], s.v[5], s.v[6], s.v[7]);
}

// canonical words -> Montgomery form, in place (LV, RV after the upload)
__global__ void __launch_bounds__(256)
k_ipa_to_mont(uint32_t* __restrict__ v, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  scm s;
  ipa_ld(s, v + 8 * i);
  ipa_st(v + 8 * i, scm_from_words(s.v));
}

// One workgroup per proof and round.  fold != 0: first fold the vectors of length 2 * len with the previous
// round's challenge (u, 1/u: canonical words per proof):
//     lv[j] = lv[j] u + lv[len + j] / u,  rv[j] = rv[j] / u + rv[len + j] u           (j < len)
//     cG[idx] *= hi ? u : 1/u,  cH[idx] *= hi ? 1/u : u     (hi: idx mod 2 len >= len; all pn entries)
// then, if rows != 0, the scalars of this round's L and R over the ORIGINAL generators (half = len / 2):
//     idx mod len >= half:  L gets lv[jj] cG[idx] on G_idx,        R gets rv[jj] cH[idx] on H_idx
//     else:                 L gets rv[half + jj] cH[idx] on H_idx,  R gets lv[half + jj] cG[idx] on G_idx
//     and both the weighted cross terms  <lv_lo, rv_hi> w  resp.  <lv_hi, rv_lo> w  on B
// written as canonical scalars + generator indices, rows 2 p (L) and 2 p + 1 (R) of pn + 1 terms each.
// rows == 0 (after the last fold): out_ab[p] = (lv[0], rv[0]) canonical.
__global__ void __launch_bounds__(256)
k_ipa_round(uint32_t* __restrict__ LV, uint32_t* __restrict__ RV, uint32_t* __restrict__ CG, uint32_t* __restrict__ CH,
            const uint32_t* __restrict__ W /*[batch][8] canonical*/, const uint32_t* __restrict__ U /*[batch][16]: u, 1/u canonical*/,
            uint32_t pn, uint32_t len, uint32_t gens_capacity, uint32_t fold, uint32_t rows,
            uint32_t* __restrict__ st_scalars, uint32_t* __restrict__ st_index, uint32_t* __restrict__ out_ab) {
  __shared__ uint32_t red[2][4][8];
  const uint32_t p = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  uint32_t* lv = LV + (uint64_t)p * pn * 8;
  uint32_t* rv = RV + (uint64_t)p * pn * 8;
  uint32_t* cg = CG + (uint64_t)p * pn * 8;
  uint32_t* ch = CH + (uint64_t)p * pn * 8;
  if (fold) {
    scm u, ui;
    ipa_ld(u, U + (uint64_t)p * 16);
    ipa_ld(ui, U + (uint64_t)p * 16 + 8);
    u = scm_from_words(u.v);
    ui = scm_from_words(ui.v);
    for (uint32_t j = t; j < len; j += nt) {
      scm a, b, c, d;
      ipa_ld(a, lv + 8 * j); ipa_ld(b, lv + 8 * (len + j)); ipa_ld(c, rv + 8 * j); ipa_ld(d, rv + 8 * (len + j));
      ipa_st(lv + 8 * j, scm_add(scm_mul(a, u), scm_mul(b, ui)));
      ipa_st(rv + 8 * j, scm_add(scm_mul(c, ui), scm_mul(d, u)));
    }
    for (uint32_t idx = t; idx < pn; idx += nt) {
      const bool hi = (idx % (2 * len)) >= len;
      scm g, h;
      ipa_ld(g, cg + 8 * idx); ipa_ld(h, ch + 8 * idx);
      ipa_st(cg + 8 * idx, scm_mul(g, hi ? u : ui));       // plain x Montgomery = plain
      ipa_st(ch + 8 * idx, scm_mul(h, hi ? ui : u));
    }
    __threadfence_block();
    __syncthreads();
  }
  if (!rows) {
    if (t == 0) {
      scm a, b;
      ipa_ld(a, lv); ipa_ld(b, rv);
      uint32_t wa[8], wb[8];
      scm_to_words(wa, a); scm_to_words(wb, b);
      for (int q = 0; q < 8; ++q) { out_ab[(uint64_t)p * 16 + q] = wa[q]; out_ab[(uint64_t)p * 16 + 8 + q] = wb[q]; }
    }
    return;
  }
  const uint32_t half = len / 2, row_len = pn + 1;
  uint32_t* sL = st_scalars + (uint64_t)(2 * p) * row_len * 8;
  uint32_t* sR = st_scalars + (uint64_t)(2 * p + 1) * row_len * 8;
  uint32_t* iL = st_index + (uint64_t)(2 * p) * row_len;
  uint32_t* iR = st_index + (uint64_t)(2 * p + 1) * row_len;
  for (uint32_t idx = t; idx < pn; idx += nt) {
    const uint32_t j = idx % len;
    const bool hi = j >= half;
    const uint32_t jj = hi ? j - half : j;
    scm g, h, a, b;
    ipa_ld(g, cg + 8 * idx); ipa_ld(h, ch + 8 * idx);
    if (hi) {
      ipa_ld(a, lv + 8 * jj); ipa_ld(b, rv + 8 * jj);
      ipa_st(sL + 8 * idx, scm_mul(a, g)); iL[idx] = 2 + idx;                        // a_L on G_R
      ipa_st(sR + 8 * idx, scm_mul(b, h)); iR[idx] = 2 + gens_capacity + idx;        // b_L on H_R
    } else {
      ipa_ld(a, lv + 8 * (half + jj)); ipa_ld(b, rv + 8 * (half + jj));
      ipa_st(sL + 8 * idx, scm_mul(b, h)); iL[idx] = 2 + gens_capacity + idx;        // b_R on H_L
      ipa_st(sR + 8 * idx, scm_mul(a, g)); iR[idx] = 2 + idx;                        // a_R on G_L
    }
  }
  // cL = <lv[0..half), rv[half..len)>, cR = <lv[half..len), rv[0..half)>
  scm cl = scm_zero(), cr = scm_zero();
  for (uint32_t j = t; j < half; j += nt) {
    scm a0, a1, b0, b1;
    ipa_ld(a0, lv + 8 * j); ipa_ld(a1, lv + 8 * (half + j)); ipa_ld(b0, rv + 8 * j); ipa_ld(b1, rv + 8 * (half + j));
    cl = scm_add(cl, scm_mul(a0, b1));
    cr = scm_add(cr, scm_mul(a1, b0));
  }
#pragma unroll 1
  for (int d = 32; d >= 1; d >>= 1) {
    scm o, q;
#pragma unroll
    for (int k = 0; k < 8; ++k) { o.v[k] = (uint32_t)__shfl_down((int)cl.v[k], d); q.v[k] = (uint32_t)__shfl_down((int)cr.v[k], d); }
    cl = scm_add(cl, o);
    cr = scm_add(cr, q);
  }
  if ((t & 63) == 0) {
    for (int k = 0; k < 8; ++k) { red[0][t >> 6][k] = cl.v[k]; red[1][t >> 6][k] = cr.v[k]; }
  }
  __syncthreads();
  if (t == 0) {
    scm tl = scm_zero(), tr = scm_zero(), w;
    for (uint32_t wv = 0; wv < (nt >> 6); ++wv) {
      scm x, y;
      for (int k = 0; k < 8; ++k) { x.v[k] = red[0][wv][k]; y.v[k] = red[1][wv][k]; }
      tl = scm_add(tl, x);
      tr = scm_add(tr, y);
    }
    ipa_ld(w, W + (uint64_t)p * 8);            // plain w: Montgomery x plain = plain
    ipa_st(sL + 8 * pn, scm_mul(tl, w)); iL[pn] = 0;
    ipa_st(sR + 8 * pn, scm_mul(tr, w)); iR[pn] = 0;
  }
}

}  // namespace zk
