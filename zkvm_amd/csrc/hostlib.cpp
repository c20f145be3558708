// hostlib.cpp -- the host-only part of the product (scalar field, Merlin transcript,
// R1CS verifier scalar preparation) as a plain C++ shared library, so that the CPU test
// tier can exercise it without a GPU.  libzkgpu.so compiles the very same headers.
#include "r1cs_verifier.hpp"

#include <cstring>

using namespace zk;

extern "C" {

// op: 0 add, 1 sub, 2 mul, 3 neg(a), 4 invert(a), 5 reduce only; inputs are 64-byte wide values
int zkhost_scalar_op(int op, const uint8_t a64[64], const uint8_t b64[64], uint8_t out[32]) {
  const Scalar a = Scalar::from_wide(a64), b = Scalar::from_wide(b64);
  Scalar r;
  switch (op) {
    case 0: r = a + b; break;
    case 1: r = a - b; break;
    case 2: r = a * b; break;
    case 3: r = -a; break;
    case 4: r = a.invert(); break;
    case 5: r = a; break;
    default: return -1;
  }
  r.to_bytes(out);
  return 0;
}

int zkhost_scalar_is_canonical(const uint8_t b[32]) {
  Scalar s;
  return Scalar::from_canonical(b, s) ? 1 : 0;
}

// Transcript::new(label); n x append_message(labels[i], msgs[i]); challenge_bytes(ch_label, out)
int zkhost_merlin(const char* label, int n, const char* const* labels, const uint8_t* const* msgs, const size_t* lens,
                  const char* ch_label, uint8_t* out, size_t out_len) {
  Transcript t(label);
  for (int i = 0; i < n; ++i) t.append_message(labels[i], msgs[i], lens[i]);
  t.challenge_bytes(ch_label, out, out_len);
  return 0;
}

// cloak::prepare_tx; buffers sized by the caller: dyn 32 * 64 each, static 32 * (2 + 2 * cap), index 4 * (2 + 2 * cap)
int zkhost_cloak_prepare(const uint8_t* commitments, size_t n_in, size_t n_out, const uint8_t* proof, size_t proof_len,
                         const uint8_t r64[64], size_t gens_capacity, uint8_t* dyn_scalars, uint8_t* dyn_points,
                         size_t* n_dyn, uint8_t* static_scalars, uint32_t* static_index, size_t* n_static,
                         size_t* padded_n) {
  VerifierMsm m;
  if (!cloak::prepare_tx(commitments, n_in, n_out, proof, proof_len, Scalar::from_wide(r64), gens_capacity, m)) return 1;
  *n_dyn = m.dyn_scalars.size() / 32;
  *n_static = m.static_scalars.size() / 32;
  *padded_n = m.padded_n;
  std::memcpy(dyn_scalars, m.dyn_scalars.data(), m.dyn_scalars.size());
  std::memcpy(dyn_points, m.dyn_points.data(), m.dyn_points.size());
  std::memcpy(static_scalars, m.static_scalars.data(), m.static_scalars.size());
  std::memcpy(static_index, m.static_index.data(), m.static_index.size() * 4);
  return 0;
}

}  // extern "C"
