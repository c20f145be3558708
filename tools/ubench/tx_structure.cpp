// Host-side microbenchmark (no GPU): the payment VM split into its stack machine (tx_structure: full plan / keys-only plan) and
// the keys-only pass with its hashing.  build: g++ -O2 -std=c++17 -pthread -I zkvm_amd/csrc -o /tmp/tx_structure tools/ubench/tx_structure.cpp
#include "zkvm_tx.hpp"
#include <chrono>
#include <cstdio>
#include <fstream>
using namespace zk; using namespace zk::zkvm;
int main() {
  std::ifstream f1("tests/golden/tx_2x2_1024_wrappers.bin", std::ios::binary), f2("tests/golden/cloak_2x2_1024.bin", std::ios::binary);
  std::vector<uint8_t> w((std::istreambuf_iterator<char>(f1)), {}), c((std::istreambuf_iterator<char>(f2)), {});
  uint32_t count = rd32(&c[8]), n_in = rd32(&c[12]), n_out = rd32(&c[16]), plen = rd32(&c[20]);
  size_t wcom = 64 * (n_in + n_out), rec = wcom + plen, pos = 24;
  std::vector<std::vector<uint8_t>> txs;
  for (uint32_t i = 0; i < count; ++i) {
    uint32_t n = rd32(&w[pos]);
    std::vector<uint8_t> t(w.begin() + pos + 4, w.begin() + pos + 4 + n);
    uint8_t l[4] = {(uint8_t)plen, (uint8_t)(plen >> 8), (uint8_t)(plen >> 16), (uint8_t)(plen >> 24)};
    t.insert(t.end(), l, l + 4);
    t.insert(t.end(), c.begin() + 24 + rec * i + wcom, c.begin() + 24 + rec * (i + 1));
    txs.push_back(t); pos += 4 + n;
  }
  TxStatement st; TxPlan P; TxSlots out;
  for (int mode = 0; mode < 3; ++mode) {
    auto t0 = std::chrono::steady_clock::now();
    size_t jobs = 0;
    for (int rep = 0; rep < 20; ++rep)
      for (auto& t : txs) {
        P.only = mode == 1 ? P_MUSIG : 0xff;
        tx_structure(t.data(), t.size(), st, P, out);
        jobs += P.jobs.size();
        if (mode == 2) { const uint8_t* p = t.data(); size_t l = t.size(); TxStatement s2; tx_prepare_many(&p, &l, &s2, 1, true, P_MUSIG); }
      }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("mode %d: %.3f us per tx (%zu jobs per tx)\n", mode, dt / (20 * txs.size()) * 1e6, jobs / (20 * txs.size()));
  }
}
