// Host-side microbenchmark (no GPU): the payment VM (zkvm_tx.hpp, tx_prepare) over the committed transactions on 1 .. 32
// threads, through the library's worker pool and through plain threads; and the pool's wake-up cost.
// build: g++ -O2 -std=c++17 -pthread -I zkvm_amd/csrc -o /tmp/vm_threads tools/ubench/vm_threads.cpp ; run from the repo root
#include "zkvm_tx.hpp"
#include "host_pool.hpp"
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include <thread>
using namespace zk; using namespace zk::zkvm;
int main(int argc, char** argv) {
  // fixture: wrappers + proofs
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
  const size_t N = 8192;
  for (int nt : {1, 2, 4, 8, 12, 16, 24, 32}) {
    std::vector<TxStatement> st(N);
    auto t0 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < 3; ++rep) {
      // groups of eight transactions: hashed in lockstep on AVX-512 (argv[1] = "scalar": one at a time)
      const bool x8 = !(argc > 1 && std::string(argv[1]) == "scalar");
      std::function<void(size_t)> f = [&](size_t g) {
        const uint8_t* p[8]; size_t l[8];
        for (int q = 0; q < 8; ++q) { p[q] = txs[(8 * g + q) % count].data(); l[q] = txs[(8 * g + q) % count].size(); }
        tx_prepare_many(p, l, &st[8 * g], 8, x8);
      };
      HostPool::get().run(N / 8, nt, f);
    }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 3;
    int ok = 0; for (auto& s : st) ok += s.status == TX_OK;
    printf("%2d threads: %.2f ms per 8192 (%.2f us CPU per tx), %d ok\n", nt, dt * 1e3, dt * nt / N * 1e6, ok);
    // the pool's fixed cost: a job of nothing
    std::function<void(size_t)> nop = [](size_t) {};
    auto t1 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < 200; ++rep) HostPool::get().run(64, nt, nop);
    printf("           empty pool job: %.1f us\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() / 200 * 1e6);
  }
}
