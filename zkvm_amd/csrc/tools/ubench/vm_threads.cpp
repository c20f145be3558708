#include "/root/repo/zkvm_amd/csrc/zkvm_tx.hpp"
#include "/root/repo/zkvm_amd/csrc/host_pool.hpp"
#include <chrono>
#include <cstdio>
#include <fstream>
#include <thread>
using namespace zk; using namespace zk::zkvm;
int main(int argc, char** argv) {
  // fixture: wrappers + proofs
  std::ifstream f1("/root/repo/tests/golden/tx_2x2_1024_wrappers.bin", std::ios::binary), f2("/root/repo/tests/golden/cloak_2x2_1024.bin", std::ios::binary);
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
  for (int nt : {1, 2, 4, 8, 16}) {
    std::vector<TxStatement> st(N);
    auto t0 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < 3; ++rep) {
      std::function<void(size_t)> f = [&](size_t i) { st[i] = tx_prepare(txs[i % count].data(), txs[i % count].size()); };
      if (argc > 1) { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { for (size_t i = t; i < N; i += nt) f(i); }); for (auto& x : th) x.join(); }
      else HostPool::get().run(N, nt, f);
    }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 3;
    int ok = 0; for (auto& s : st) ok += s.status == TX_OK;
    printf("%2d threads: %.2f ms per 8192 (%.2f us CPU per tx), %d ok\n", nt, dt * 1e3, dt * nt / N * 1e6, ok);
  }
}
