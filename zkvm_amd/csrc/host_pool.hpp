// Persistent host worker threads behind host_parallel (zkgpu.hip).
//
// The host stages on the product path (the transaction VM and ids of zkgpu_tx_verify_batch, the witness rows of the
// device prover) are short -- a millisecond or two per call -- and creating and joining a dozen threads per call
// costs a good part of that.  The workers are created on first use, sleep on a condition variable between calls and
// are joined when the library is unloaded.  One call uses the pool at a time; a second caller arriving meanwhile (two
// prover calls in flight on two contexts) waits for its turn; a process forked after the pool was made runs with
// threads of its own.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <sched.h>
#include <unistd.h>

namespace zk {

// How many threads a host stage may keep busy when the caller does not say: the CPUs this process may run on (its affinity
// mask) and, inside a container, the CPU-time quota of its control group (cpu.max of cgroup v2, cpu.cfs_quota_us of v1) --
// not the machine's core count: threads beyond the quota do not run side by side, they use up the group's time slice
// and the whole process, the thread that feeds the GPU included, is paused until the next period.
inline int usable_cpus() {
  static const int n = [] {
    int cpus = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0 && a < cpus) cpus = a; }
    auto quota = [](const char* path, bool v2) -> double {
      FILE* f = fopen(path, "r");
      if (!f) return 0;
      char buf[64] = {0};
      const size_t got = fread(buf, 1, sizeof buf - 1, f);
      fclose(f);
      if (!got) return 0;
      if (v2) {
        long long q = 0, per = 0;
        if (sscanf(buf, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) return (double)q / (double)per;
        return 0;                                                       // "max ..."
      }
      return atof(buf);
    };
    double q = quota("/sys/fs/cgroup/cpu.max", true);
    if (q <= 0) {
      const double us = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false), per = quota("/sys/fs/cgroup/cpu/cpu.cfs_period_us", false);
      if (us > 0 && per > 0) q = us / per;
    }
    if (q > 0 && q < cpus) cpus = (int)(q + 0.5);
    return cpus > 0 ? cpus : 1;
  }();
  return n;
}

class HostPool {
 public:
  static HostPool& get() {
    static HostPool p;
    return p;
  }
  static constexpr int MAX_WORKERS = 31;

  // f(i) for i < n on the caller and up to nt - 1 workers; false when the pool cannot take the call (forked process)
  bool run(size_t n, int nt, const std::function<void(size_t)>& f) {
    if (getpid() != owner_) return false;
    std::unique_lock<std::mutex> busy(run_mu_);      // a second caller waits its turn: the stages are short, and a dozen
                                                     // more threads beside a full pool would only take its cores away
    {
      std::lock_guard<std::mutex> lk(mu_);
      grow(nt - 1);
      job_ = &f;
      n_ = n;
      grain_ = std::max<size_t>(1, n / ((size_t)nt * 8));
      next_.store(0, std::memory_order_relaxed);
      want_ = std::min<int>(nt - 1, (int)th_.size());
      pending_ = want_;
      ++gen_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
    return true;
  }

 private:
  HostPool() : owner_(getpid()) {}
  ~HostPool() {
    if (getpid() != owner_) return;   // forked copy: the threads never existed here, their handles are left alone
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
    delete &th_;
  }
  void grow(int workers) {   // mu_ held
    workers = std::min(workers, MAX_WORKERS);
    while ((int)th_.size() < workers) {
      const int id = (int)th_.size();
      th_.emplace_back([this, id, seen = gen_] { worker(id, seen); });
    }
  }
  void work() {
    for (;;) {
      const size_t a = next_.fetch_add(grain_, std::memory_order_relaxed);
      if (a >= n_) return;
      const size_t b = std::min(n_, a + grain_);
      for (size_t i = a; i < b; ++i) (*job_)(i);
    }
  }
  void worker(int id, uint64_t seen) {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
      if (stop_) return;
      seen = gen_;
      if (id >= want_) continue;
      lk.unlock();
      work();
      lk.lock();
      if (--pending_ == 0) done_.notify_one();
    }
  }

  const pid_t owner_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread>& th_ = *new std::vector<std::thread>();
  const std::function<void(size_t)>* job_ = nullptr;
  size_t n_ = 0, grain_ = 1;
  std::atomic<size_t> next_{0};
  int want_ = 0, pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

// f(i) for i < n on up to `threads` host threads (0: usable_cpus(): affinity mask and control-group quota)
inline void host_parallel(size_t n, int threads, const std::function<void(size_t)>& f) {
  const int nt = (int)std::max<size_t>(1, std::min<size_t>(n / 16 + 1, (size_t)std::min<int>(threads > 0 ? threads : usable_cpus(), 256)));
  if (nt <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
  if (HostPool::get().run(n, std::min(nt, HostPool::MAX_WORKERS + 1), f)) return;
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { for (size_t i = (size_t)t; i < n; i += (size_t)nt) f(i); });
  for (auto& t : th) t.join();
}

}  // namespace zk
