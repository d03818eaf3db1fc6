// host_pool.hpp -- a few persistent host threads for the host-side builders (balltree.cpp).
//
// Creating a thread costs more than building a 2048-point tree (measured: 0.1-1.5 ms per create+join under the
// container runtimes this library runs in), so the workers are started once, on first use, and sleep on a condition
// variable in between.  A task that no worker has picked up by the time its owner needs the result is run by the owner
// itself (`claim`), so nested fork/join cannot deadlock, a process that fork()ed away from its workers still makes
// progress, and a machine with one core simply builds serially.
// (Round 5 tried workers that poll ~100 us before they sleep and owners that run other queued tasks while they wait: on the
// 256-thread GPU box a 6 x 2048 tree took 0.25 ms against 0.21 ms with the sleeping workers below, and one call in fifty
// 1.7 ms -- dropped, profiles/r05_experiments.md section 7.  KDEHIP_HOST_THREADS caps the workers, 0 = serial.)
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>

namespace kdehip {

class HostPool {
 public:
  struct Task {
    std::function<void()> fn;
    std::atomic<int> state{0};  // 0 queued, 1 claimed (running), 2 done
    std::exception_ptr error;   // what fn threw, if anything (written before state 2; rethrown by join)
  };
  using Ticket = std::shared_ptr<Task>;

  static HostPool &get() {
    static HostPool *pool = new HostPool();  // never destroyed: the workers may outlive static destruction
    return *pool;
  }
  int workers() const { return nworkers_.load(std::memory_order_relaxed); }

  Ticket submit(std::function<void()> fn) {
    Ticket t = std::make_shared<Task>();
    t->fn = std::move(fn);
    if (workers() > 0) {
      {
        std::lock_guard<std::mutex> lock(mu_);
        queue_.push_back(t);
      }
      cv_.notify_one();
    }
    return t;
  }

  // returns when the task has run -- on a worker, or here if none has started it yet; an exception the task threw
  // (an allocation failure, say) is thrown again here, in the thread that waits for the result
  void join(const Ticket &t) {
    if (!run_if_unclaimed(*t)) {
      bool done = false;
      for (int spin = 0; spin < 4096 && !done; ++spin) {
        done = t->state.load(std::memory_order_acquire) == 2;
        if (!done) cpu_relax();
      }
      if (!done) {
        std::unique_lock<std::mutex> lock(done_mu_);
        done_cv_.wait(lock, [&] { return t->state.load(std::memory_order_acquire) == 2; });
      }
    }
    if (t->error) std::rethrow_exception(t->error);
  }

 private:
  HostPool() {
    const unsigned hw = std::thread::hardware_concurrency();
    int want = hw > 1 ? static_cast<int>(hw > kMaxWorkers ? kMaxWorkers : hw - 1) : 0;
    if (const char *e = std::getenv("KDEHIP_HOST_THREADS")) {  // worker threads beside the caller (0 = build serially)
      const int n = std::atoi(e);
      if (n >= 0 && n < want) want = n;
    }
    for (int i = 0; i < want; ++i) {
      try {
        std::thread([this] { work(); }).detach();
        nworkers_.fetch_add(1, std::memory_order_relaxed);
      } catch (const std::system_error &) {
        break;
      }
    }
    // a fork()ed child has none of the workers (and must not touch locks other threads held): its owners run their own tasks
    self_.store(this, std::memory_order_release);
    pthread_atfork(nullptr, nullptr, [] {
      if (HostPool *p = self_.load(std::memory_order_acquire)) p->nworkers_.store(0, std::memory_order_relaxed);
    });
  }
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  static bool run_if_unclaimed(Task &t) {
    int expect = 0;
    if (!t.state.compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) return false;
    try {
      t.fn();
    } catch (...) {
      t.error = std::current_exception();
    }
    t.state.store(2, std::memory_order_release);
    return true;
  }
  void work() {
    for (;;) {
      Ticket t;
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&] { return !queue_.empty(); });
        t = std::move(queue_.front());
        queue_.pop_front();
      }
      if (run_if_unclaimed(*t)) {
        std::lock_guard<std::mutex> lock(done_mu_);  // (pairs with the predicate check in join)
        done_cv_.notify_all();
      }
    }
  }

  static constexpr unsigned kMaxWorkers = 15;  // (measured on the 256-thread MI355X host: 7 -> 15 gains 10 % at 1e5 points)
  inline static std::atomic<HostPool *> self_{nullptr};
  std::atomic<int> nworkers_{0};
  std::mutex mu_, done_mu_;
  std::condition_variable cv_, done_cv_;
  std::deque<Ticket> queue_;
};

// A group of tasks that refer to the caller's frame: wait() rethrows the first exception a task threw; leaving the frame
// any other way (an exception of the caller's own work) still waits for every task first.
class TaskGroup {
 public:
  explicit TaskGroup(HostPool &pool) : pool_(pool) {}
  TaskGroup(const TaskGroup &) = delete;
  TaskGroup &operator=(const TaskGroup &) = delete;
  ~TaskGroup() {
    for (auto &t : tickets_) {
      try { pool_.join(t); } catch (...) {}
    }
  }
  void run(std::function<void()> fn) { tickets_.push_back(pool_.submit(std::move(fn))); }
  void wait() {
    std::exception_ptr first;
    for (auto &t : tickets_) {
      try { pool_.join(t); } catch (...) { if (!first) first = std::current_exception(); }
    }
    tickets_.clear();
    if (first) std::rethrow_exception(first);
  }

 private:
  HostPool &pool_;
  std::vector<HostPool::Ticket> tickets_;
};

}  // namespace kdehip
