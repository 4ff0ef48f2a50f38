// Short-lived worker threads of a host-side pass (bounding boxes, staging copies, per-scan or per-pair loops).  Host only.
#pragma once
#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

// Runs work() on up to n_threads threads, the calling one included; work() pulls its items from a counter the caller shares with it, so a
// thread that could not be created (std::system_error) just leaves its share to the others.  An exception inside a worker is caught there
// and rethrown on the calling thread after every thread has been joined — it never reaches a thread's top frame (std::terminate), and a
// partially built pool is never destroyed joinable.
template <class Work>
inline void pvlm_run_workers(size_t n_threads, Work&& work) {
  std::exception_ptr failure;
  std::mutex lock;
  auto guarded = [&]() {
    try { work(); } catch (...) { std::lock_guard<std::mutex> g(lock); if (!failure) failure = std::current_exception(); }
  };
  std::vector<std::thread> pool;
  try {
    pool.reserve(n_threads);
    for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(guarded);
  } catch (...) {}
  guarded();
  for (std::thread& t : pool) t.join();
  if (failure) std::rethrow_exception(failure);
}

// upper limit of the worker threads of one host-side pass: 16 (a shared host rarely gives a short-lived pass more), PVLM_HOST_THREADS overrides
inline size_t pvlm_thread_cap() {
  static const size_t cap = [] { const char* e = std::getenv("PVLM_HOST_THREADS"); const long v = e ? std::atol(e) : 0; return v > 0 ? (size_t)v : (size_t)16; }();
  return cap;
}
