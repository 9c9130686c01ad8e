// host_util.h -- host-side plumbing of the engine: error handling, grow-only device
// buffers, pinned staging, HIP-event profiling.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/ofdm_hip.h"

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

struct ProfSpan {
  int kernel;
  hipEvent_t a, b;
};

struct Profiler {
  bool on = false;
  double total_ms[OFDM_K_COUNT] = {0};
  uint64_t launches[OFDM_K_COUNT] = {0};
  std::vector<ProfSpan> pool;  // events, reused call after call
  size_t used = 0;

  void begin(int k, hipStream_t s) {
    if (!on) return;
    if (used == pool.size()) {
      ProfSpan sp;
      sp.kernel = k;
      (void)hipEventCreate(&sp.a);
      (void)hipEventCreate(&sp.b);
      pool.push_back(sp);
    }
    pool[used].kernel = k;
    (void)hipEventRecord(pool[used].a, s);
  }
  void end(hipStream_t s) {
    if (!on) return;
    (void)hipEventRecord(pool[used].b, s);
    used++;
  }
  // call after the stream has been synchronised
  void collect() {
    for (size_t i = 0; i < used; i++) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, pool[i].a, pool[i].b) == hipSuccess) {
        total_ms[pool[i].kernel] += (double)ms;
        launches[pool[i].kernel] += 1;
      }
    }
    used = 0;
  }
  void reset() {
    for (int i = 0; i < OFDM_K_COUNT; i++) {
      total_ms[i] = 0;
      launches[i] = 0;
    }
    used = 0;
  }
  void destroy() {
    for (auto& sp : pool) {
      (void)hipEventDestroy(sp.a);
      (void)hipEventDestroy(sp.b);
    }
    pool.clear();
    used = 0;
  }
};
