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
  bool busy;
};

// Per-kernel HIP-event timing.  Spans may sit on different streams (the transmit and receive sides of a handle run
// on their own): collect() takes the spans whose end event has completed and leaves the others for a later call.
struct Profiler {
  bool on = false;
  double total_ms[OFDM_K_COUNT] = {0};
  uint64_t launches[OFDM_K_COUNT] = {0};
  std::vector<ProfSpan> pool;  // events, reused call after call
  int open = -1;               // span begun and not yet ended

  void begin(int k, hipStream_t s) {
    if (!on) return;
    int idx = -1;
    for (size_t i = 0; i < pool.size(); i++)
      if (!pool[i].busy) {
        idx = (int)i;
        break;
      }
    if (idx < 0) {
      ProfSpan sp;
      sp.kernel = k;
      sp.busy = false;
      (void)hipEventCreate(&sp.a);
      (void)hipEventCreate(&sp.b);
      pool.push_back(sp);
      idx = (int)pool.size() - 1;
    }
    pool[idx].kernel = k;
    pool[idx].busy = true;
    (void)hipEventRecord(pool[idx].a, s);
    open = idx;
  }
  void end(hipStream_t s) {
    if (!on || open < 0) return;
    (void)hipEventRecord(pool[open].b, s);
    open = -1;
  }
  void collect() {
    for (size_t i = 0; i < pool.size(); i++) {
      if (!pool[i].busy || (int)i == open) continue;
      if (hipEventQuery(pool[i].b) != hipSuccess) continue;  // still running (another stream): next time
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, pool[i].a, pool[i].b) == hipSuccess) {
        total_ms[pool[i].kernel] += (double)ms;
        launches[pool[i].kernel] += 1;
      }
      pool[i].busy = false;
    }
  }
  void reset() {
    for (int i = 0; i < OFDM_K_COUNT; i++) {
      total_ms[i] = 0;
      launches[i] = 0;
    }
    for (auto& sp : pool) sp.busy = false;
    open = -1;
  }
  void destroy() {
    for (auto& sp : pool) {
      (void)hipEventDestroy(sp.a);
      (void)hipEventDestroy(sp.b);
    }
    pool.clear();
    open = -1;
  }
};
