// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (decides how the
// channel-filter inner loop should be written).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float h) {
  f2 a[8];
  f2 x[8];
  for (int i = 0; i < 8; i++) {
    a[i] = f2{(float)threadIdx.x, 1.0f};
    x[i] = f2{(float)i, 2.0f};
  }
  f2 hh = f2{h, h};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (MODE == 0) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(hh), "v"(x[(i + r) & 7]));
        } else {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(h), "v"(x[(i + r) & 7].x));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(h), "v"(x[(i + r) & 7].y));
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* d;
  const int blocks = 256 * 8, iters = 2000;
  hipMalloc(&d, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      if (mode == 0)
        hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f);
      else
        hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      // complex FMAs (2 flops x 2 lanes-of-pair) per launch
      double cfma = (double)blocks * 256 * iters * 64;
      printf("%s: %.3f ms  %.2f T complex-fma/s  (%.1f TFLOP/s)\n", mode == 0 ? "v_pk_fma_f32" : "2 x v_fma_f32", ms,
             cfma / ms / 1e9, cfma * 4 / ms / 1e9);
    }
  }
  return 0;
}
