// Microbenchmark: does v_mfma_f32_32x32x2_f32 (dependent chain) run concurrently with another wave's VALU work
// on the same SIMD?  Block = 512 threads = 8 waves = 2 per SIMD.  mode 0: all 8 waves MFMA; 1: all VALU;
// 2: waves 0-3 MFMA, waves 4-7 VALU (one of each per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(512) k(float* out, int iters, int mode, int nm, int nv) {
  const int w = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0) || (mode == 2 && w < 4);
  float r = 0.f;
  if (do_mfma) {
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)(threadIdx.x & 7);
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (j < nm) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
    for (int i = 0; i < 16; i++) r += acc[i];
  } else {
    f2 acc[8];
    f2 x[8];
    for (int i = 0; i < 8; i++) {
      acc[i] = f2{(float)threadIdx.x, 1.0f};
      x[i] = f2{(float)i, 2.0f};
    }
    f2 hh = f2{0.5f, 0.5f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int rr = 0; rr < 8; rr++) {
        if (rr < nv) {
#pragma unroll
          for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(hh), "v"(x[(i + rr) & 7]));
        }
      }
    }
    for (int i = 0; i < 8; i++) r += acc[i].x + acc[i].y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  float* d;
  const int blocks = 256 * 4, iters = 2000;
  hipMalloc(&d, blocks * 512 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[3] = {"8 waves MFMA (8 per iter)", "8 waves VALU (64 pk_fma per iter)", "4 MFMA + 4 VALU waves"};
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, iters, mode, 8, 8);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%-36s %.3f ms\n", names[mode], ms);
    }
  }
  // per-instruction cost: blocks/256 = 4 blocks per CU sequentially (1 resident at 512 thr? up to 4), report cycles/MFMA/SIMD
  return 0;
}
