// Attainable rate of the fp32 matrix instructions on this board: v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32, operands held in
// registers (no LDS, no memory), 8 independent accumulator tiles per wave, 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_rate.hip -o /tmp/mfma_f32_rate && /tmp/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float *__restrict__ out, int iters, float seed) {
  const float a0 = seed + threadIdx.x * 0.001f, b0 = seed * 0.5f - threadIdx.x * 0.002f;
  float s = 0;
  if constexpr (SHAPE == 32) {
    f32x16 acc[8] = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 8; u++) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0 + u, b0 - u, acc[u], 0, 0, 0);
    }
    for (int u = 0; u < 8; u++) for (int e = 0; e < 16; e++) s += acc[u][e];
  } else {
    f32x4 acc[8] = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 8; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0 + u, b0 - u, acc[u], 0, 0, 0);
    }
    for (int u = 0; u < 8; u++) for (int e = 0; e < 4; e++) s += acc[u][e];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float *out; hipMalloc(&out, 256 * 8 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 4; wps *= 2)
    for (int shape = 0; shape < 2; shape++) {
      const int threads = 256, blocks = 256 * wps, iters = 1 << 15;         // 4 waves per block = one per SIMD; wps blocks per CU
      float ms;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < 4; i++) { if (shape == 0) k<32><<<blocks, threads>>>(out, iters, 1.25f); else k<16><<<blocks, threads>>>(out, iters * 2, 1.25f); }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      const double flop = 4.0 * blocks * 4.0 * iters * 8.0 * 4096.0;
      printf("%s, %d wave(s) per SIMD: %7.1f TFLOP/s (%.3f of 157.3)\n", shape == 0 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", wps, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
    }
  return 0;
}
