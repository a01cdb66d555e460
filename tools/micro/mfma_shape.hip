// Micro-check of MI355X_MICROARCH.md 'DVFS give-back' item 7 on this box: bf16 MFMA throughput on RANDOM operands,
// 32x32x16 against 16x16x32, same 64 x 128 output tile per wave (128 accumulator registers), operands re-read from LDS
// every step, 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const bf16x8 *__restrict__ src, float *__restrict__ out, int iters) {
  __shared__ bf16x8 lds[4096];                                   // 64 KB of operands
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[(blockIdx.x & 7) * 4096 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (SHAPE == 32) {
    f32x16 acc[2][4] = {};
    for (int it = 0; it < iters; it++) {
      const int base = ((it * 8 + wave) * 37) & 4095;
      bf16x8 a[2], b[4];
      for (int r = 0; r < 2; r++) a[r] = lds[(base + r * 64 + lane) & 4095];
      for (int c = 0; c < 4; c++) b[c] = lds[(base + 128 + c * 64 + lane) & 4095];
      for (int r = 0; r < 2; r++)
        for (int c = 0; c < 4; c++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], b[c], acc[r][c], 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 2; r++) for (int c = 0; c < 4; c++) for (int e = 0; e < 16; e++) s += acc[r][c][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    f32x4 acc[4][8] = {};
    for (int it = 0; it < iters; it++) {                       // one iteration = K 32: same flops as TWO iterations of the other shape
      const int base = ((it * 8 + wave) * 37) & 4095;
      bf16x8 a[4], b[8];
      for (int r = 0; r < 4; r++) a[r] = lds[(base + r * 64 + lane) & 4095];
      for (int c = 0; c < 8; c++) b[c] = lds[(base + 256 + c * 64 + lane) & 4095];
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 8; c++) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], b[c], acc[r][c], 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 8; c++) for (int e = 0; e < 4; e++) s += acc[r][c][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

int main() {
  const int nblk = 256 * 8;
  std::vector<unsigned short> h(8 * 4096 * 8);
  srand(1);
  for (auto &v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  bf16x8 *src; float *out;
  hipMalloc(&src, h.size() * 2); hipMalloc(&out, nblk * 512 * 4);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rnd = 0; rnd < 3; rnd++) {
    for (int shape = 0; shape < 2; shape++) {
      const int iters32 = 4096;                                 // K = 16 per iteration -> per wave 8 MFMAs x 32 KFLOP
      float ms;
      if (shape == 0) { k<32><<<nblk, 512>>>(src, out, 64); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 4; i++) k<32><<<nblk, 512>>>(src, out, iters32); hipEventRecord(e1); }
      else { k<16><<<nblk, 512>>>(src, out, 32); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 4; i++) k<16><<<nblk, 512>>>(src, out, iters32 / 2); hipEventRecord(e1); }
      hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      const double flop = 4.0 * nblk * 8.0 * iters32 * 8.0 * 32768.0;
      printf("round %d  %s: %8.3f ms  %7.1f TFLOP/s\n", rnd, shape == 0 ? "32x32x16" : "16x16x32", ms, flop / ms / 1e9);
    }
  }
  return 0;
}
