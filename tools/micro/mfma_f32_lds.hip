// v_mfma_f32_32x32x2_f32 fed from LDS (the headline block's and the conv family's situation): naive order (read, wait, 8 MFMAs) against
// software-pipelined reads (next step's operands requested before this step's MFMAs), 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_lds.hip -o /tmp/mfma_f32_lds && /tmp/mfma_f32_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool PIPE, int BLK>
__global__ __launch_bounds__(512, BLK) void k(const float *__restrict__ src, float *__restrict__ out, int iters) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = src[(blockIdx.x & 7) * 16384 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2][4] = {};
  float a[2][2], b[2][4];
  auto rd = [&](int it, int set) {
    const int base = ((it * 8 + wave) * 389) & 16383;
    for (int r = 0; r < 2; r++) a[set][r] = lds[(base + r * 64 + lane) & 16383];
    for (int c = 0; c < 4; c++) b[set][c] = lds[(base + 128 + c * 64 + lane) & 16383];
  };
  if (PIPE) rd(0, 0);
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (PIPE) rd(it + h + 1, (h + 1) & 1); else rd(it + h, h);
      for (int r = 0; r < 2; r++)
        for (int c = 0; c < 4; c++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[h][r], b[h][c], acc[r][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int r = 0; r < 2; r++) for (int c = 0; c < 4; c++) for (int e = 0; e < 16; e++) s += acc[r][c][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// 4-byte reads, NB k-steps' operands requested together, then their 8 NB MFMAs back to back: is it the read width or the number of
// read -> wait -> MFMA turnarounds that costs?
template <int NB>
__global__ __launch_bounds__(512, 2) void kb(const float *__restrict__ src, float *__restrict__ out, int iters) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = src[(blockIdx.x & 7) * 16384 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2][4] = {};
  for (int it = 0; it < iters; it += NB) {
    float a[NB][2], b[NB][4];
#pragma unroll
    for (int u = 0; u < NB; u++) {
      const int base = (((it + u) * 8 + wave) * 389) & 16383;
      for (int r = 0; r < 2; r++) a[u][r] = lds[(base + r * 64 + lane) & 16383];
      for (int c = 0; c < 4; c++) b[u][c] = lds[(base + 128 + c * 64 + lane) & 16383];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NB; u++)
      for (int r = 0; r < 2; r++)
        for (int c = 0; c < 4; c++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][r], b[u][c], acc[r][c], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int r = 0; r < 2; r++) for (int c = 0; c < 4; c++) for (int e = 0; e < 16; e++) s += acc[r][c][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// the same with 16-byte LDS reads: a lane's operand for FOUR consecutive k-steps in one ds_read_b128 (a [row][k] image, k contiguous)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int BLK>
__global__ __launch_bounds__(512, BLK) void k4(const float *__restrict__ src, float *__restrict__ out, int iters) {
  __shared__ f32x4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = reinterpret_cast<const f32x4 *>(src)[(blockIdx.x & 7) * 4096 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2][4] = {};
  for (int it = 0; it < iters; it += 4) {
    const int base = ((it * 2 + wave) * 389) & 4095;
    f32x4 a[2], b[4];
    for (int r = 0; r < 2; r++) a[r] = lds[(base + r * 64 + lane) & 4095];
    for (int c = 0; c < 4; c++) b[c] = lds[(base + 128 + c * 64 + lane) & 4095];
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
      for (int r = 0; r < 2; r++)
        for (int c = 0; c < 4; c++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r][kk], b[c][kk], acc[r][c], 0, 0, 0);
  }
  float s = 0;
  for (int r = 0; r < 2; r++) for (int c = 0; c < 4; c++) for (int e = 0; e < 16; e++) s += acc[r][c][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  std::vector<float> h(8 * 16384);
  srand(1);
  for (auto &v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
  float *src, *out; hipMalloc(&src, h.size() * 4); hipMalloc(&out, 256 * 2 * 512 * 4);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 1 << 14;
  for (int v = 0; v < 4; v++) {
    const int blk = (v & 1) ? 2 : 1, nblk = 256 * blk;
    float ms;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      for (int i = 0; i < 4; i++) {
        if (v == 0) k<false, 1><<<nblk, 512>>>(src, out, iters);
        else if (v == 1) k<false, 2><<<nblk, 512>>>(src, out, iters);
        else if (v == 2) k<true, 1><<<nblk, 512>>>(src, out, iters);
        else k<true, 2><<<nblk, 512>>>(src, out, iters);
      }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 4.0 * nblk * 8.0 * iters * 8.0 * 4096.0;
    printf("%s, %d waves per SIMD: %7.1f TFLOP/s (%.3f of 157.3)\n", v < 2 ? "read -> wait -> 8 MFMAs      " : "reads one step ahead of MFMAs", 2 * blk, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
  }
  for (int nb = 1; nb <= 4; nb *= 2) {
    const int nblk = 512;
    float ms;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      for (int i = 0; i < 4; i++) { if (nb == 1) kb<1><<<nblk, 512>>>(src, out, iters); else if (nb == 2) kb<2><<<nblk, 512>>>(src, out, iters); else kb<4><<<nblk, 512>>>(src, out, iters); }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 4.0 * nblk * 8.0 * iters * 8.0 * 4096.0;
    printf("4-byte reads of %d k-step(s) together, then %2d MFMAs, 4 waves per SIMD: %7.1f TFLOP/s (%.3f of 157.3)\n", nb, 8 * nb, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
  }
  for (int blk = 1; blk <= 2; blk++) {
    const int nblk = 256 * blk;
    float ms;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      for (int i = 0; i < 4; i++) { if (blk == 1) k4<1><<<nblk, 512>>>(src, out, iters); else k4<2><<<nblk, 512>>>(src, out, iters); }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 4.0 * nblk * 8.0 * iters * 8.0 * 4096.0;
    printf("16-byte reads, four k-steps each , %d waves per SIMD: %7.1f TFLOP/s (%.3f of 157.3)\n", 2 * blk, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
  }
  return 0;
}
