// What a wave-wide 16-byte load costs the CU's vector-memory path, by address pattern (data L2-resident): 8 waves per CU, each
// issuing NLOAD independent buffer_load_dwordx4 per iteration; rows of the pattern are 64,000 B apart (one channel row of a
// 16,000-sample clip).   hipcc --offload-arch=gfx950 -O3 tools/micro/load_pattern.hip -o /tmp/lp && /tmp/lp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWB = 64000;
template <int ROWS>                       // ROWS rows x (1024 / ROWS) contiguous bytes per wave-instruction
__global__ __launch_bounds__(512, 2) void k(const float *src, unsigned bytes, float *out, int iters, unsigned long long *cyc) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)bytes, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int SEG = 64 / ROWS;          // lanes per row
  const unsigned voff = (unsigned)((lane / SEG) * ROWB + (lane % SEG) * 16 + wave * 1024 * 4);
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    u32x4 v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)(((it * 8 + e) & 15) * ROWS * ROWB), 0);
#pragma unroll
    for (int e = 0; e < 8; e++) acc ^= v[e];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc[0] == 0x12345678u) out[threadIdx.x] = (float)acc[1];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ROWS> double run(const float *d, unsigned bytes, float *o, unsigned long long *c, int iters) {
  k<ROWS><<<256, 512>>>(d, bytes, o, 10, c);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<ROWS><<<256, 512>>>(d, bytes, o, iters, c);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double loads_per_cu = (double)iters * 8 * 8;                 // wave-instructions per CU
  printf("%2d rows x %4d B: %7.3f ms  -> %6.1f ns per wave-load per CU = %5.1f B/ns/CU (%.2f TB/s chip)\n", ROWS, 1024 / ROWS, ms,
         ms * 1e6 / loads_per_cu, 1024.0 / (ms * 1e6 / loads_per_cu), 256 * 1024.0 / (ms * 1e6 / loads_per_cu) / 1000);
  return ms;
}
int main() {
  const unsigned bytes = 16u * 64 * ROWB + 65536;                    // 16 x ROWS x 64 KB rows fit: 65 MB max, mostly L2/MALL resident rows
  float *d, *o; unsigned long long *c;
  (void)hipMalloc(&d, bytes); (void)hipMemset(d, 0, bytes); (void)hipMalloc(&o, 4096); (void)hipMalloc(&c, 256 * 8);
  for (int rep = 0; rep < 2; rep++) {
    run<1>(d, bytes, o, c, 2000); run<2>(d, bytes, o, c, 2000); run<4>(d, bytes, o, c, 2000); run<8>(d, bytes, o, c, 2000); run<16>(d, bytes, o, c, 2000);
  }
  return 0;
}
