// The bf16 block's mix without the block: per "tile" and CU 4096 v_mfma_f32_32x32x16_bf16 on random operands (134 MFLOP) beside
// 262 KB of fp32 read from HBM and 262 KB written back (streams touched once, like h / skip in and h' / skip out), nothing else --
// no weight stream, no staging, no gate, no transposes.  What tile rate does the 1400 W cap allow THAT?  (rocm-smi polled meanwhile.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_hbm_mix.hip -o /tmp/mfma_hbm_mix -lpthread && /tmp/mfma_hbm_mix [seconds] [workgroups per CU]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one workgroup per CU (512 threads = 8 waves x (64 x 128 accumulators)), persistent over `tiles` tiles; per tile and wave 512 MFMAs
// with one 16-byte load and one 16-byte store per lane every 16 MFMAs (32 of each per tile: 512 threads x 32 x 16 B = 262 KB each way)
template <bool MEM, bool MMA>
__global__ __launch_bounds__(512, 2) void mix_k(const bf16x8 *__restrict__ ops, const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                                float *__restrict__ out, int tiles) {
  __shared__ bf16x8 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = ops[(blockIdx.x & 7) * 4096 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2][4] = {};
  f32x4 carry = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < tiles; t++) {
    const size_t tile = (size_t)t * gridDim.x + blockIdx.x;      // 16384 f32x4 (262 KB) per tile
    const f32x4 *sp = src + tile * 16384 + threadIdx.x;
    f32x4 *dp = dst + tile * 16384 + threadIdx.x;
#pragma unroll 1
    for (int g = 0; g < 32; g++) {
      f32x4 v = carry;
      if (MEM) v = __builtin_nontemporal_load(sp + g * 512);
      if (MMA) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
          const int base = (((t * 64 + g * 2 + s) * 8 + wave) * 37) & 4095;
          bf16x8 a[2], b[4];
          for (int r = 0; r < 2; r++) a[r] = lds[(base + r * 64 + lane) & 4095];
          for (int c = 0; c < 4; c++) b[c] = lds[(base + 128 + c * 64 + lane) & 4095];
          for (int r = 0; r < 2; r++)
            for (int c = 0; c < 4; c++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], b[c], acc[r][c], 0, 0, 0);
        }
      }
      if (MEM) __builtin_nontemporal_store(v + carry, dp + g * 512);
      carry = v;
    }
  }
  float s = carry[0];
  for (int r = 0; r < 2; r++) for (int c = 0; c < 4; c++) for (int e = 0; e < 16; e++) s += acc[r][c][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

static bool smi(double &w, double &mhz) {
  FILE *f = popen("rocm-smi --showpower --showclocks 2>/dev/null", "r");
  if (!f) return false;
  char line[512]; bool pw = false, ck = false;
  while (fgets(line, sizeof line, f)) {
    const char *p = strstr(line, "Package Power (W): ");
    if (p && !strstr(line, "Max")) { w = atof(p + 19); pw = true; }
    p = strstr(line, "sclk clock level:");
    if (p) { const char *q = strchr(p, '('); if (q) { mhz = atof(q + 1); ck = true; } }
  }
  pclose(f);
  return pw && ck;
}

int main(int argc, char **argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 4.0;
  const int bpc = argc > 2 ? atoi(argv[2]) : 1;                   // workgroups per CU (1 or 2)
  const int nblk = 256 * bpc, tiles = 125 / bpc;                  // ~125 tiles per CU and launch = the block kernel's 256-clip launch
  std::vector<unsigned short> h(8 * 4096 * 8);
  srand(1);
  for (auto &v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  bf16x8 *ops; float *out; f32x4 *src, *dst;
  const size_t n4 = (size_t)nblk * tiles * 16384;                 // 8.4 GB each way per launch
  hipMalloc(&ops, h.size() * 2); hipMalloc(&out, nblk * 512 * 4); hipMalloc(&src, n4 * 16); hipMalloc(&dst, n4 * 16);
  hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemset(src, 0x3c, n4 * 16);                                  // (finite, non-zero fp32 pattern)
  const char *names[] = {"MFMA + HBM streams (the block's mix)", "HBM streams alone", "MFMA alone"};
  for (int cs = 0; cs < 3; cs++) {
    std::atomic<bool> stop{false};
    std::vector<double> ws, cl;
    std::thread th([&] { while (!stop) { double w, c; if (smi(w, c)) { ws.push_back(w); cl.push_back(c); } std::this_thread::sleep_for(std::chrono::milliseconds(200)); } });
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      for (int i = 0; i < 4; i++) {
        if (cs == 0) mix_k<true, true><<<nblk, 512>>>(ops, src, dst, out, tiles);
        else if (cs == 1) mix_k<true, false><<<nblk, 512>>>(ops, src, dst, out, tiles);
        else mix_k<false, true><<<nblk, 512>>>(ops, src, dst, out, tiles);
        launches++;
      }
      hipDeviceSynchronize();
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop = true; th.join();
    double w = 0, c = 0; size_t n = 0;
    for (size_t i = 2; i < ws.size(); i++) { w += ws[i]; c += cl[i]; n++; }
    const double ms = el / launches * 1e3, tl = (double)nblk * tiles;
    printf("%-38s %7.3f ms per launch of %d tiles: %6.1f TFLOP/s, %5.2f TB/s (read + write) = %.3f of 8 TB/s; %5.0f W, %5.0f MHz\n", names[cs], ms,
           (int)tl, cs == 1 ? 0.0 : tl * 4096 * 32768.0 / ms / 1e9, cs == 2 ? 0.0 : tl * 524288.0 / ms / 1e9, cs == 2 ? 0.0 : tl * 524288.0 / ms / 1e9 / 8.0,
           n ? w / n : 0.0, n ? c / n : 0.0);
  }
  return 0;
}
